"""Generates tests/golden/spconv_ref_*.npz with the REFERENCE's own CPU functors
(oracle/_ref/sparse_conv_ext, compiled from /root/reference/mmdet3d/ops/spconv by oracle/ref_build.py):
get_indice_pairs_3d (rulebook) and indice_conv_fp32 / indice_conv_backward_fp32 on CPU tensors.

Runs in the build container (CPU):   python tests/golden/make_spconv_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_build  # noqa: E402
from oracle import get_conv_output_size  # noqa: E402

# name: (batch, spatial_shape, points/sample, cin, cout, ksize, stride, padding, subm, seed)
CASES = {
    "subm3":    (2, (12, 10, 7), 150, 5, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 0),
    "conv_s2":  (2, (12, 10, 7), 150, 16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0, 1),
    "conv_p110": (2, (13, 11, 9), 200, 8, 8, (3, 3, 3), (2, 2, 2), (1, 1, 0), 0, 2),   # stage-3 padding
    "conv_out": (3, (9, 9, 5), 120, 8, 16, (1, 1, 3), (1, 1, 2), (0, 0, 0), 0, 3),      # conv_out geometry
    "subm_dense": (1, (4, 4, 4), 64, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 4),      # fully dense grid
}


def main():
    ext = ref_build.load_ref("sparse_conv_ext")
    for name, (B, shape, npts, cin, cout, ks, st, pd, subm, seed) in CASES.items():
        rng = np.random.default_rng(seed)
        idx = []
        for b in range(B):
            lin = rng.choice(int(np.prod(shape)), size=min(npts, int(np.prod(shape))), replace=False)
            xyz = np.stack(np.unravel_index(lin, shape), 1)
            idx.append(np.concatenate([np.full((len(lin), 1), b), xyz], 1))
        indices = np.concatenate(idx).astype(np.int32)
        rng.shuffle(indices, axis=0)
        out_shape = list(shape) if subm else get_conv_output_size(list(shape), list(ks), list(st), list(pd), [1, 1, 1])
        res = ext.get_indice_pairs_3d(torch.from_numpy(indices), B, out_shape, list(shape), list(ks), list(st), list(pd),
                                      [1, 1, 1], [0, 0, 0], subm, 0)
        out_inds, pairs, num = [t.clone() for t in res]
        feats = rng.standard_normal((indices.shape[0], cin)).astype(np.float32)
        w = (rng.standard_normal(ks + (cin, cout)) * 0.2).astype(np.float32)
        out = ext.indice_conv_fp32(torch.from_numpy(feats), torch.from_numpy(w), pairs, num, out_inds.shape[0], 0, subm)
        og = rng.standard_normal(tuple(out.shape)).astype(np.float32)
        gi, gw = ext.indice_conv_backward_fp32(torch.from_numpy(feats), torch.from_numpy(w), torch.from_numpy(og), pairs,
                                               num, 0, subm)
        path = os.path.join(HERE, f"spconv_ref_{name}.npz")
        np.savez_compressed(path, indices=indices, batch_size=B, spatial_shape=np.array(shape), ksize=np.array(ks),
                            stride=np.array(st), padding=np.array(pd), subm=subm, out_shape=np.array(out_shape),
                            out_indices=out_inds.numpy(), indice_pairs=pairs.numpy(), indice_num=num.numpy(),
                            features=feats, filters=w, out=out.numpy(), out_grad=og, in_grad=gi.numpy(),
                            filter_grad=gw.numpy())
        print(name, "N", indices.shape[0], "M", out_inds.shape[0], "pairs", int(num.sum()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
