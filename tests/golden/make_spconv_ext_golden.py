"""Generates tests/golden/spconv_ext_*.npz with the REFERENCE's own CPU functors (oracle/_ref/sparse_conv_ext, compiled
from /root/reference/mmdet3d/ops/spconv by oracle/ref_build.py) for the parts of the sparse_conv_ext surface beyond the
SparseEncoder's: transposed and dilated rulebooks, 2D rulebooks, the "inverse" convolution, and max pooling
(get_indice_pairs_{2,3}d, indice_conv_fp32 with inverse=1, indice_maxpool_fp32, indice_maxpool_backward_fp32).

Runs in the build container (CPU):   python tests/golden/make_spconv_ext_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_build  # noqa: E402
from oracle import get_conv_output_size, get_deconv_output_size  # noqa: E402

# name: (batch, spatial_shape, points/sample, channels, ksize, stride, padding, dilation, out_padding, subm, transpose, seed)
CASES = {
    "deconv3_k3s2":  (2, (6, 5, 4), 40, 6, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (1, 1, 1), 0, 1, 10),
    "deconv3_k2s2":  (2, (5, 6, 3), 50, 4, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), (0, 0, 0), 0, 1, 11),
    "deconv3_mixed": (1, (7, 4, 5), 60, 4, (3, 1, 2), (2, 1, 3), (1, 0, 0), (1, 1, 1), (0, 0, 1), 0, 1, 12),
    "subm3_dil2":    (2, (10, 9, 8), 200, 5, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 2, 2), (0, 0, 0), 1, 0, 13),
    "subm3_k2":      (2, (8, 8, 6), 150, 5, (2, 2, 2), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0), 1, 0, 14),
    "conv3_dil2":    (2, (10, 9, 8), 160, 6, (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2), (0, 0, 0), 0, 0, 15),
    "conv3_k2s2":    (2, (10, 8, 6), 160, 6, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), (0, 0, 0), 0, 0, 16),
    "subm2_k3":      (3, (14, 11), 70, 7, (3, 3), (1, 1), (1, 1), (1, 1), (0, 0), 1, 0, 17),
    "conv2_k3s2":    (3, (14, 11), 70, 7, (3, 3), (2, 2), (1, 1), (1, 1), (0, 0), 0, 0, 18),
    "deconv2_k2s2":  (2, (7, 9), 30, 4, (2, 2), (2, 2), (0, 0), (1, 1), (0, 0), 0, 1, 19),
}


def main():
    ext = ref_build.load_ref("sparse_conv_ext")
    for name, (B, shape, npts, C, ks, st, pd, dl, op, subm, transpose, seed) in CASES.items():
        nd = len(shape)
        rng = np.random.default_rng(seed)
        idx = []
        for b in range(B):
            lin = rng.choice(int(np.prod(shape)), size=min(npts, int(np.prod(shape))), replace=False)
            idx.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
        indices = np.concatenate(idx).astype(np.int32)
        rng.shuffle(indices, axis=0)
        if subm:
            out_shape = list(shape)
        elif transpose:
            out_shape = get_deconv_output_size(list(shape), list(ks), list(st), list(pd), list(dl), list(op))
        else:
            out_shape = get_conv_output_size(list(shape), list(ks), list(st), list(pd), list(dl))
        fn = ext.get_indice_pairs_2d if nd == 2 else ext.get_indice_pairs_3d
        res = fn(torch.from_numpy(indices), B, [int(v) for v in out_shape], list(shape), list(ks), list(st), list(pd),
                 list(dl), list(op), subm, transpose)
        out_inds, pairs, num = [t.clone() for t in res]
        M = out_inds.shape[0]
        feats = rng.standard_normal((indices.shape[0], C)).astype(np.float32)
        w = (rng.standard_normal(tuple(ks) + (C, C + 1)) * 0.2).astype(np.float32)
        out = ext.indice_conv_fp32(torch.from_numpy(feats), torch.from_numpy(w), pairs, num, M, 0, subm)
        # the coupled inverse convolution: features on the OUTPUT rows back to the input rows (conv.py:153-158, 206-213)
        feats_o = rng.standard_normal((M, C)).astype(np.float32)
        inv = ext.indice_conv_fp32(torch.from_numpy(feats_o), torch.from_numpy(w), pairs, num, indices.shape[0], 1, 0)
        # max pooling over the same rulebook (pool.py:40-69), inputs with exact ties and negatives
        pf = np.round(rng.standard_normal((indices.shape[0], C)) * 2).astype(np.float32) / 2
        pooled = ext.indice_maxpool_fp32(torch.from_numpy(pf), pairs, num, M)
        pg = rng.standard_normal((M, C)).astype(np.float32)
        pool_grad = ext.indice_maxpool_backward_fp32(torch.from_numpy(pf), pooled, torch.from_numpy(pg), pairs, num)
        path = os.path.join(HERE, f"spconv_ext_{name}.npz")
        np.savez_compressed(path, indices=indices, batch_size=B, spatial_shape=np.array(shape), ksize=np.array(ks),
                            stride=np.array(st), padding=np.array(pd), dilation=np.array(dl), out_padding=np.array(op),
                            subm=subm, transpose=transpose, out_shape=np.array(out_shape, dtype=np.int64),
                            out_indices=out_inds.numpy(), indice_pairs=pairs.numpy(), indice_num=num.numpy(),
                            features=feats, filters=w, out=out.numpy(), features_out=feats_o, inverse_out=inv.numpy(),
                            pool_features=pf, pooled=pooled.numpy(), pool_out_grad=pg, pool_in_grad=pool_grad.numpy())
        print(name, "N", indices.shape[0], "M", M, "pairs", int(num.sum()), os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
