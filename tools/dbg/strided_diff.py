import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, "tests")
from test_gpu_keyorder import random_sorted_set, _filters
from bevfusion_amd.spconv import fused, ops as sops
dev = torch.device("cuda:0")
dtype = torch.float16
batch, shape, n, pad, variant = 2, [24, 20, 9], 2500, (1, 1, 1), 3000256
rng = np.random.default_rng(n)
c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
m = c4.shape[0]; cap = m + 50
buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev); buf[:m] = ct
n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
lvl = fused.Level(buf, cap, n_dev, batch, shape, linear_order=True)
x = torch.from_numpy(rng.standard_normal((cap, 16)).astype(np.float32) * 0.5).to(dev).to(dtype)
w = _filters(rng, (3, 3, 3), 16, 32, dev, dtype)
img = sops.make_filter_image(w)
rows = sops.slab_block_rows(16, variant)
out, nbr = lvl.downsample([3, 3, 3], [2, 2, 2], list(pad))
meta = lvl.down_slab([3, 3, 3], [2, 2, 2], list(pad), rows)
mo = int(out.n_dev.item())
scale = torch.from_numpy(rng.uniform(0.7, 1.3, 32).astype(np.float32)).to(dev)
shift = torch.from_numpy(rng.standard_normal(32).astype(np.float32) * 0.1).to(dev)
for name, kw in (("plain", dict()), ("bn", dict(bn_scale=scale, bn_shift=shift)), ("bnrelu", dict(bn_scale=scale, bn_shift=shift, relu=True)), ("relu", dict(relu=True))):
    got = sops.sparse_conv_slab(x, img, meta, out.n_cap, 16, 32, num_out_dev=out.n_dev, variant=variant, **kw)[:mo]
    ref = sops.sparse_conv_tiled(x, img, nbr, out.n_cap, 27, 16, 32, num_out_dev=out.n_dev, **kw)[:mo]
    d = (got != ref)
    print(name, "mismatches", int(d.sum()), "of", d.numel(), "cols", d.any(0).nonzero().flatten().tolist()[:40])
    if d.any():
        r, c = d.nonzero()[0].tolist()
        raw = sops.sparse_conv_slab(x, img, meta, out.n_cap, 16, 32, num_out_dev=out.n_dev, variant=variant)[:mo]
        v = float(raw[r, c]); s, h = float(scale[c]), float(shift[c])
        print(" at", r, c, "got", float(got[r, c]), "ref", float(ref[r, c]), "conv", v, "scale", s, "shift", h,
              "fma32", np.float32(np.float32(v) * np.float32(s)) + np.float32(h), "exact", v * s + h)
