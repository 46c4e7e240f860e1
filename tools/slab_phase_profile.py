"""Cycle breakdown of the slab kernel's main loop (needs `python -m bevfusion_amd.build --profiling`): per wave and step, cycles
spent issuing DMA requests, multiplying (LDS fragment reads + MFMAs), waiting for DMA (counted vmcnt) and at the barrier.
    python tools/slab_phase_profile.py [frames=8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import _capi, synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
lib = _capi.load()
cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape, ind = list(cfg["sparse_shape"]), vc.int().contiguous()
for cout, ks, st, pd in [(32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]:
    rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
    c = cout
    f = torch.randn(ind.shape[0], c, device=dev).half()
    w = (torch.randn(27, c, c, device=dev) / (27 * c) ** 0.5).half()
    img = sops.make_filter_image(w.view(27, 1, 1, c, c))
    for v in (sops.slab_variants(c) if os.environ.get("SLAB_PHASES") == "1" else []):
        meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v)
        prof = torch.zeros(8, dtype=torch.int64, device=dev)
        lib.bevamd_spconv_slab_set_profile_buffer(_capi.ptr(prof))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v)
        e.record()
        e.synchronize()
        lib.bevamd_spconv_slab_set_profile_buffer(None)
        p = prof.cpu().tolist()
        waves = max(p[4], 1)
        tot = sum(p[:4])
        print(f"{c:3d}->{c:<3d} variant {v}: {s.elapsed_time(e) * 1e3:7.1f} us (timers on); per wave: issue {p[0] / waves:8.0f}  multiply {p[1] / waves:8.0f}  "
              f"dma-wait {p[2] / waves:8.0f}  barrier {p[3] / waves:8.0f} cycles  ({100 * p[0] / tot:.0f} / {100 * p[1] / tot:.0f} / {100 * p[2] / tot:.0f} / {100 * p[3] / tot:.0f} %)")

    # compile-time ablation (tools/slab_ablation.sh rebuilds the library per mask): time the default variant of this build
    mask = lib.bevamd_spconv_slab_ablation_mask()
    v = sops.slab_variants(c)[0]
    meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
    for _ in range(3):
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v)
    e.record()
    e.synchronize()
    print(f"ABL mask {mask:3d}  {c}->{c} variant {v}: {s.elapsed_time(e) * 100:7.1f} us")
