import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevfusion_amd import _capi, synth
from bevfusion_amd.spconv import ops as sops
from bevfusion_amd.voxel import voxelize_batch
frames = 8; dev = torch.device("cuda", 0); lib = _capi.load(); cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape, ind = list(cfg["sparse_shape"]), vc.int().contiguous()
rbs = sops.build_rulebook(ind, frames, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], 1, False)
ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
c = 32
f = torch.randn(ind.shape[0], c, device=dev).half(); res = torch.randn(ind.shape[0], c, device=dev).half()
img = sops.make_filter_image((torch.randn(27, c, c, device=dev) / (27 * c) ** 0.5).half().view(27, 1, 1, c, c))
for v in [int(x) for x in sys.argv[1:]]:
    meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
    for kw in (dict(), dict(residual=res, relu=True)):
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v, **kw)
        prof = torch.zeros(8, dtype=torch.int64, device=dev)
        lib.bevamd_spconv_slab_set_profile_buffer(_capi.ptr(prof))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v, **kw); e.record(); e.synchronize()
        lib.bevamd_spconv_slab_set_profile_buffer(None)
        p = prof.cpu().tolist(); w = max(p[4], 1)
        print(f"variant {v} {'res+relu' if kw else 'plain':8s}: {s.elapsed_time(e) * 1e3:7.1f} us; per wave cycles (x100 ns ticks?): wait {p[0] / w:9.0f} planes {p[1] / w:9.0f} finish {p[2] / w:9.0f} other {p[3] / w:9.0f} waves {p[4]}")
