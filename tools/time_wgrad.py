"""Time the filter gradient of the SubM layers of the flagship encoder at the training batch (4 frames, 120 k-voxel cap): the gather
kernel (bevamd_spconv_conv_wgrad) against the staged-rows kernel (bevamd_spconv_conv_wgrad_slab), level by level.
    [FRAMES=4] [ORDER=key|first] python tools/time_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402
from tools.sweep_spconv import timeit  # noqa: E402


def main():
    frames = int(os.environ.get("FRAMES", "4"))
    order = os.environ.get("ORDER", "key")
    dev = torch.device("cuda", 0)
    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][0], order=order)
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    stages = [(16, None), (32, ((3, 3, 3), (2, 2, 2), (1, 1, 1))), (64, ((3, 3, 3), (2, 2, 2), (1, 1, 1))), (128, ((3, 3, 3), (2, 2, 2), (1, 1, 0)))]
    total_old = total_new = 0.0
    for c, down in stages:
        if down is not None:
            ks, st, pd = down
            rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
            ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
        rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
        x = torch.randn(ind.shape[0], c, device=dev).half()
        g = torch.randn(ind.shape[0], c, device=dev).half()
        w = torch.zeros(3, 3, 3, c, c, device=dev).half()
        nbr, nbr_t = rb.conv_tables()
        os.environ["BEVAMD_SPCONV_WGRAD_SLAB"] = "0"
        lib = sops._capi.load()
        wsb = lib.bevamd_spconv_wgrad_workspace_bytes(27, c, c)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        fg = torch.empty_like(w)

        def old():
            rc = lib.bevamd_spconv_conv_wgrad(sops._capi.ptr(x), sops._capi.ptr(g), 1, sops._capi.ptr(nbr), nbr.shape[1], rb.num_out, 27, c, c,
                                              sops._capi.ptr(fg), sops._capi.ptr(ws), wsb, sops._capi.stream_ptr(dev))
            assert rc == 0

        t_old = min(timeit(old)[0] for _ in range(3))
        os.environ["BEVAMD_SPCONV_WGRAD_SLAB"] = "1"
        rb._slab128 = None
        meta = rb.slab_meta_wgrad(c)
        if meta is None:
            print(f"{c:3d}->{c:<3d} rows={rb.num_out:8d} gather {t_old:7.1f} us   staged rows: not eligible (rows not in linear order)", flush=True)
            total_old += 4 * t_old
            total_new += 4 * t_old
            continue
        t_new = min(timeit(lambda: sops.sparse_conv_wgrad_slab(x, g, meta, c, c))[0] for _ in range(3))
        if os.environ.get("WGS_PROF"):   # BEVAMD_LIB = a -DBEVAMD_WGS_PROF build: cycle sums of the phases over all waves
            prof = torch.zeros(8, dtype=torch.int64, device=dev)
            lib.bevamd_spconv_wgrad_slab_set_profile_buffer(sops._capi.ptr(prof))
            sops.sparse_conv_wgrad_slab(x, g, meta, c, c)
            torch.cuda.synchronize()
            lib.bevamd_spconv_wgrad_slab_set_profile_buffer(None)
            pv = prof.tolist()
            tot = sum(pv[:4]) or 1
            print(f"    phases (share of wave cycles, {pv[4]} waves, {tot / max(pv[4], 1):.0f} cycles per wave): issue+bake {pv[0] / tot:.2f}  multiply {pv[1] / tot:.2f}  "
                  f"dma wait {pv[2] / tot:.2f}  barrier {pv[3] / tot:.2f}", flush=True)
        new = sops.sparse_conv_wgrad_slab(x, g, meta, c, c).float()
        old()
        err = float((new.view(-1) - fg.float().view(-1)).abs().max() / (1 + fg.float().abs().max()))
        nblk = (rb.num_out + 127) // 128
        cnt = (meta.hdr[:nblk * 24].view(torch.int32).view(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF).float()
        print(f"{c:3d}->{c:<3d} rows={rb.num_out:8d} gather {t_old:7.1f} us   staged rows {t_new:7.1f} us  ({t_old / t_new:4.1f} x)   "
              f"max rel diff {err:.1e}   staged rows per plane: mean {float(cnt.mean()):.0f} max {int(cnt.max())}", flush=True)
        total_old += 4 * t_old
        total_new += 4 * t_new
    print(f"16 SubM layers (4 per level): gather {total_old / 1e3:.2f} ms, staged rows {total_new / 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
