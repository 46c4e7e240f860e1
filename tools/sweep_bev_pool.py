"""Within-process A/B of the bev_pool forward kernel variants on the flagship-size synthetic frame
(interleaved rounds, median + min per variant).  GPU box only:  python tools/sweep_bev_pool.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import _capi, synth  # noqa: E402
from bevfusion_amd.bev_pool import BevPoolPlan  # noqa: E402

NAMES = {1: "wave/cell U4", 2: "wave/cell U8", 3: "coop NW4 U4", 4: "coop NW4 U8", 5: "coop NW8 U4", 6: "coop NW8 U2",
         7: "coop NW4 U2", 14: "U4 batched tail", 15: "U8 batched tail", 16: "striped U4 SW2 bt", 17: "striped U8 SW2 bt",
         18: "striped U4 SW1 bt", 19: "striped U8 SW1 bt", 23: "tile 2x4 U8 bt", 25: "tile 2x4 U4 bt"}
for _r in (1, 2, 4, 16):   # + 100 * R: the stripe -> XCD map moves on every R lines
    for _b in (18, 19):
        NAMES[_b + 100 * _r] = NAMES[_b] + f" rot{_r}"
SAME_ORDER_AS = {v: 1 for v in NAMES if v >= 14 or v == 2}   # one wave per cell, rows in order: the same bits as variant 1
# BEVAMD_SWEEP_ONCE=1: three launches per variant and no timing (for a rocprofv3 --pmc pass: FETCH_SIZE per kernel name)
ONCE = os.environ.get("BEVAMD_SWEEP_ONCE", "0") == "1"
BATCHES = tuple(int(v) for v in os.environ.get("BEVAMD_SWEEP_BATCHES", "1,8").split(","))
VARIANTS = tuple(int(v) for v in os.environ.get("BEVAMD_SWEEP_VARIANTS", "").split(",") if v) or tuple(NAMES)


def main():
    dev = torch.device("cuda:0")
    lib = _capi.load()
    for dtype in (torch.float32, torch.bfloat16):
        for batch in BATCHES:
            inp = synth.bev_pool_inputs(batch=batch, seed=0)
            H, W, D = (int(v) for v in inp["nx"])
            geom = torch.from_numpy(inp["geom"]).to(dev)
            x = torch.from_numpy(inp["feats"]).to(dev).to(dtype)
            plan = BevPoolPlan.from_geometry(geom, batch, inp["origin"], inp["dx"], inp["nx"], want_intervals=True)
            n_kept, n_int = plan.n_kept(), plan.n_intervals()
            C = x.shape[1]
            out = torch.empty((batch, D, H, W, C), dtype=torch.float32, device=dev)
            alg = n_kept * C * x.element_size() + n_int * 24 + out.numel() * 4

            def run(v):
                rc = lib.bevamd_bev_pool_forward_cells_tuned(_capi.ptr(x), int(dtype == torch.bfloat16), _capi.ptr(plan.order),
                                                            _capi.ptr(plan.cell_start), _capi.ptr(out), plan.n, C, batch, D, H,
                                                            W, v, _capi.stream_ptr(dev))
                _capi.check(rc, f"variant {v}")

            run(1)
            ref = out.clone()
            times = {v: [] for v in VARIANTS}
            refs = {1: ref}
            for v in VARIANTS:
                out.fill_(float("nan"))
                run(v)
                err = float((out - ref).abs().max())
                assert err < 1e-3, (v, err)
                if v in SAME_ORDER_AS and SAME_ORDER_AS[v] in refs:
                    assert torch.equal(out, refs[SAME_ORDER_AS[v]]), f"variant {v} differs from {SAME_ORDER_AS[v]}"
            if ONCE:
                for v in VARIANTS:
                    for _ in range(3):
                        run(v)
                torch.cuda.synchronize()
                print(f"--- dtype={dtype} batch={batch}: 3 launches per variant done")
                continue
            for _ in range(15):
                for v in VARIANTS:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        run(v)
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / 5)
            print(f"--- dtype={dtype} batch={batch} kept={n_kept} intervals={n_int} alg_bytes={alg / 1e6:.1f} MB")
            for v in VARIANTS:
                med, mn = float(np.median(times[v])), float(np.min(times[v]))
                print(f"  v{v} {NAMES[v]:14s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  -> {alg / med / 1e6:7.0f} GB/s "
                      f"({alg / med / 1e6 / 80:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
