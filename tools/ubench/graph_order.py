"""When does a forked branch of a replayed HIP graph start?  A chain of N tiny kernels on the capture stream; a side branch of M
kernels on a second stream that depends on node K of the chain; the side branch is ISSUED (= its nodes are created) either right
after node K ("early") or after the whole chain ("late").  Run under `rocprofv3 --kernel-trace`; tools/ubench/graph_order_report.py
prints, per variant, how long after node K's end the first side kernel starts.

    python tools/ubench/graph_order.py [N=60] [K=10] [M=5] [replays=20]"""
import sys
import torch

N, K, M, R = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 60), (2, 10), (3, 5), (4, 20)))
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)


def build(order):
    a = torch.zeros(4096, device=dev)
    b = torch.zeros(4096, device=dev)
    c = torch.zeros(4096, device=dev)
    g = torch.cuda.CUDAGraph()

    def branch():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(M):
                b.sin_()            # the side branch: sin kernels

    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        c.cos_()                    # marks the start of a replay
        ev = None
        for i in range(N):
            a.add_(1.0)             # the chain: add kernels
            if i == K:
                if order == "early":
                    branch()
                else:
                    ev = torch.cuda.Event()
                    ev.record(main)
        if order == "late":
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for _ in range(M):
                    b.sin_()
        main.wait_stream(side)
        a.mul_(2.0)
    return g, (a, b, c)


for order in ("early", "late"):
    g, keep = build(order)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    # tag: one tanh kernel in front of each variant's timed replays
    keep[2].tanh_()
    torch.cuda.synchronize()
    for _ in range(R):
        g.replay()                  # back to back: the host runs ahead of the GPU like in the benchmark
    torch.cuda.synchronize()
    keep[2].tanh_()
    torch.cuda.synchronize()
    for _ in range(R):
        g.replay()
        torch.cuda.synchronize()    # one at a time: the launch itself is exposed
print("done", N, K, M, R)
