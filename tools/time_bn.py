"""Time the training-mode BatchNorm kernels (csrc/sparse_bn.hip) on the four level shapes of the flagship encoder at the training batch
(4 frames): statistics, apply, backward reduce, backward apply — us per launch and the HBM rate on their algorithmic bytes.
    [BEVAMD_BN_SLABS=512] [BEVAMD_BN_UNROLL=4|8] python tools/time_bn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import _capi  # noqa: E402
from bevfusion_amd.spconv import bn as sbn  # noqa: E402
from tools.sweep_spconv import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lib = _capi.load()
    tot = [0.0, 0.0, 0.0]
    for n, c, layers in ((480000, 16, 5), (843116, 32, 5), (372373, 64, 5), (94589, 128, 5)):
        x = torch.randn(n, c, device=dev).half()
        dy = torch.randn(n, c, device=dev).half()
        bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
        mean = torch.empty(c, device=dev)
        invstd = torch.empty(c, device=dev)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        s1, s2 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ws = sbn._workspace(dev, c)
        st = _capi.stream_ptr(dev)
        P = _capi.ptr

        def stats():
            assert lib.bevamd_sparse_bn_stats(P(x), 1, n, c, c, 1e-3, 0.01, P(mean), P(invstd), P(bn.running_mean), P(bn.running_var), P(ws), ws.numel(), st) == 0

        def apply():
            assert lib.bevamd_sparse_bn_apply(P(x), 1, n, c, c, P(mean), P(invstd), P(bn.weight), P(bn.bias), None, 0, 1, P(y), c, st) == 0

        def bwd():
            assert lib.bevamd_sparse_bn_backward(P(dy), c, P(y), c, P(x), c, 1, n, c, 1, P(mean), P(invstd), P(bn.weight), P(s1), P(s2), P(dx), c, None, 0,
                                                 P(ws), ws.numel(), st) == 0

        t_s = min(timeit(stats)[0] for _ in range(3))
        t_a = min(timeit(apply)[0] for _ in range(3))
        t_b = min(timeit(bwd)[0] for _ in range(3))
        mb = n * c * 2 / 1e6
        print(f"[{n:7d}, {c:3d}] {mb:5.1f} MB/tensor: stats {t_s:6.1f} us ({mb / t_s:5.2f} TB/s)  apply {t_a:6.1f} us ({2 * mb / t_a:5.2f} TB/s)  "
              f"backward (reduce + apply) {t_b:6.1f} us ({(3 + 4) * mb / t_b:5.2f} TB/s on 7 tensor passes)", flush=True)
        tot[0] += layers * t_s
        tot[1] += layers * t_a
        tot[2] += layers * t_b
    print(f"20 layers: stats {tot[0] / 1e3:.2f} ms, apply {tot[1] / 1e3:.2f} ms, backward {tot[2] / 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
