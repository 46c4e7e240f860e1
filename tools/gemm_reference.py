"""What a dense GEMM library gets on the SAME multiply-adds as the wide SubM layers: the materialised im2col matrix [rows, 27 * cin]
(random fp16, already resident — no gather, no rulebook) times the filter [27 * cin, cout] through torch.matmul (hipBLASLt / rocBLAS).
The staged-rows kernels do this product AND the gather; the figure says how far from the machine's practical MFMA rate they are.
    python tools/gemm_reference.py [frames]"""
import sys
import torch

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rows = {32: 2075436, 64: 788022, 128: 192226}   # flagship synthetic clouds, 8 frames (profiles/r06_bench_full.json)
    dev = torch.device("cuda", 0)
    for c, r8 in rows.items():
        r = r8 * frames // 8
        a = torch.randn(r, 27 * c, device=dev, dtype=torch.float16)
        w = (torch.randn(27 * c, c, device=dev) / (27 * c) ** 0.5).half()
        out = torch.empty(r, c, device=dev, dtype=torch.float16)
        us = min(timeit(lambda: torch.matmul(a, w, out=out)) for _ in range(3))
        gf = 2.0 * r * 27 * c * c / 1e9
        print(f"dense GEMM [{r} x {27 * c}] x [{27 * c} x {c}] fp16: {us:7.1f} us = {gf / us * 1e3:6.1f} TFLOP/s (reads {a.numel() * 2 / 1e6:.0f} MB of A: {a.numel() * 2 / us / 1e6:.2f} TB/s)", flush=True)
        del a
    # the machine's square-GEMM rate on random data, for scale
    for n in (4096, 8192):
        a = torch.randn(n, n, device=dev, dtype=torch.float16); b = torch.randn(n, n, device=dev, dtype=torch.float16)
        us = min(timeit(lambda: torch.matmul(a, b)) for _ in range(3))
        print(f"dense GEMM {n}^3 fp16: {us:8.1f} us = {2.0 * n ** 3 / us * 1e-6:6.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
