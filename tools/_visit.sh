timeout 1200 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu 2>&1 | tail -25
