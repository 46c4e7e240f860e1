timeout 300 python tests/graph_probe.py --ahead 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_spconv_fused.py -x -q 2>&1 | tail -3
