timeout 900 python -m pytest tests/test_gpu_spconv_slab.py tests/test_gpu_keyorder.py tests/test_gpu_flagship_oracle.py tests/test_gpu_fused_train.py -x -q -m gpu 2>&1 | tail -3
