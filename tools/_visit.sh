timeout 900 python -m pytest tests/test_gpu_spconv_slab.py -x -q -m gpu 2>&1 | tail -15
for e in 0 1 3; do EPI=$e timeout 300 python tools/time_slab_variant.py 64:1644228 64:4200128 2>&1 | grep -E "variant|rror"; done
