for i in 1 2; do
BEVAMD_LIB=bevfusion_amd/lib/exp/noskip.so python tools/time_slab_variant.py 16:3000256 2>&1 | grep "16->16"
python tools/time_slab_variant.py 16:3000256 2>&1 | grep "16->16"
done
timeout 900 python -m pytest tests/test_gpu_spconv_slab.py tests/test_gpu_keyorder.py -x -q 2>&1 | tail -3
