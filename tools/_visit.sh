for rep in 1 2 3; do
timeout 300 python tools/time_slab_variant.py 32:4000112 2>&1 | grep variant
BEVAMD_LIB=bevfusion_amd/lib/exp/oldepi.so timeout 300 python tools/time_slab_variant.py 32:4000112 2>&1 | grep variant
done
