# filter-stationary 32 -> 32 kernel ablations (wrong results by design): 1 no row requests, 2 no fragment reads, 4 one MFMA per tap, 8 no epilogue, 16 no slot / residual requests
timeout 300 python tools/time_slab_variant.py 32:4000112 2>&1 | grep variant
for m in 1 2 4 8 16 6 31; do
  BEVAMD_LIB=bevfusion_amd/lib/exp/fabl$m.so timeout 300 python tools/time_slab_variant.py 32:4000112 2>&1 | grep -E "variant|rror"
done
timeout 300 python tools/time_slab_variant.py 32:4000112 2>&1 | grep variant
