for pct in 100 75 150 200; do echo "pct $pct"; BEVAMD_SLAB_GRID_PCT=$pct EPI=3 timeout 300 python tools/time_slab_variant.py 32:4100128 64:1644228 128:1644220 2>&1 | grep -E "variant|rror"; done
