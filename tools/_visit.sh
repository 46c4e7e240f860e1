for f in 1 2 3 4; do echo "frames $f"; FRAMES=$f EPI=3 timeout 300 python tools/time_slab_variant.py 128:1642220 128:1644220 64:1644228 64:1644220 2>&1 | grep -E "variant|rror"; done
