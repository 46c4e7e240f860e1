#!/bin/bash
# tools/exp_build.sh <name> "<extra hipcc flags>" file.hip [file.hip ...]: an EXPERIMENT build of the library — the named sources
# recompiled with the extra flags, everything else from the regular objects — into bevfusion_amd/lib/exp/<name>.so (git-ignored,
# travels with gpurun).  Select it at run time with BEVAMD_LIB=bevfusion_amd/lib/exp/<name>.so (A/B on the same box).
set -e
name="$1"; flags="$2"; shift 2
cd "$(dirname "$0")/.."
python -m bevfusion_amd.build > /dev/null
mkdir -p bevfusion_amd/lib/exp/obj_$name
objs=""
for o in bevfusion_amd/lib/obj/*.o; do
  b=$(basename "$o" .o); use="$o"
  for f in "$@"; do
    if [ "$(basename "$f" .hip)" = "$b" ]; then
      use="bevfusion_amd/lib/exp/obj_$name/$b.o"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $flags -c "bevfusion_amd/csrc/$b.hip" -o "$use" &
    fi
  done
  objs="$objs $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "bevfusion_amd/lib/exp/$name.so" $objs
echo "bevfusion_amd/lib/exp/$name.so"
