#!/bin/bash
# Compile-time ablation of the slab kernels (runs ON the GPU box: hipcc is in the image): for every mask rebuild
# csrc/spconv_slab_f16.hip with -DBEVAMD_SLAB_ABL=mask, relink, time the default variants at 8 frames.  Restores mask 0 at the end.
#   bash tools/slab_ablation.sh "0 3 4 8 12 16 31 32 63 128 256"
masks=${1:-"0 3 4 12 16 31 63 128 256"}
cd $GRAFT_REPO_ROOT
obj=bevfusion_amd/lib/obj
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -fno-gpu-rdc"
for m in $masks 0; do
  /opt/rocm/bin/hipcc $flags -DBEVAMD_SLAB_ABL=$m -c bevfusion_amd/csrc/spconv_slab_f16.hip -o $obj/spconv_slab_f16.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bevfusion_amd/lib/libbevfusion_amd.so $obj/*.o || exit 1
  python tools/slab_phase_profile.py 8 2>&1 | grep "^ABL"
done
