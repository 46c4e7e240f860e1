#!/bin/bash
# round 5, visit 2: predicated zero reads (ID flag 4) and 64-row blocks on the register-filter slab kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_spconv_slab.py tests/test_gpu_reference_gpu_goldens.py -x -q > gpurun_out/v2_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v2_tests.log | tail -8 | cut -c1-300
timeout 600 python tools/time_slab_variant.py 64:1644222 64:1644226 128:1644220 128:1644224 128:1642220 128:1642224 2>&1 | tail -8
FRAMES=1 timeout 300 python tools/time_slab_variant.py 64:1644222 64:1644226 128:1644220 128:1644224 128:1642220 128:1642224 2>&1 | tail -8
# LDS counters of the 64-channel layer, base vs predicated
for spec in "base64:64:1644222" "pred64:64:1644226" "base128:128:1644220" "pred128:128:1644224"; do
  tag=${spec%%:*}; rest=${spec#*:}; cin=${rest%%:*}; variant=${rest##*:}
  out=$R/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
  i=0
  for grp in "SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- python $R/tools/prof_slab.py $cin $variant 8 3 > $out/g${i}_run.log 2>&1)
  done
  for d in $out/g*/; do python tools/rocprof_summary.py $d 2>/dev/null | grep -E "spconv_(stream|resident|slab)" ; done > $out/summary.txt
  echo "== $tag"; cut -c1-260 $out/summary.txt
  find $out -name "*.db" -delete; find $out -name "*agent_info.csv" -delete
done
# in the step: default variants vs predicated ones
for rep in 1 2; do
  for v in "" "64:1644226" "64:1644226,128:1644224"; do
    BEVAMD_SPCONV_SLAB_VARIANTS=$v timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('variants [$v]'.ljust(40), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()})"
  done
done
