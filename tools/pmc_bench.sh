#!/bin/bash
# rocprofv3 PMC passes over a bench command (one counter group per pass, kernel-trace only — never combined with other trace
# domains), summarised per kernel as JSON + text:
#   tools/pmc_bench.sh <tag> [bench.py args...]      e.g.  tools/pmc_bench.sh infer --no-graph     tools/pmc_bench.sh train --mode train-step --amp
# --no-graph for the inference step: counters are collected per eager dispatch (the same kernels the HIP graph replays).
tag=${1:-x}; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/g${i}_run.log 2>&1
  echo "group $i ($grp) rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out gpurun_out/pmcb_${tag}.json > gpurun_out/pmcb_${tag}.txt
head -60 gpurun_out/pmcb_${tag}.txt | cut -c1-200
find $out -name "*.db" -delete; find $out -name "*agent_info.csv" -delete; find $out -name "*kernel_trace.csv" -size +8M -delete; find $out -name "*counter_collection.csv" -size +8M -delete
