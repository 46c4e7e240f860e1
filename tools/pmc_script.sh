#!/bin/bash
# PMC passes over an arbitrary python command:  tools/pmc_script.sh <tag> <kernel-name filter> -- python tools/xyz.py args
tag=$1; filt=$2; shift 3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcs_$tag
rm -rf $out; mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- "$@" > $out/g${i}_run.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out gpurun_out/pmcs_${tag}.json | grep -E "kernel|$filt" | cut -c1-170
find $out -name "*agent_info.csv" -delete
