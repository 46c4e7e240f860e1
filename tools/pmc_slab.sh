#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass; never combined with trace domains) over ONE SubM layer shape:
#   tools/pmc_slab.sh <tag> <cin> <variant (-1 = gather kernel)> [frames]
tag=${1:-p}; cin=${2:-64}; variant=${3:-0}; frames=${4:-8}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- python $GRAFT_REPO_ROOT/tools/prof_slab.py $cin $variant $frames 3 > $out/g${i}_run.log 2>&1
  echo "group $i rc=$?"
done
cd $GRAFT_REPO_ROOT
for d in $out/g*/; do python tools/rocprof_summary.py $d 2>/dev/null | grep -E "spconv_(stream|resident|slab)" ; done > $out/summary.txt
find $out -name "*.db" -delete; find $out -name "*agent_info.csv" -delete
wc -l $out/summary.txt
