#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-tr}
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_ddp.py tests/test_gpu_modules.py -q -x 2>&1 | tail -4
for mode in "" "--amp"; do
  timeout 400 python bench.py --mode train-step --no-cpu-baseline $mode > gpurun_out/${tag}_train${mode}.log 2>&1
  echo "== train $mode rc=$?"; tail -1 gpurun_out/${tag}_train${mode}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -15 gpurun_out/${tag}_train${mode}.log
done
rm -rf gpurun_out/prof_$tag
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --steps 5 --warmup 2 --no-cpu-baseline $2 > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt 2>&1
head -30 gpurun_out/prof_${tag}_summary.txt | cut -c1-150
find gpurun_out/prof_$tag -name "*.db" -size +20M -delete
