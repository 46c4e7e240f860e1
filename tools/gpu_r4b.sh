#!/bin/bash
# round 4, second GPU visit: column-path tests, bench with extras + the child-process CPU baseline, PMC of the fused pooling kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_keyorder.py tests/test_gpu_bev_pool.py -x -q > gpurun_out/r4b_new_tests.log 2>&1
echo "== new tests rc=$?"; tail -8 gpurun_out/r4b_new_tests.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r4b_bench.log 2>gpurun_out/r4b_bench.err
echo "== bench rc=$?"; tail -3 gpurun_out/r4b_bench.err | cut -c1-400
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4b_bench.log') if x.startswith('{')]
if l:
    r=json.loads(l[-1]); print(r['value'], r['ms_per_step'], r['config']['stage_ms']); print(json.dumps(r.get('extra'))[:3000])
    cb=r['cpu_baseline']; print({k:cb[k] for k in cb if k!='sample'})
PY
bash tools/pmc_script.sh fused "bev_" -- python $GRAFT_REPO_ROOT/tools/time_fused_pool.py 8 > gpurun_out/r4b_pmc_fused.txt 2>&1; cat gpurun_out/r4b_pmc_fused.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4b_tests.log 2>&1
echo "== all tests rc=$?"; tail -5 gpurun_out/r4b_tests.log | cut -c1-300
