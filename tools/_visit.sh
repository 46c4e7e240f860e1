timeout 900 python -m pytest tests/test_gpu_fused_train.py -x -q 2>&1 | tail -3
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline > gpurun_out/v10_train.log 2>&1; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last_full.json'))
print(d['ms_per_step'], d['config']['stage_ms'])
print('fwd per step', d['config']['encoder_fwd_per_step_ms'][-6:])
PY
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline --train-inputs static > gpurun_out/v10_train.log 2>&1; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last_full.json'))
print(d['ms_per_step'], d['config']['stage_ms'])
PY
