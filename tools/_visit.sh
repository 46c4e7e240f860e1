timeout 900 python -m pytest tests/test_gpu_fused_train.py tests/test_gpu_wgrad_slab.py -x -q 2>&1 | tail -5
for m in "" "--train-inputs static"; do
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline $m > gpurun_out/v11_train.log 2>&1; grep "^{" gpurun_out/v11_train.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['encoder_path'], d['config']['stage_ms'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || tail -20 gpurun_out/v11_train.log
done
