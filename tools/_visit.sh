timeout 900 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_sparse_bn.py -x -q 2>&1 | tail -5
for s in 192 256 512; do echo "== slabs $s"; BEVAMD_BN_SLABS=$s timeout 200 python tools/time_bn.py 2>&1 | tail -1; done
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline > gpurun_out/v1_train_aug.log 2>&1; grep "^{" gpurun_out/v1_train_aug.log | tail -1 | cut -c1-2500
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline --train-inputs static > gpurun_out/v1_train_static.log 2>&1; grep "^{" gpurun_out/v1_train_static.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['stage_ms'])"
