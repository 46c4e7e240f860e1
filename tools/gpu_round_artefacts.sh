#!/bin/bash
# the artefacts of a round that go into profiles/ (names carry $ROUND_TAG, default r06) — — whole suite, default bench line, kernel-trace stats, step timelines (8 frames, 1
# frame), per-kernel PMC of the inference step, per-layer convolution profile, bev_pool traffic, train-step traces
mkdir -p gpurun_out
RT=${ROUND_TAG:-r06}
RN=$(echo $RT | tr -cd 0-9 | sed "s/^0*//")   # round number of the tag (r06a -> 6)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# health check first (round 5: a box whose GPU faulted in every process ate a whole visit): smoke() must pass, else stop at once
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${RT}_health.log 2>&1 || { echo "== health check failed: stop"; tail -5 gpurun_out/${RT}_health.log; exit 3; }
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${RT}_tests.log 2>&1
rc=$?; echo "== tests rc=$rc"; grep -E "^E |FAILED|passed|failed" gpurun_out/${RT}_tests.log | tail -4 | cut -c1-300
if [ $rc -ge 124 ]; then echo "== the test run died (rc $rc): stop"; exit 3; fi
fi
# kernel-trace stats + timeline of the default step
rm -rf gpurun_out/prof_${RT}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${RT} -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${RT}_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_${RT} --roofline-json gpurun_out/${RT}_roofline_rocprof.json $RN > gpurun_out/${RT}_bench_kernel_trace_stats.txt 2>&1
grep -A2 "by setting" gpurun_out/${RT}_bench_kernel_trace_stats.txt | cut -c1-400
python tools/graph_timeline.py gpurun_out/prof_${RT} > gpurun_out/${RT}_step_timeline.txt 2>&1
head -3 gpurun_out/${RT}_step_timeline.txt | tail -1
# the default line AFTER the kernel trace of this visit, whose solo split it quotes as rocprof_kernel_us (same box, same code)
cp gpurun_out/${RT}_roofline_rocprof.json profiles/roofline_rocprof.json
timeout 900 python bench.py > gpurun_out/${RT}_bench.log 2>gpurun_out/${RT}_bench.err; cp gpurun_out/bench_last_full.json gpurun_out/${RT}_bench_full.json
echo "== bench rc=$?"; grep "^{" gpurun_out/${RT}_bench.log | tail -1 > gpurun_out/${RT}_bench_line.json; cut -c1-300 gpurun_out/${RT}_bench_line.json
# single-frame timeline
rm -rf gpurun_out/prof_${RT}b1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${RT}b1 -o b -- python $R/bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${RT}b1_run.log 2>&1)
python tools/graph_timeline.py gpurun_out/prof_${RT}b1 > gpurun_out/${RT}_step_timeline_batch1.txt 2>&1
grep "^{" gpurun_out/prof_${RT}b1_run.log | tail -1 > gpurun_out/${RT}_bench_line_batch1.json
head -3 gpurun_out/${RT}_step_timeline_batch1.txt | tail -1
# per-kernel PMC of the inference step (eager dispatches) + per-layer convolution profile + bev_pool traffic
bash tools/pmc_bench.sh infer --no-graph > gpurun_out/${RT}_pmc_run.log 2>&1
cp gpurun_out/pmcb_infer.txt gpurun_out/${RT}_pmc_infer_per_kernel.txt; cp gpurun_out/pmcb_infer.json gpurun_out/${RT}_pmc_infer_per_kernel.json
head -12 gpurun_out/${RT}_pmc_infer_per_kernel.txt | cut -c1-160
python tools/spconv_layers_profile.py gpurun_out/prof_${RT} gpurun_out/pmcb_infer gpurun_out/${RT}_bench_full.json gpurun_out/${RT}_spconv_layers.json > gpurun_out/${RT}_spconv_layers.txt 2>&1
head -30 gpurun_out/${RT}_spconv_layers.txt | cut -c1-200
python tools/update_bev_pool_traffic.py gpurun_out/pmcb_infer.json $RN > gpurun_out/${RT}_bev_pool_traffic.log 2>&1; cp profiles/bev_pool_traffic.json gpurun_out/${RT}_bev_pool_traffic.json; tail -4 gpurun_out/${RT}_bev_pool_traffic.log
# training step: lines + kernel trace of the --amp step (SKIP_TRAIN=1: the training path did not change since the last visit)
if [ -n "$SKIP_TRAIN" ]; then find gpurun_out -name "*.db" -delete; find gpurun_out/pmcb_infer -name "*.csv" -size +4M -delete; exit 0; fi
for mode in "--amp" "" "--amp --train-inputs static"; do
  tagm=$(echo "$mode" | tr -d ' ' | sed 's/--/_/g')
  timeout 400 python bench.py --mode train-step --no-cpu-baseline $mode > gpurun_out/${RT}_train${tagm}.log 2>&1
  grep "^{" gpurun_out/${RT}_train${tagm}.log | tail -1 > gpurun_out/${RT}_bench_line_train_step${tagm}.json
  python -c "import sys,json; d=json.load(open('gpurun_out/${RT}_bench_line_train_step${tagm}.json')); print('train ${mode}', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('encoder_path'), {k: round(v,2) for k,v in d['config']['stage_ms'].items()})"
done
rm -rf gpurun_out/prof_${RT}t
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${RT}t -o b -- python $R/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${RT}t_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_${RT}t > gpurun_out/${RT}_train_step_amp_kernel_trace_stats.txt 2>&1
python tools/train_timeline.py gpurun_out/prof_${RT}t > gpurun_out/${RT}_train_step_amp_timeline.txt 2>&1
head -12 gpurun_out/${RT}_train_step_amp_kernel_trace_stats.txt | cut -c1-150
rm -rf gpurun_out/prof_${RT}tf
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${RT}tf -o b -- python $R/bench.py --mode train-step --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${RT}tf_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_${RT}tf > gpurun_out/${RT}_train_step_fp32_kernel_trace_stats.txt 2>&1
find gpurun_out -name "*.db" -delete; find gpurun_out/pmcb_infer -name "*.csv" -size +4M -delete
