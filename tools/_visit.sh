# scratch: the command list of the current GPU visit (overwritten per visit; tools/gpu_visit.sh and tools/gpu_round_artefacts.sh are the kept ones)
timeout 900 python -m pytest tests/test_gpu_spconv_slab.py -x -q -m gpu 2>&1 | tail -3
