#!/bin/bash
# round 6, job h: the rest of the -m gpu suite (from the test that failed in job g on)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 2500 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step scene timeout -k 10 300 python -m pytest tests/test_fullframe_gpu.py -q -m gpu -s -k "optimizer_trained_nets"
step suite timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=10 --deselect tests/test_bench_gpu.py --deselect tests/test_coresidency_gpu.py -k "not full_frame_sampler_indices and not test_abi"
