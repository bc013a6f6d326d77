#!/bin/bash
# round 5, job b: where the paired-workgroup corruption enters (per-layer xor of the B operand, inputs, last accumulators) + the new GPU tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b
step() { name=$1; shift; "$@" > gpurun_out/r05b/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1500 gpurun_out/r05b/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step repro timeout -k 10 400 python tools/coresidency_repro.py pairdbg 40 --out gpurun_out/r05b/repro_pairdbg.txt
step newtests timeout -k 10 900 python -m pytest -x -q -s tests/test_quality_gate_gpu.py tests/test_coresidency_gpu.py tests/test_mirror_gpu.py tests/test_dist_gpu.py tests/test_bench_gpu.py
