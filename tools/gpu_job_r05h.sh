#!/bin/bash
# round 5, job h: the two-pass sampler at kappa = 2: full-frame parity on the seven weight sets, the kappa scan, the frame's stage times
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05h
timeout -k 10 900 python -m pytest tests/test_fullframe_gpu.py tests/test_render_gpu.py tests/test_ops_gpu.py tests/test_soak_gpu.py -x -q -s > gpurun_out/r05h/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r05h/pytest.log; grep "second pass\|mismatch" gpurun_out/r05h/pytest.log | cut -c1-260 | head -30
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python tools/kappa_scan.py > gpurun_out/r05h/kappa_scan.txt 2>&1; rc=$?; echo "kappa_scan rc=$rc"; cut -c1-400 gpurun_out/r05h/kappa_scan.txt
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python tools/perf_ab.py --rounds 7 --frames 10 --configs "lib=;lib=pk" > gpurun_out/r05h/ab.txt 2>&1; cat gpurun_out/r05h/ab.txt
