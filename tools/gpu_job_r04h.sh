#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_fullframe_gpu.py tests/test_ops_gpu.py -m gpu -x -q -s > gpurun_out/r04h_tests.log 2>&1 || { grep -v "^$" gpurun_out/r04h_tests.log | tail -60; exit 1; }
grep "\[full frame" gpurun_out/r04h_tests.log | cut -c1-400; tail -3 gpurun_out/r04h_tests.log
python tools/kappa_scan.py > gpurun_out/r04h_kappa.log 2>&1 || { tail -20 gpurun_out/r04h_kappa.log; exit 1; }
cat gpurun_out/r04h_kappa.log
