#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06x
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_refine16_gpu.py -q -m gpu -s -x > $O/t.log 2>&1; rc=$?; echo rc=$rc; grep "refine 16x16\] frame size" $O/t.log; tail -4 $O/t.log
