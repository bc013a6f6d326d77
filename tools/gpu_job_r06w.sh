#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06w
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mirror_gpu.py tests/test_engine_gpu.py -q -m gpu -s -x > $O/mirror.log 2>&1; echo rc=$?; tail -5 $O/mirror.log; grep "frame driver on the scene\|pnrf_preset" $O/mirror.log | head
