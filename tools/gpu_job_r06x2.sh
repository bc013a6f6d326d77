#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06x2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_shapes_gpu.py tests/test_ops_gpu.py tests/test_render_gpu.py tests/test_refine16_gpu.py -q -m gpu -x > $O/t.log 2>&1; rc=$?; echo rc=$rc; tail -5 $O/t.log
