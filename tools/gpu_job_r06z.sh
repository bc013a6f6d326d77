#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06z
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_render_gpu.py tests/test_shapes_gpu.py -q -m gpu -x > $O/t.log 2>&1; rc=$?; echo rc=$rc; tail -5 $O/t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/perf_ab.py --rounds 7 --frames 10 --configs "lib=;lib=prev;lib=nopf;lib=,nerf=f16;lib=prev,nerf=f16" > $O/ab.txt 2>&1; echo rc=$?; tail -30 $O/ab.txt
