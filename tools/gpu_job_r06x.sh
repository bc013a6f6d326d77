#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06x
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_refine16_gpu.py -q -m gpu -s -x > $O/t.log 2>&1; rc=$?; echo rc=$rc; tail -30 $O/t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/perf_ab.py --rounds 7 --frames 10 --configs "lib=;lib=,refine=refine_16x16" > $O/ab.txt 2>&1; echo rc=$?; tail -30 $O/ab.txt
