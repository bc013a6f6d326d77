#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06y
mkdir -p $O
timeout -k 10 400 python tools/perf_ab.py --rounds 5 --frames 10 --configs "lib=;lib=,path=ops;lib=,refine=refine_16x16,path=ops;lib=,refine=refine_16x16" > $O/ab.txt 2>&1; echo rc=$?; tail -30 $O/ab.txt
