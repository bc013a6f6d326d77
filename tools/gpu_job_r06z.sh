#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06z
mkdir -p $O
timeout -k 10 400 python tools/perf_ab.py --rounds 7 --frames 10 --configs "lib=;lib=r5;lib=fixn" > $O/ab.txt 2>&1; echo rc=$?; tail -30 $O/ab.txt
