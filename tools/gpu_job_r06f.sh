#!/bin/bash
# round 6, job f: free shape parameters — parity on off-Fern shapes, then the default bench line (no regression from the run-time layer counts)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1800 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step shapes timeout -k 10 900 python -m pytest tests/test_shapes_gpu.py -q -m gpu -s -x
step bench timeout -k 10 600 python bench.py
