#!/bin/bash
# round 6, job c: quality gate on both scenes, both presets
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 900 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step gate timeout -k 10 600 python -m pytest tests/test_quality_gate_gpu.py -q -m gpu -s
