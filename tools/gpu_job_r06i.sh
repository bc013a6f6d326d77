#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06i
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 2500 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step t timeout -k 10 600 python -m pytest tests/test_fullframe_gpu.py tests/test_shapes_gpu.py -q -m gpu -s -k "optimizer_trained_nets or full_frame_on_off"
