#!/bin/bash
# round 6, job d: the scene-trained nets through the full-frame parity tests, the kappa scan, the bench legs
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1200 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step fullframe timeout -k 10 900 python -m pytest tests/test_fullframe_gpu.py -q -m gpu -s -k "scene or population or optimizer_trained"
step kappa_scan timeout -k 10 300 python tools/kappa_scan.py
step bench timeout -k 10 600 python bench.py
