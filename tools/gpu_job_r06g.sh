#!/bin/bash
# round 6, job g: the whole -m gpu suite at HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 2500 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step suite timeout -k 10 1150 python -m pytest tests -q -m gpu -x --durations=15
