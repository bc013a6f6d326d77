#!/bin/bash
# round 6, job u: with the register-file reservation: the stress (all phases), the default bench line, the GPU suite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06u
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1200 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step stress timeout -k 10 600 python tools/coresidency_stress.py --calls 100000 --out $O/coresidency_stress.json
step bench timeout -k 10 400 python bench.py --no-train
step suite timeout -k 10 1100 python -m pytest tests -q -m gpu -x
