#!/bin/bash
# round 6, job s: the row-checking stress and the determinism soak on the round-6 kernels (run-time layer counts, templated refine head)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1500 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step stress timeout -k 10 500 python tools/coresidency_stress.py --calls 40000 --out $O/coresidency_stress.json
step soak timeout -k 10 400 python tools/soak.py --frames 3000 --chunked 60
