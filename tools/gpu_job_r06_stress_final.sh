#!/bin/bash
# round 6, last: the co-residency stress (all phases) and the wide-call reproducer on the final library
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_stress_final
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 900 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step stress timeout -k 10 700 python tools/coresidency_stress.py --calls 100000 --out $O/coresidency_stress.json
step wide_repro timeout -k 10 300 python tools/wide_repro.py --calls 100000 --kinds 0 0 0
