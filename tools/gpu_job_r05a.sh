#!/bin/bash
# round 5, job a: reproduce + diagnose the co-residency hazard with the paired debug build; placement evidence and the long stress on the shipped shapes
# (a step that was killed at its time limit ends the job: no further GPU step behind a hung one)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05a
step() { name=$1; shift; "$@" > gpurun_out/r05a/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 700 gpurun_out/r05a/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step repro timeout -k 10 300 python tools/coresidency_repro.py pairdbg 8 --out gpurun_out/r05a/repro_pairdbg.txt
step stress_dbg timeout -k 10 300 python tools/coresidency_stress.py --calls 20000 --variant dbg --out gpurun_out/r05a/stress_dbg.json
step stress_shipped timeout -k 10 400 python tools/coresidency_stress.py --calls 100000 --out gpurun_out/r05a/stress_shipped.json
