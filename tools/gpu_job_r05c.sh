#!/bin/bash
# round 5, job c: which stage is the origin of the paired-workgroup wrong rows (operator-level chunks, every intermediate compared) + VALU price list
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05c
step() { name=$1; shift; "$@" > gpurun_out/r05c/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 3000 gpurun_out/r05c/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step valu timeout -k 10 120 pronerf_amd/lib/valu_rate_probe
step stage timeout -k 10 600 python tools/coresidency_stage.py pairdbg 40 --out gpurun_out/r05c/stage_pair.txt
