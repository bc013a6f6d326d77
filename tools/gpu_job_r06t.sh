#!/bin/bash
# round 6, job t: with every wide kernel's allocation forced to 240 registers: do the other stages fail beside the small kernel too
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06t
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; grep -c '"call"' $O/$name.log; tail -n 1 $O/$name.log | cut -c1-400; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step fixn240_sampler timeout -k 10 250 python tools/wide_repro.py --calls 200000 --kinds 0 0 0 --only sampler --variant fixn240
step fixn240_nerf timeout -k 10 250 python tools/wide_repro.py --calls 100000 --kinds 0 0 0 --only nerf --variant fixn240
step fixn240_sampler_split timeout -k 10 250 python tools/wide_repro.py --calls 100000 --kinds 0 0 0 --only sampler --sampler sampler_split --variant fixn240
