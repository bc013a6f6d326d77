#!/bin/bash
# round 6, job t: wide fused kernels that own their SIMDs' register file (256 VGPRs per wave): the refine stage alone and the whole call beside foreign kind 0; the narrow shape too
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06t
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; grep -c '"call"' $O/$name.log; tail -n 1 $O/$name.log | cut -c1-400; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step own256_refine timeout -k 10 250 python tools/wide_repro.py --calls 300000 --kinds 0 0 0 --only refine
step v248_refine timeout -k 10 250 python tools/wide_repro.py --calls 300000 --kinds 0 0 0 --only refine --variant v248
step own256_full timeout -k 10 250 python tools/wide_repro.py --calls 150000 --kinds 0 0 0
step narrow_full timeout -k 10 250 python tools/wide_repro.py --calls 150000 --kinds 0 0 0 --shape narrow --chunk 1024
step narrow_refine timeout -k 10 250 python tools/wide_repro.py --calls 300000 --kinds 0 0 0 --shape narrow --chunk 1024 --only refine
