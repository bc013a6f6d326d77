#!/bin/bash
# round 6, job e: the mixed pair (ELU stages of frame i+1 beside the NeRF stage of frame i), timed; bench scene legs
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 1500 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step mixed timeout -k 10 400 python tools/mixed_pair_probe.py --out $O/mixed_pair.json
step mixed_scene timeout -k 10 400 python tools/mixed_pair_probe.py --weights scene3d --out $O/mixed_pair_scene3d.json
