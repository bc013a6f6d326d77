#!/bin/bash
# round 6, job b: train the fixture nets on the consistent scene (with fine texture)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 900 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step train timeout -k 10 500 python tools/make_trained_fixture.py --scene consistent --stage1 ${S1:-20000} --stage2 ${S2:-20000} --out $O/trained_scene3d.npz
