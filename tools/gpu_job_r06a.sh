#!/bin/bash
# round 6, job a: the ADVICE tests (wide weight-gradient tiles, mm_input slices) + train the fixture nets on the geometrically consistent scene
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a
mkdir -p $O
step() { name=$1; shift; "$@" > $O/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -c 900 $O/$name.log; echo; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step tests timeout -k 10 500 python -m pytest tests/test_train_gpu.py tests/test_mirror_gpu.py -x -q -m gpu -k "weight_gradient or reference_driver"
step train timeout -k 10 500 python tools/make_trained_fixture.py --scene consistent --stage1 ${S1:-20000} --stage2 ${S2:-20000} --out $O/trained_scene3d.npz
