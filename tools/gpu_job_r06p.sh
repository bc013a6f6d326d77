#!/bin/bash
# round 6, profiles: the default bench line, rocprofv3 kernel stats and --pmc passes (inference), then the training workloads
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06p
timeout -k 10 700 bash tools/profile_round.sh r06 > gpurun_out/r06p/profile_round.log 2>&1; rc=$?; echo "profile_round rc=$rc"; tail -5 gpurun_out/r06p/profile_round.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 450 bash tools/profile_train.sh r06 > gpurun_out/r06p/profile_train.log 2>&1; rc=$?; echo "profile_train rc=$rc"; tail -5 gpurun_out/r06p/profile_train.log
