#!/bin/bash
# round 5, job d: paired-workgroup failure rate under four epilogue variants (operator-level stage check, 40 frames = 29 800 chunks each)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05d
for v in pair pair_noslp pair_pin pair_wait; do
  timeout -k 10 300 python tools/coresidency_stage.py $v 40 --out gpurun_out/r05d/stage_$v.txt > gpurun_out/r05d/stage_$v.log 2>&1
  rc=$?; echo "$v rc=$rc: $(grep TOTAL gpurun_out/r05d/stage_$v.txt)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
