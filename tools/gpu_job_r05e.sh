#!/bin/bash
# round 5, job e: micro-reproducer of the packed-fp32 co-execution hazard (both builds) + the paired renderer without packed fp32 instructions
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05e
timeout -k 10 200 pronerf_amd/lib/pkf32_coexec_probe 16384 > gpurun_out/r05e/probe_pk.json 2> gpurun_out/r05e/probe_pk.err; rc=$?; echo "probe pk rc=$rc"; cat gpurun_out/r05e/probe_pk.json; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 200 pronerf_amd/lib/pkf32_coexec_probe_nopk 16384 > gpurun_out/r05e/probe_nopk.json 2> gpurun_out/r05e/probe_nopk.err; rc=$?; echo "probe nopk rc=$rc"; cat gpurun_out/r05e/probe_nopk.json; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
for v in "pair_nopk 60" "pair_noslp 120"; do
  set -- $v
  timeout -k 10 400 python tools/coresidency_stage.py $1 $2 --out gpurun_out/r05e/stage_$1.txt > gpurun_out/r05e/stage_$1.log 2>&1
  rc=$?; echo "$1 rc=$rc: $(grep TOTAL gpurun_out/r05e/stage_$1.txt)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
