#!/bin/bash
# Runs on the MI355X box (gpurun): un-profiled bench line, kernel-trace stats, then one --pmc pass per counter set.
# usage: bash tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:?tag}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"      # a second run into the same tag must not leave the first one's CSVs beside its own
export TMPDIR=/tmp
python3 -c "from pronerf_amd import build; print(build._digest('inference'))" > "$OUT/csrc_digest.txt"      # what bench.py checks before quoting this profile
python3 bench.py --steps 20 --warmup 10 --no-train > "$OUT/bench.json"
python3 -c "import json,sys; j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); json.dump(j['shard_rehearsal'], open('$OUT/shard_rehearsal.json','w'), indent=1)"
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train --no-shard-rehearsal --no-optimizer-weights --steady-seconds 0 > "$OUT/bench_under_rocprof.json"
echo "stats done"
LIGHT="--no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train --no-shard-rehearsal --no-optimizer-weights --steady-seconds 0"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_')"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 bench.py --steps 3 --warmup 1 $LIGHT > "$D.json"
  echo "pmc $C done"
done
# where the waves' cycles go (round 4: "the counter that binds"): issue stalls vs parked vs active, LDS issue stalls, LDS array cycles.  Optional
# passes: a counter name this rocprofv3 does not know must not lose the passes above.
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-80)"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 bench.py --steps 3 --warmup 1 $LIGHT > "$D.json" 2> "$D.err" || echo "optional pass failed: $C"
  echo "pmc $C done"
done
find "$OUT" -name "*.csv" | head -30
