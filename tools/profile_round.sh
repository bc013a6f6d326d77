#!/bin/bash
# Runs on the MI355X box (gpurun): un-profiled bench line, kernel-trace stats, then one --pmc pass per counter set.
# usage: bash tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -eo pipefail
TAG=${1:?tag}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "from pronerf_amd import build; print(build._digest())" > "$OUT/csrc_digest.txt"      # what bench.py checks before quoting this profile
python3 bench.py --steps 10 --warmup 2 --no-chunked --no-train > "$OUT/bench.json"
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train > "$OUT/bench_under_rocprof.json"
echo "stats done"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  D="$OUT/pmc_$(echo $C | tr ' ' '_')"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$D" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-eager-baseline --no-sustained --no-chunked --no-variants --no-train > "$D.json"
  echo "pmc $C done"
done
find "$OUT" -name "*.csv" | head -30
