#!/bin/bash
# Runs on the MI355X box (gpurun): kernel trace + FETCH_SIZE / WRITE_SIZE passes of the training iterations at BASELINE size.
# Each counter is collected twice per workload, over A and B iterations; tools/train_pmc_summary.py takes (B - A) / (B_iters - A_iters),
# which removes the set-up kernels.   usage: bash tools/profile_train.sh <tag>    outputs under gpurun_out/prof_train_<tag>/
set -eo pipefail
TAG=${1:?tag}
OUT=$PWD/gpurun_out/prof_train_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"      # a second run into the same tag must not leave the first one's CSVs beside its own
export TMPDIR=/tmp
python3 -c "from pronerf_amd import build; print(build._digest('training'))" > "$OUT/csrc_digest.txt"      # what bench.py checks before quoting this profile
for WLD in stage2_iteration stage1_explore_64 stage1_explore_256; do
  python3 tools/train_iter.py --workload $WLD --iters 20 --warmup 3 > "$OUT/$WLD.json"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$WLD" -- python3 tools/train_iter.py --workload $WLD --iters 10 --warmup 2 > "$OUT/stats_$WLD.json"
  for C in FETCH_SIZE WRITE_SIZE; do
    for IT in 2 8; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${WLD}_${C}_$IT" -- python3 tools/train_iter.py --workload $WLD --iters $IT --warmup 0 > "$OUT/pmc_${WLD}_${C}_$IT.json"
    done
    echo "$WLD $C done"
  done
done
