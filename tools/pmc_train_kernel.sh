#!/bin/bash
# On the MI355X box: counters of the training kernels whose name matches <filter>, one --pmc pass per counter group.
# usage: bash tools/pmc_train_kernel.sh <workload> <tag> <filter> "<counters>" ["<counters>" ...]
set -eo pipefail
WLD=${1:?workload}; TAG=${2:?tag}; FILT=${3:?filter}; shift 3
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "$@"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 tools/train_iter.py --workload $WLD --iters 3 --warmup 1 > "$OUT/p$i.json"
  python3 tools/pmc_kernels.py "$OUT/p$i" "$FILT"
  rm -rf "$OUT/p$i"
  i=$((i+1))
done
