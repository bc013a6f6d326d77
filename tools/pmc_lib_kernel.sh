#!/bin/bash
# as pmc_train_kernel.sh with a library variant:  bash tools/pmc_lib_kernel.sh <workload> <tag> <filter> <lib or ""> "<counters>" ...
set -eo pipefail
WLD=${1:?workload}; TAG=${2:?tag}; FILT=${3:?filter}; LIB=$4; shift 4
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "$@"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 tools/train_iter.py --workload $WLD --iters 3 --warmup 1 ${LIB:+--lib $LIB} > "$OUT/p$i.json"
  python3 tools/pmc_kernels.py "$OUT/p$i" "$FILT"
  rm -rf "$OUT/p$i"
  i=$((i+1))
done
