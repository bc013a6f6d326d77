#!/bin/bash
# On the MI355X box: kernel-trace statistics of one training workload.   usage: bash tools/kstats.sh <workload> <tag> [train_iter.py args]
set -eo pipefail
WLD=${1:?workload}; TAG=${2:?tag}; shift 2
OUT=$PWD/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 tools/train_iter.py --workload $WLD --iters 10 --warmup 2 "$@" > "$OUT/run.json"
F=$(find "$OUT/raw" -name '*kernel_stats.csv' | head -1)
cp "$F" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-70s calls %4s avg us %9.1f  total ms %8.3f' % (r['Name'].replace('(anonymous namespace)::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, int(r['TotalDurationNs']) / 1e6))
PY
