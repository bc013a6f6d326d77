#!/bin/bash
# round-4 job B: kernel timeline of 1024-ray calls in the three shapes
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04b
for SH in wide narrow single; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04b/$SH -- python3 tools/small_call_trace.py --rays 1024 --calls 80 --shape $SH > gpurun_out/r04b/$SH.log 2>&1 || { tail -20 gpurun_out/r04b/$SH.log; exit 1; }
  echo "== $SH"; python3 tools/small_call_trace.py --parse gpurun_out/r04b/$SH
done
