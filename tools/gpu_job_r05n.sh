#!/bin/bash
# round 5, job n: the SHIPPED shapes beside a foreign bf16-MFMA kernel with 240 registers per wave — library as built (no packed fp32) and the same sources with packed fp32
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05n
timeout -k 10 400 python tools/coresidency_stress.py --calls 60000 --variant pk --out gpurun_out/r05n/stress_pk.json > gpurun_out/r05n/stress_pk.log 2>&1; rc=$?
echo "pk rc=$rc"; python3 -c "
import json; j=json.load(open('gpurun_out/r05n/stress_pk.json')); print({k:(v['calls' if 'calls' in v else 'frames'], v['rows_differ']) for k,v in j.items() if isinstance(v,dict) and 'rows_differ' in v})"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python tools/coresidency_stress.py --calls 100000 --out gpurun_out/r05n/stress_shipped.json > gpurun_out/r05n/stress_shipped.log 2>&1; rc=$?
echo "shipped rc=$rc"; python3 -c "
import json; j=json.load(open('gpurun_out/r05n/stress_shipped.json')); print({k:(v['calls' if 'calls' in v else 'frames'], v['rows_differ']) for k,v in j.items() if isinstance(v,dict) and 'rows_differ' in v})"
