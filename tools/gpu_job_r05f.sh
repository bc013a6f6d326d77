#!/bin/bash
# round 5, job f: evidence of the fix (paired build with / without packed fp32), the long stress of the shipped shapes, the full GPU suite, the bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05f
for v in "pair_pk 60" "pair 160"; do
  set -- $v
  timeout -k 10 500 python tools/coresidency_stage.py $1 $2 --out gpurun_out/r05f/stage_$1.txt > gpurun_out/r05f/stage_$1.log 2>&1
  rc=$?; echo "$1 rc=$rc: $(grep TOTAL gpurun_out/r05f/stage_$1.txt)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
timeout -k 10 400 python tools/coresidency_stress.py --calls 100000 --out gpurun_out/r05f/stress_shipped.json > gpurun_out/r05f/stress_shipped.log 2>&1; rc=$?
echo "stress rc=$rc"; tail -c 400 gpurun_out/r05f/stress_shipped.log; echo
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05f/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r05f/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 900 python bench.py > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; rc=$?
echo "bench rc=$rc"; python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r05f/bench.json') if l.startswith('{')][-1])
print({k: j[k] for k in ('value', 'ms_per_step', 'vs_baseline')}, {k: round(v['ms'], 4) for k, v in j['kernels'].items()}, j['roofline']['frac'])
print('two-pass', j['sampler_two_pass']['fraction'], 'optimizer', {k: j['weights_optimizer'].get(k) for k in ('ms_per_frame', 'hip_vs_eager_rgb_psnr_db', 'kernels_ms', 'sampler_two_pass')})
t = j.get('train', {})
print('train', {k: (v.get('ms_per_iter') if isinstance(v, dict) else v) for k, v in t.items()})
PY
