#!/bin/bash
# round 6, final check: the driver's sequence — build is shipped; smoke, the full GPU suite, the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06_final
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_final/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -3 gpurun_out/r06_final/smoke.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_final/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r06_final/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 900 python bench.py > gpurun_out/r06_final/bench.json 2> gpurun_out/r06_final/bench.err; rc=$?; echo "bench rc=$rc"
python3 - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r06_final/bench.json') if l.startswith('{')][-1])
print({k: j[k] for k in ('value', 'ms_per_step', 'vs_baseline', 'frame_sha256')}, {k: round(v['ms'], 4) for k, v in j['kernels'].items()})
print('roofline', {k: j['roofline'][k] for k in ('frac', 'achieved', 'traffic')}, 'cpu', j['cpu_baseline']['value'], 'two-pass', round(j['sampler_two_pass']['fraction'], 4))
print('optimizer', j['weights_optimizer']['ms_per_frame'], j['weights_optimizer']['hip_vs_eager_rgb_psnr_db']); print('scene3d', {k: j['weights_scene3d'].get(k) for k in ('ms_per_frame', 'psnr_vs_ground_truth_db', 'quality_preset_ms_per_frame', 'hip_vs_eager_rgb_psnr_db', 'calibrated')})
t = j['train']
print('train', {k: (round(v['ms'], 4) if isinstance(v, dict) and 'ms' in v else None) for k, v in t.items()})
PY
