#!/bin/bash
# round-4 job E: full GPU suite on the re-applied wide / narrow shapes + chunked streams; bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r04e_tests.log 2>&1 || { tail -40 gpurun_out/r04e_tests.log; exit 1; }
tail -3 gpurun_out/r04e_tests.log
python bench.py --steps 20 --warmup 10 --no-train > gpurun_out/r04e_bench.json 2> gpurun_out/r04e_bench.err || { tail -30 gpurun_out/r04e_bench.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r04e_bench.json').read().strip().splitlines()[-1])
print('ms/frame', j['ms_per_step'], 'steady', j.get('steady_state', {}).get('ms_per_frame'), 'kernels', {k: round(v['ms'], 4) for k, v in j['kernels'].items()})
print('chunked', {k: (round(v, 2) if isinstance(v, float) else v) for k, v in j['chunked_1024'].items()})
sr = j['shard_rehearsal']
print('shards', {w: (round(s['ms_slowest'], 4), round(s['speedup_bound'], 2)) for w, s in sr['shards'].items()}, {c: round(x['ms_per_call'], 4) for c, x in sr['calls'].items()}, sr.get('bit_identical_to_one_call_frame'))
print('traffic', j['roofline']['traffic'], j['roofline']['traffic_source'])
PY
