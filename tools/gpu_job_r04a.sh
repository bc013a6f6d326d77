#!/bin/bash
# round-4 job A: new launch shapes — parity tests, full-frame A/B, shard / small-call rehearsal
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_render_gpu.py tests/test_ops_gpu.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r04a_tests.log 2>&1 || { tail -30 gpurun_out/r04a_tests.log; exit 1; }
tail -3 gpurun_out/r04a_tests.log
python tools/perf_ab.py --rounds 5 --frames 10 --configs "lib=,shape=wide;lib=,shape=narrow;lib=" > gpurun_out/r04a_ab.log 2>&1 || { tail -30 gpurun_out/r04a_ab.log; exit 1; }
cat gpurun_out/r04a_ab.log
python tools/shard_scaling.py --shapes wide narrow single auto --stages --out gpurun_out/r04a_shards.json > /dev/null 2> gpurun_out/r04a_shards.err || { tail -30 gpurun_out/r04a_shards.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04a_shards.json'))
for k, v in d.items():
    print(k, {w: round(s['ms_slowest'], 4) for w, s in v['shards'].items()}, {c: round(x['ms_per_call'], 4) for c, x in v['calls'].items()}, v.get('bit_identical_to_one_call_frame'))
    for w, s in v['shards'].items():
        print('   shard', w, {a: round(b, 4) for a, b in s['ms'].get('rank0_stages', {}).items()})
    for c, x in v['calls'].items():
        print('   call', c, {a: round(b, 4) for a, b in x.get('stages', {}).items()})
PY
