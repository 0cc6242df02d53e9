#!/bin/bash
# soak: 300 training steps (ring wraps, second stream, prep launch) -- the cost must fall and stay finite
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z9
mkdir -p $R/$O
cd $R
timeout -k 10 300 python tools/bench_train.py --steps 300 --warmup 2 > $O/soak.json 2> $O/soak.err; echo "rc=$?"
timeout -k 10 300 python tools/bench_train.py --steps 3 --warmup 0 > $O/start.json 2> $O/start.err; echo "rc=$?"
python - <<'PY'
import json
a=json.loads(open('gpurun_out/r2z9/start.json').read().strip().splitlines()[-1])
b=json.loads(open('gpurun_out/r2z9/soak.json').read().strip().splitlines()[-1])
print('cost after 3+2 steps', a['last_cost'], 'after 300+ steps', b['last_cost'], 'rate', round(b['value'],1), round(b['ms_per_step'],2))
PY
