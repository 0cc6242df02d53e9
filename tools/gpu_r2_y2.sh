#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2y2
mkdir -p $R/$O
cd $R
timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train.json 2> $O/train.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2y2/train.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],2), d['last_cost'])
PY
timeout -k 10 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_harness.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -5 $O/test_train.log | cut -c1-300
