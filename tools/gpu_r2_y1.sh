#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2y1
mkdir -p $R/$O
cd $R
DK_TRAIN_STREAMS=0 timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_s0.json 2> $O/train_s0.err; echo "rc=$?"
timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_s1.json 2> $O/train_s1.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("s0","s1"):
    try:
        d=json.loads(open('gpurun_out/r2y1/train_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2), d['last_cost'])
    except Exception as e: print(n,'ERR',e)
PY
timeout -k 10 400 python -m pytest tests/test_gpu_train.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -5 $O/test_train.log | cut -c1-300
