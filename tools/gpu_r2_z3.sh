#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z3
mkdir -p $R/$O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -15 $O/test_train.log | cut -c1-400
DK_DETERMINISTIC=1 timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_det.json 2> $O/train_det.err; echo "rc=$?"
timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train.json 2> $O/train.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("train_det","train"):
    try:
        d=json.loads(open('gpurun_out/r2z3/%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2), d['last_cost'])
    except Exception as e: print(n,'ERR',e)
PY
