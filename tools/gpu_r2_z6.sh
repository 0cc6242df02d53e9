#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z6
mkdir -p $R/$O
cd $R
for rep in 1 2; do
DK_LIB=$R/build_abl/libdk_prev.so timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_prev$rep.json 2> $O/train_prev$rep.err; echo "rc=$?"
timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_low$rep.json 2> $O/train_low$rep.err; echo "rc=$?"
DK_WGRAD_PRIO=0 timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_high$rep.json 2> $O/train_high$rep.err; echo "rc=$?"
done
python - <<'PY'
import json
for n in ("train_prev1","train_low1","train_high1","train_prev2","train_low2","train_high2"):
    try:
        d=json.loads(open('gpurun_out/r2z6/%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2))
    except Exception as e: print(n,'ERR',e)
PY
