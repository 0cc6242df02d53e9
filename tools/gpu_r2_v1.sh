#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2v1
mkdir -p $R/$O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_train.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -5 $O/test_train.log | cut -c1-300
DK_TRAIN_TUNE=0 timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 > $O/train_notune.json 2> $O/train_notune.err; echo "rc=$?"
DK_TRAIN_WINO=0 timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 > $O/train_nowino.json 2> $O/train_nowino.err; echo "rc=$?"
timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 > $O/train_tune.json 2> $O/train_tune.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("notune","nowino","tune"):
    try:
        d=json.loads(open('gpurun_out/r2v1/train_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['all_conv_kernels'], d.get('train_tune'))
    except Exception as e: print(n, 'ERR', e)
PY
