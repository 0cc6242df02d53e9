#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2v8
mkdir -p $R/$O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_train.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -15 $O/test_train.log | cut -c1-300
timeout -k 10 200 python tools/train_layers.py > $O/layers.txt 2> $O/layers.err; echo "layers rc=$?"
grep "s2" $O/layers.txt; tail -1 $O/layers.txt
timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 > $O/train.json 2> $O/train.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2v8/train.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['all_conv_kernels'])
PY
