#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2k4
mkdir -p $R/$O
cd $R
for t in 1024 512 256 2048; do
DK_WGRAD_BLOCKS_1X1=$t timeout -k 10 200 python tools/bench_train.py --steps 5 --warmup 2 > $O/train_$t.json 2> $O/train_$t.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r2k4/train_$t.json').read().strip().splitlines()[-1])
print($t, round(d['value'],1), round(d['ms_per_step'],2), [(k['kernel'], round(k['ms_per_step'],2), round(k['tflops'],1)) for k in d['roofline']['kernels'] if 'wgrad' in k['kernel']])
PY
done
