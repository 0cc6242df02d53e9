#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2k2
mkdir -p $R/$O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_train.py -q -m gpu -k "conv_backward or tiny_train or split_train or real_train" > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout -k 10 200 python tools/bench_train.py --steps 5 --warmup 2 > $O/train.json 2> $O/train.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r2k2/train.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [(k['kernel'], round(k['ms_per_step'],2), round(k['tflops'],1)) for k in d['roofline']['kernels'] if 'wgrad' in k['kernel']])
PY
