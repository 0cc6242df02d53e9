#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2g1
mkdir -p $R/$O
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-200 $O/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2g1/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['all_conv_kernels'])
for k in d['roofline']['kernels']: print(k)
PY
timeout -k 10 600 python -m pytest tests/test_gpu_net.py -q -m gpu -x > $O/test_net.log 2>&1; echo "pytest net rc=$?"
tail -5 $O/test_net.log
