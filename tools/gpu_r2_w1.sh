#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2w1
mkdir -p $R/$O
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2w1/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['all_conv_kernels'])
for k in d['roofline']['kernels']: print(k)
PY
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "pytest all rc=$?"
tail -8 $O/test_all.log | cut -c1-300
