#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2t1
mkdir -p $R/$O
cd $R
DK_WINO_MAXC=1024 timeout -k 10 600 python -m pytest tests/test_gpu_net.py -q -m gpu -k "golden or baseline" > $O/test_maxc.log 2>&1; echo "pytest maxc=1024 rc=$?"
tail -6 $O/test_maxc.log | cut -c1-250
DK_WINO_MAXC=1024 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_maxc.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2t1/bench_maxc.json').read().strip().splitlines()[-1])
print('maxc=1024:', d['value'], d['frac_of_fp32_mfma_roofline'])
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2t1/bench.json').read().strip().splitlines()[-1])
print('default:', d['value'], d['frac_of_fp32_mfma_roofline'])
for k in d['roofline']['kernels']: print(k)
PY
