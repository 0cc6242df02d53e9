#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z5
mkdir -p $R/$O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "pytest all rc=$?"
tail -6 $O/test_all.log | cut -c1-300
DK_DETERMINISTIC=1 timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/train_det.json 2> $O/train_det.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2z5/train_det.json').read().strip().splitlines()[-1])
print('det', round(d['value'],1), round(d['ms_per_step'],2))
PY
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
