#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2l1
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "pytest all rc=$?"
tail -5 $O/test_all.log
/usr/bin/time -v timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2l1/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['e2e_u8_frames_to_boxes_images_per_sec'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])
PY
