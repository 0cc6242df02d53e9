#!/bin/bash
# C4 (yolov4 608 b=8 train step) round-2b artifacts + the two new train tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2x1
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "derived_weights or parity or stopbackward or train_networks" > $O/test.log 2>&1; echo "pytest rc=$?"
tail -4 $O/test.log | cut -c1-300
timeout -k 10 900 bash tools/make_profiles.sh $O/c4 tools/bench_train.py --steps 8 --warmup 2 > $O/mp.log 2>&1; echo "make_profiles rc=$?"
python tools/train_trace_summary.py $(ls $O/c4/stats/*/*kernel_trace.csv | head -1) 6 > $O/c4/steady_table.md
rm -rf $O/c4/stats $O/c4/FETCH_SIZE $O/c4/WRITE_SIZE $O/c4/MFMA
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2x1/c4/run.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],2), d['frac_of_fp32_mfma_roofline'], d['roofline']['kernel'], d['roofline']['frac'])
PY
head -3 $O/c4/steady_table.md; tail -9 $O/c4/steady_table.md
