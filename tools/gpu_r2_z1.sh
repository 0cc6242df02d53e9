#!/bin/bash
# round-2b final artifacts: C3 + C4 profiles, C2 / C5 bench lines, full GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z1
mkdir -p $R/$O
cd $R
timeout -k 10 600 bash tools/make_profiles.sh $O/c3 > $O/mp_c3.log 2>&1; echo "c3 rc=$?"
rm -rf $O/c3/stats $O/c3/FETCH_SIZE $O/c3/WRITE_SIZE $O/c3/MFMA
cd $R
timeout -k 10 600 bash tools/make_profiles.sh $O/c4 tools/bench_train.py --steps 8 --warmup 2 > $O/mp_c4.log 2>&1; echo "c4 rc=$?"
cd $R
python tools/train_trace_summary.py $(ls $O/c4/stats/*/*kernel_trace.csv | head -1) 6 > $O/c4/steady_table.md
rm -rf $O/c4/stats $O/c4/FETCH_SIZE $O/c4/WRITE_SIZE $O/c4/MFMA
timeout -k 10 200 python bench.py --cfg yolov4-tiny --batch 32 --steps 50 --warmup 5 --no-cpu-baseline > $O/c2.json 2> $O/c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --cfg yolov4-csp --batch 32 --half --steps 20 --warmup 3 --no-cpu-baseline > $O/c5.json 2> $O/c5.err; echo "c5 rc=$?"
python - <<'PY'
import json
for n in ("c3/run","c4/run","c2","c5"):
    try:
        d=json.loads(open('gpurun_out/r2z1/%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel'] if 'kernel' in d['roofline'] else '', round(d['roofline']['frac'],3), d.get('frac_of_fp32_mfma_roofline'))
    except Exception as e: print(n,'ERR',e)
PY
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "pytest all rc=$?"
tail -6 $O/test_all.log | cut -c1-300
