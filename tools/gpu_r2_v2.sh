#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2v2
mkdir -p $R/$O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py tests/test_gpu_extra.py -q -m gpu -x > $O/test_train.log 2>&1; echo "pytest rc=$?"
tail -5 $O/test_train.log | cut -c1-300
timeout -k 10 200 python tools/bench_train.py --steps 8 --warmup 2 > $O/train_tune.json 2> $O/train_tune.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("tune",):
    try:
        d=json.loads(open('gpurun_out/r2v2/train_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['all_conv_kernels'])
    except Exception as e: print(n, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o train -- python3 $R/tools/bench_train.py --steps 8 --warmup 2 > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*/train_kernel_stats.csv $O/prof/train_kernel_stats.csv 2>/dev/null | head -1)
cp $f $O/kernel_stats.csv
find $O/prof -name "*.db" -delete; find $O/prof -name "*trace*" -delete
head -40 $O/kernel_stats.csv | cut -c1-200
