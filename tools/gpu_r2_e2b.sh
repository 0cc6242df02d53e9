#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2e2
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "c4" > $O/test_c4.log 2>&1; echo "pytest rc=$?" | tee -a $O/test_c4.log
grep -E "passed|failed|FAILED|C4|worst|fp64" $O/test_c4.log | tail -12
timeout -k 10 420 bash tools/make_profiles.sh $O/c5 bench.py --cfg yolov4-csp --batch 32 --half --steps 20 --warmup 3 --no-cpu-baseline; echo "profiles c5 rc=$?"
cut -c1-300 $R/$O/c5/run.json
head -24 $R/$O/c5/kernel_table.md
timeout -k 10 420 bash tools/make_profiles.sh $O/c4 tools/bench_train.py --steps 5 --warmup 2; echo "profiles c4 rc=$?"
cut -c1-600 $R/$O/c4/run.json
head -30 $R/$O/c4/kernel_table.md
