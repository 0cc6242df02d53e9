#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2q1
mkdir -p $R/$O
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "pytest all rc=$?"
tail -5 $O/test_all.log
timeout -k 10 300 bash tools/make_profiles.sh $O/c4 tools/bench_train.py --steps 5 --warmup 2; echo "profiles c4 rc=$?"
cut -c1-200 $R/$O/c4/run.json
