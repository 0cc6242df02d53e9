#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2i1
mkdir -p $R/$O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_extra.py -q -m gpu > $O/test_train.log 2>&1; echo "pytest train rc=$?"
tail -4 $O/test_train.log
timeout -k 10 300 python tools/bench_train.py --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; echo "bench_train rc=$?"
cut -c1-330 $O/train.json
