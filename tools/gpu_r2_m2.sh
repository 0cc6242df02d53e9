#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2m2
mkdir -p $R/$O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k "adam or tiny_train or split" > $O/test.log 2>&1; echo "pytest rc=$?"
tail -30 $O/test.log | cut -c1-250
