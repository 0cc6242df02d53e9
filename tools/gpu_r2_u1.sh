#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2u1
mkdir -p $R/$O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_net.py -q -m gpu -k "rehearsal" > $O/test.log 2>&1; echo "pytest rc=$?"
tail -12 $O/test.log | cut -c1-300
