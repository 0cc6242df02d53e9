#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2f4
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -s -k "winograd" > $O/test_wino.log 2>&1; rc=$?; echo "pytest wino rc=$rc"
grep -E "passed|failed|Error" $O/test_wino.log | cut -c1-220 | tail -5
[ $rc -eq 0 ] || exit 1
DK_SWEEP_FILTER=k3s1 timeout -k 10 400 python tools/conv_sweep.py cfg/yolov4.cfg 16 5 > $O/sweep_k3s1.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_k3s1.json
tail -1 $O/sweep_k3s1.log
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "c4" > $O/test_c4.log 2>&1; echo "pytest c4 rc=$?"
grep -E "passed|failed|C4 |^E " $O/test_c4.log | cut -c1-400 | tail
