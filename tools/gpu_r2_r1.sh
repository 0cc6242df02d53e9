#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2r1
mkdir -p $R/$O
cd $R
DK_SWEEP_FILTER=k3s1 timeout -k 10 500 python tools/conv_sweep.py cfg/yolov4.cfg 8 4 > $O/sweep_b8.log 2>&1; echo rc=$?
tail -1 $O/sweep_b8.log
