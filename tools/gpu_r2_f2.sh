#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2f2
mkdir -p $R/$O
cd $R
for d in 1 4 5; do
DK_DEBUG_DROP=$d DK_SWEEP_FILTER=k3s1 timeout -k 10 200 python tools/conv_sweep.py cfg/yolov4.cfg 16 5 > $O/sweep_drop$d.log 2>&1 || exit 1
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_drop$d.json
tail -1 $O/sweep_drop$d.log
done
