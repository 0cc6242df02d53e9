#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2f5
mkdir -p $R/$O
cd $R
for b in 1 2 4 8 16 7; do
DK_LIB=$R/build_abl/libdk_wabl$b.so DK_SWEEP_FILTER=k3s1 timeout -k 10 200 python tools/conv_sweep.py cfg/yolov4.cfg 16 5 > $O/sweep_wabl$b.log 2>&1 || exit 1
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_wabl$b.json
tail -1 $O/sweep_wabl$b.log
done
