#!/bin/bash
# round 2, GPU session C: traffic-drop diagnostics on the 1x1 layers + the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2c
mkdir -p $R/$O
cd $R
for drop in 0 3 1 2; do
DK_DEBUG_DROP=$drop DK_SWEEP_FILTER=k1 timeout -k 10 200 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep_k1_drop$drop.log 2>&1; echo "sweep drop $drop rc=$?"
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_k1_drop$drop.json
done
DK_DEBUG_DROP=3 DK_SWEEP_FILTER=k3s1 timeout -k 10 200 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep_k3_drop3.log 2>&1; echo "sweep k3 drop rc=$?"
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_k3_drop3.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
tail -15 $O/test.log
