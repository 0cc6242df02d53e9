#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2s1
mkdir -p $R/$O
cd $R
for ring in 3 4; do
DK_WINO_RING=$ring DK_SWEEP_FILTER=k3s1 timeout -k 10 300 python tools/conv_sweep.py cfg/yolov4.cfg 16 5 > $O/sweep_ring$ring.log 2>&1 || exit 1
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_ring$ring.json
echo "ring $ring: $(tail -1 $O/sweep_ring$ring.log)"
done
