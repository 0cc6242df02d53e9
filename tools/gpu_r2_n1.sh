#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2n1
mkdir -p $R/$O
cd $R
for st in 0 2 4 8; do
DK_STAGGER=$st DK_SWEEP_FILTER=k1 timeout -k 10 300 python tools/conv_sweep.py cfg/yolov4.cfg 16 5 > $O/sweep_st$st.log 2>&1 || exit 1
echo "stagger $st: $(tail -1 $O/sweep_st$st.log)"
done
