#!/bin/bash
# round 3 call c: Winograd schedules -- parity of every configuration, then the per-shape sweep (one process)
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3c
mkdir -p $R/$O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "winograd or clip_clamps" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/tests.log
DK_SWEEP_FILTER=k3s1 timeout -k 10 600 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep.log 2>&1; echo "sweep rc=$?"
cut -c1-60,200-400 $O/sweep.log | tail -15
