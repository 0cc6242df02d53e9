#!/bin/bash
# round 3 call e: Winograd: 16-byte stores + complementary schedule -- parity, per-shape sweep, stamps
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3e
mkdir -p $R/$O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "winograd or clip_clamps" > $O/tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/tests.log
DK_SWEEP_FILTER=k3s1 timeout -k 10 600 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep.log 2>&1; echo "sweep rc=$?"
cut -c1-60,200-400 $O/sweep.log | tail -15
for cfg in wino_64x64 wino_64x64_pipe_compl; do
for sh in "16 128 76 76 128" "16 512 19 19 1024"; do
  DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py $sh $cfg >> $O/stamps.txt 2>&1; echo "rc=$?"
done; done
cat $O/stamps.txt
