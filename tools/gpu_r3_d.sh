#!/bin/bash
# round 3 call d: phase timeline of the Winograd kernel (stamped diagnostic build) on four shapes
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3d
mkdir -p $R/$O
cd $R
for sh in "16 128 76 76 128" "16 256 38 38 256" "16 512 19 19 1024" "16 64 152 152 64"; do
  DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py $sh >> $O/stamps.txt 2>&1; echo "rc=$?"
done
cat $O/stamps.txt
