#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3f
mkdir -p $R/$O
cd $R
for sh in "16 128 76 76 128" "16 32 304 304 64"; do
  DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py $sh wino_64x64 >> $O/stamps.txt 2>&1; echo "rc=$?"
done
grep -v "^  \|stage period\|zero barrier\|^barrier" $O/stamps.txt
