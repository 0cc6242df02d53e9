#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3h
mkdir -p $R/$O
cd $R
DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py 16 128 76 76 128 wino_64x64 17 res >> $O/stamps.txt 2>&1; echo "rc=$?"
DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py 16 128 76 76 128 wino_64x64 17 >> $O/stamps.txt 2>&1; echo "rc=$?"
DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py 16 512 19 19 1024 wino_64x64 8 >> $O/stamps.txt 2>&1; echo "rc=$?"
DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py 16 256 38 38 256 wino_64x64 17 res >> $O/stamps.txt 2>&1; echo "rc=$?"
grep -v "^  \|zero barrier\|^barrier" $O/stamps.txt
