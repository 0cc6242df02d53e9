#!/bin/bash
# stage anatomy under ablation (stamped builds): 0 full, 1 no input DMA, 2 no filter DMA, 3 neither, 4 no transform, 7 only MFMA + fragment reads
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3i
mkdir -p $R/$O
cd $R
for a in "" _abl1 _abl2 _abl3 _abl4 _abl7; do
  echo "=== ablation '$a'" >> $O/stamps.txt
  for cfg in wino_64x64 wino_64x64_pipe_compl; do
  DK_LIB=$R/build_abl/libdk_wstamp$a.so timeout -k 10 200 python tools/wino_stamps.py 16 128 76 76 128 $cfg 17 >> $O/stamps.txt 2>&1; echo "rc=$?"
  done
done
grep "===\|stage period\|  \|whole loop\|TOTAL" $O/stamps.txt
