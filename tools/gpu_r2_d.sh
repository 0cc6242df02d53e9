#!/bin/bash
# round 2, GPU session D: ablation sweeps (what bounds the conv kernels), TrainNetworks debug, new tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2d
mkdir -p $R/$O
cd $R
for b in 0 1 2 4 6 8 16 30 31; do
  lib=$R/build_abl/libdk_abl$b.so
  [ $b -eq 0 ] && lib=$R/darknet_amd/libdarknet_amd.so
  for f in k1 k3s1; do
    DK_LIB=$lib DK_SWEEP_FILTER=$f timeout -k 10 120 python tools/conv_sweep.py cfg/yolov4.cfg 16 8 > $O/abl${b}_$f.log 2>&1; echo "abl $b $f rc=$?"
    cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/abl${b}_$f.json
  done
done
timeout -k 10 200 python tools/debug_trainnets.py > $O/debug_trainnets.log 2>&1; echo "debug rc=$?"; grep "^step" $O/debug_trainnets.log
timeout -k 10 600 python -m pytest tests/test_gpu_extra.py tests/test_gpu_net.py tests/test_gpu_ops.py -q -m gpu > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
tail -25 $O/test.log
