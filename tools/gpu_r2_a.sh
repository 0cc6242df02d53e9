#!/bin/bash
# round 2, GPU session A: new 1x1 LDS-DMA kernel -- parity, sweep, PMC stall breakdown old vs new
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2a
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dma1x1 or conv_forward_vs" > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
tail -3 $O/test.log
DK_SWEEP_FILTER=k1 timeout -k 10 300 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep_k1.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep_k1.json
tail -40 $O/sweep_k1.log
for cfg in 3 16 18; do
  timeout -k 10 200 bash tools/pmc_one.sh $O/pmc_128x76_c$cfg 16 128 76 76 128 1 1 0 17 $cfg > $O/pmc_128x76_c$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
for cfg in 10 16 18; do
  timeout -k 10 200 bash tools/pmc_one.sh $O/pmc_512x38_c$cfg 16 512 38 38 256 1 1 0 8 $cfg > $O/pmc_512x38_c$cfg.log 2>&1; echo "pmc $cfg rc=$?"
done
tail -30 $O/pmc_*/summary.txt
