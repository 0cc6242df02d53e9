#!/bin/bash
# round 2, GPU session B: persistent loader/consumer 1x1 kernel -- parity, sweep, bench
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2b
mkdir -p $R/$O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dma1x1 or conv_forward_vs or direct3x3" > $O/test.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/test.log
tail -5 $O/test.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep.log 2>&1; echo "sweep rc=$?"
cp gpurun_out/conv_sweep_yolov4.cfg_b16.json $O/sweep.json
grep -E "^L|^total" $O/sweep.log | cut -c1-60,70-400 | tail -60
for bpc in 1 3; do
DK_PERS_BPC=$bpc DK_SWEEP_FILTER=k1 timeout -k 10 300 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep_k1_bpc$bpc.log 2>&1; echo "sweep bpc $bpc rc=$?"
done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
