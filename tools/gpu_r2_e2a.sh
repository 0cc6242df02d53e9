#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2e2
mkdir -p $R/$O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_net.py -q -m gpu -k "c4 or tiny_train or nms or train_networks or two_replicas" > $O/test.log 2>&1; echo "pytest rc=$?" | tee -a $O/test.log
grep -E "passed|failed|FAILED|C4|worst|fp64" $O/test.log | tail -12
timeout -k 10 330 bash tools/make_profiles.sh $O/c3 bench.py --steps 20 --warmup 3 --no-cpu-baseline; echo "profiles c3 rc=$?"
cat $R/$O/c3/run.json | cut -c1-400
cat $R/$O/c3/kernel_table.md | head -30
timeout -k 10 240 bash tools/make_profiles.sh $O/c2 bench.py --cfg yolov4-tiny --batch 32 --steps 30 --warmup 5 --no-cpu-baseline; echo "profiles c2 rc=$?"
cat $R/$O/c2/run.json | cut -c1-300
