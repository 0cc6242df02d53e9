#!/bin/bash
# round 3 call j: the new training tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3j
mkdir -p $R/$O
cd $R
timeout -k 10 1100 python -m pytest tests/test_gpu_train.py -m gpu -x -q -s -k "every_training_kernel_variant or collective_path or clip_clamps or parity_classes or c4_yolov4" > $O/tests.log 2>&1; echo "rc=$?"
grep -v "^$" $O/tests.log | grep "passed\|failed\|Error\|C4 \|shim\|ran through" | tail -30
