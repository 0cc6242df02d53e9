#!/bin/bash
# round 3 call b: the new parity tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3b
mkdir -p $R/$O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "distinct_images or yolov4x_mish or resize_then_device_nms or resize_with_dropout or clip_clamps or caller_owned or c5_csp_512" > $O/tests.log 2>&1; echo "rc=$?"
tail -30 $O/tests.log
