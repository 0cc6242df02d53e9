#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for mc in 1024 256 0; do echo "MAXC=$mc"; DK_WINO_MAXC=$mc timeout -k 10 400 python tools/wino_margin.py yolov4 yolov4-csp 2>/dev/null | grep worst; done
