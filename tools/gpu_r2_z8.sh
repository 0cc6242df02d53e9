#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z8
mkdir -p $R/$O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_extra.py tests/test_gpu_harness.py -q -m gpu > $O/test.log 2>&1; echo "pytest rc=$?"
tail -4 $O/test.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
