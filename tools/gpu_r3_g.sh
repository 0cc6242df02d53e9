#!/bin/bash
# round 3 call g: epilogue loads hoisted (Winograd + shared epilogue): op parity, full per-shape sweep, stamps, bench
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3g
mkdir -p $R/$O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_half.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -5 $O/tests.log
timeout -k 10 900 python tools/conv_sweep.py cfg/yolov4.cfg 16 10 > $O/sweep.log 2>&1; echo "sweep rc=$?"
tail -3 $O/sweep.log | cut -c1-300
DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py 16 128 76 76 128 wino_64x64 > $O/stamps.txt 2>&1; echo "rc=$?"
grep -v "^  \|zero barrier\|^barrier" $O/stamps.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['roofline']['kernels']: print(k)
PY
