#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z2
mkdir -p $R/$O
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
DK_STAGE_THREADS=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_t1.json 2> $O/bench_t1.err; echo "bench rc=$?"
python - <<'PY'
import json
for n in ("bench","bench_t1"):
    d=json.loads(open('gpurun_out/r2z2/%s.json'%n).read().strip().splitlines()[-1])
    print(n, round(d['value'],1), round(d['e2e_images_per_sec'],1), round(d['e2e_u8_frames_to_boxes_images_per_sec'],1))
PY
timeout -k 10 200 python tools/train_layers.py > $O/layers.txt 2> $O/layers.err; echo "layers rc=$?"
tail -1 $O/layers.txt
timeout -k 10 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_harness.py -q -m gpu -x > $O/test.log 2>&1; echo "pytest rc=$?"
tail -4 $O/test.log | cut -c1-300
