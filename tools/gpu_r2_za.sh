#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2za
mkdir -p $R/$O
cd $R
for t in 4 8 12; do
DK_STAGE_THREADS=$t timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_t$t.json 2> $O/bench_t$t.err; echo "rc=$?"
done
python - <<'PY'
import json
for t in (4,8,12):
    d=json.loads(open('gpurun_out/r2za/bench_t%d.json'%t).read().strip().splitlines()[-1])
    print(t, round(d['value'],1), round(d['e2e_images_per_sec'],1), round(d['e2e_u8_frames_to_boxes_images_per_sec'],1))
PY
