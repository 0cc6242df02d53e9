#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2k3
mkdir -p $R/$O
cd $R
for t in 0 1 2 3; do
DK_WGRAD_TILE=$t timeout -k 10 200 python tools/bench_train.py --steps 5 --warmup 2 > $O/train_$t.json 2> $O/train_$t.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r2k3/train_$t.json').read().strip().splitlines()[-1])
print($t, round(d['value'],1), round(d['ms_per_step'],2), [(k['kernel'], round(k['ms_per_step'],2), round(k['tflops'],1)) for k in d['roofline']['kernels'] if 'wgrad' in k['kernel']])
PY
done
S=$(date +%s); timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2k3/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['e2e_u8_frames_to_boxes_images_per_sec'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'])
PY
