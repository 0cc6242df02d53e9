#!/bin/bash
# round 3 call k: whole GPU suite, then the default bench line (with other_configs)
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r3k
mkdir -p $R/$O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
( time timeout -k 10 580 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; cat $O/bench_time.txt | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['frac_of_fp32_mfma_roofline'], d['e2e_images_per_sec'], d['e2e_u8_frames_to_boxes_images_per_sec'])
r=d['roofline']; print({k:r[k] for k in ('kernel','frac','frac_executed_mfma','traffic','traffic_source','algorithmic_bytes','traffic_ratio')})
print(json.dumps(d['other_configs'],indent=1)); print(d['cpu_baseline'])
PY
