#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2z7
mkdir -p $R/$O
cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python tools/bench_train.py --steps 10 --warmup 2 > $O/$n.json 2> $O/$n.err
}
run base A=1
run b512 DK_WGRAD_BLOCKS=512 DK_WGRAD_BLOCKS_1X1=512
run b2048 DK_WGRAD_BLOCKS=2048 DK_WGRAD_BLOCKS_1X1=2048
run b256 DK_WGRAD_BLOCKS=256 DK_WGRAD_BLOCKS_1X1=256
run base2 A=1
python - <<'PY'
import json
for n in ("base","b512","b2048","b256","base2"):
    try:
        d=json.loads(open('gpurun_out/r2z7/%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d['value'],1), round(d['ms_per_step'],2))
    except Exception as e: print(n,'ERR',e)
PY
