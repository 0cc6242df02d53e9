#!/bin/bash
# Round-end artifacts in one GPU call: the five profile directories of profiles/roundN_* from the CURRENT sources.
# usage (through gpurun): bash tools/final_profiles.sh <name> [c3 c2 c5 c5x c4]   -> gpurun_out/<name>/<tag>/
# (c4: run tools/bench_train.py once before, in the same call -- the first training run on a fresh box is ~4 % slower)
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=$1; shift
TAGS=${@:-c3 c2 c5 c5x c4}
cd $R
for t in $TAGS; do
  case $t in
    c3) args="" ;;
    c2) args="bench.py --cfg yolov4-tiny --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs" ;;
    c5) args="bench.py --cfg yolov4-csp --batch 32 --half --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs" ;;
    c5x) args="bench.py --cfg yolov4x-mish --batch 32 --half --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs" ;;
    c4) args="tools/bench_train.py --steps 5 --warmup 2" ;;
  esac
  timeout -k 10 1000 bash tools/make_profiles.sh gpurun_out/$NAME/$t $args > gpurun_out/$NAME.$t.log 2>&1
  rc=$?
  echo "[$t] rc=$rc $(python3 -c "import json;d=json.loads(open('gpurun_out/$NAME/$t/run.json').read().strip().splitlines()[-1]);print(d.get('value'), d.get('lib_sha16'))" 2>&1 | tail -1)"
  [ $rc -ne 0 ] && exit $rc
  # keep only what is committed (the raw rocprofv3 directories exceed the merge limit)
  rm -rf gpurun_out/$NAME/$t/stats gpurun_out/$NAME/$t/FETCH_SIZE gpurun_out/$NAME/$t/WRITE_SIZE gpurun_out/$NAME/$t/MFMA
done
exit 0
