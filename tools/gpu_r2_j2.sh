#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out/r2j2
mkdir -p $R/$O
cd $R
timeout -k 10 330 bash tools/make_profiles.sh $O/c3 bench.py --steps 20 --warmup 3 --no-cpu-baseline; echo "profiles c3 rc=$?"
cut -c1-160 $R/$O/c3/run.json
head -12 $R/$O/c3/kernel_table.md | cut -c1-160
timeout -k 10 240 bash tools/make_profiles.sh $O/c2 bench.py --cfg yolov4-tiny --batch 32 --steps 30 --warmup 5 --no-cpu-baseline; echo "profiles c2 rc=$?"
cut -c1-160 $R/$O/c2/run.json
timeout -k 10 300 bash tools/make_profiles.sh $O/c5 bench.py --cfg yolov4-csp --batch 32 --half --steps 20 --warmup 3 --no-cpu-baseline; echo "profiles c5 rc=$?"
cut -c1-160 $R/$O/c5/run.json
timeout -k 10 200 python bench.py --cfg yolov4-csp --batch 32 --steps 20 --warmup 3 --no-cpu-baseline > $O/c5_f32.json 2>/dev/null; cut -c1-160 $O/c5_f32.json
