#!/bin/bash
# One parameterised driver for the GPU-box runs of this repository (replaces the per-experiment scripts of rounds 2-3).
# usage (through gpurun):  gpurun --timeout 1200 -- 'bash tools/gpu_run.sh <name> <step> [<step> ...]'
# Output goes to gpurun_out/<name>/.  Steps (each may carry arguments after a colon, '+' separated):
#   tests[:<pytest -k expression>]      pytest -m gpu (whole suite without an expression)
#   smoke                               __graft_entry__.smoke()
#   bench[:<bench.py args>]             default bench line (bench.json / bench.err), timed
#   train[:<bench_train.py args>]       tools/bench_train.py
#   sweep:<cfg>+<batch>[+<filter>[+<toggle>]]   tools/conv_sweep.py (every configuration on every conv shape; filter e.g. k3s1;
#                                       toggle = an exported int setter measured at 0 and 1 in one process)
#   stamps:<b>+<c>+<h>+<w>+<n>[+<config>+<act>[+res]]   tools/wino_stamps.py on a -DDK_WSTAMP build (tools/build_wstamp.sh first)
#   gstamps:<b>+<c>+<h>+<w>+<n>+<size>+<stride>+<pad>+<act>+<config>   tools/gather_stamps.py (tools/build_stamp.sh conv_igemm DK_GSTAMP first)
#   act:<b>+<c>+<h>+<w>+<n>+<config>     tools/act_cost.py (LINEAR / LEAKY / MISH epilogue of one 1x1 shape)
#   bn                                  tools/bn_bench.py
#   tool:<script>[+args]                 python tools/<script>.py args  (output <script>.txt)
#   profiles[:<tag>]                    tools/make_profiles.sh gpurun_out/<name>/<tag> (rocprofv3 stats + PMC passes of bench.py)
# A failing step stops the run (no GPU step is started after a failed or timed-out one).
R=${GRAFT_REPO_ROOT:-$PWD}
NAME=$1; shift
O=gpurun_out/$NAME
mkdir -p $R/$O
cd $R
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$step" != "$kind" ] && arg=${step#*:}
  arg=${arg//+/ }
  case $kind in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "$arg" > $O/tests.log 2>&1
      else timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; fi
      rc=$?; tail -6 $O/tests.log ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; tail -1 $O/smoke.log ;;
    bench) ( time timeout -k 10 580 python bench.py $arg > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; rc=$?
      grep real $O/bench_time.txt; tail -c 600 $O/bench.json ;;
    train) timeout -k 10 580 python tools/bench_train.py $arg > $O/train.json 2> $O/train.err; rc=$?; tail -c 600 $O/train.json ;;
    sweep) set -- $arg; DK_SWEEP_FILTER=$3 DK_SWEEP_TOGGLE=$4 timeout -k 10 900 python tools/conv_sweep.py cfg/$1.cfg $2 10 > $O/sweep.log 2>&1; rc=$?; grep -v "^L\|^configs" $O/sweep.log | tail -60 | cut -c1-200 ;;
    stamps) DK_LIB=$R/build_abl/libdk_wstamp.so timeout -k 10 200 python tools/wino_stamps.py $arg >> $O/stamps.txt 2>&1; rc=$?; tail -20 $O/stamps.txt ;;
    gstamps) DK_LIB=$R/build_abl/libdk_conv_igemm_stamp.so timeout -k 10 200 python tools/gather_stamps.py $arg >> $O/gstamps.txt 2>&1; rc=$?; tail -12 $O/gstamps.txt ;;
    act) timeout -k 10 200 python tools/act_cost.py $arg >> $O/act_cost.txt 2>&1; rc=$?; tail -2 $O/act_cost.txt ;;
    bn) timeout -k 10 300 python tools/bn_bench.py 20 > $O/bn_bench.txt 2>&1; rc=$?; cat $O/bn_bench.txt ;;
    tool) set -- $arg; t=$1; shift; timeout -k 10 300 python tools/$t.py "$@" >> $O/$t.txt 2>&1; rc=$?; tail -12 $O/$t.txt ;;
    profiles) bash tools/make_profiles.sh $O/${arg:-c3} > $O/profiles.log 2>&1; rc=$?; tail -12 $O/profiles.log ;;
    *) echo "unknown step $kind"; rc=2 ;;
  esac
  echo "[$kind] rc=$rc"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
