#!/bin/bash
# Dev tool: stamped diagnostic builds of the Winograd kernel -> build_abl/libdk_wstamp[_ablN].so (select with DK_LIB=...)
# usage: tools/build_wstamp.sh ["1 2 3"]   (DK_WABL ablation bits, default: none)
set -e
R=$(cd $(dirname $0)/.. && pwd)
CS=$R/darknet_amd/csrc
make -s -C $CS -j8
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$CS/kernels -I$CS/host -Wno-unused-result -Wno-return-type-c-linkage -mllvm -pragma-unroll-threshold=200000 -Wno-pass-failed"
OTHERS=$(find $CS/build -name '*.o' | grep -v -e conv3x3_wino.o)
for b in 0 $1; do
  ( d=$R/build_abl/ws$b; mkdir -p $d
    /opt/rocm/bin/hipcc $FLAGS -DDK_WSTAMP=1 -DDK_WABL=$b -c $CS/kernels/conv3x3_wino.hip -o $d/conv3x3_wino.o
    suffix=""; [ "$b" != "0" ] && suffix="_abl$b"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_abl/libdk_wstamp$suffix.so $d/conv3x3_wino.o $OTHERS -ldl -lpthread
    echo built $b ) &
done
wait
