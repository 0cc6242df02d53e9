#!/bin/bash
# Dev tool: diagnostic builds of the Winograd kernel with parts removed (DK_WABL bits, see
# conv3x3_wino.hip) -> build_abl/libdk_wabl<bits>.so; select with DK_LIB=...
# usage: tools/build_ablate_wino.sh "1 3 7 15"
set -e
R=$(cd $(dirname $0)/.. && pwd)
CS=$R/darknet_amd/csrc
make -s -C $CS -j8
mkdir -p $R/build_abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$CS/kernels -I$CS/host -Wno-unused-result -Wno-return-type-c-linkage -mllvm -pragma-unroll-threshold=200000 -Wno-pass-failed"
OTHERS=$(find $CS/build -name '*.o' | grep -v -e conv3x3_wino.o)
for b in $1; do
  ( d=$R/build_abl/w$b; mkdir -p $d
    /opt/rocm/bin/hipcc $FLAGS -DDK_WABL=$b -c $CS/kernels/conv3x3_wino.hip -o $d/conv3x3_wino.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_abl/libdk_wabl$b.so $d/conv3x3_wino.o $OTHERS -ldl -lpthread
    echo built $b ) &
done
wait
