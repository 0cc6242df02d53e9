#!/bin/bash
# Dev tool: diagnostic builds of the conv kernels with one part removed (DK_ABL bits, see
# conv_igemm.hip) -> build_abl/libdk_abl<bits>.so; select with DK_LIB=... (darknet_amd/__init__.py).
# usage: tools/build_ablate.sh "1 2 4 8 16"
set -e
R=$(cd $(dirname $0)/.. && pwd)
CS=$R/darknet_amd/csrc
make -s -C $CS -j8
mkdir -p $R/build_abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$CS/kernels -I$CS/host -Wno-unused-result -Wno-return-type-c-linkage -mllvm -pragma-unroll-threshold=200000 -Wno-pass-failed"
OTHERS=$(find $CS/build -name '*.o' | grep -v -e conv_igemm.o -e conv3x3_direct.o)
one() {
  b=$1
  d=$R/build_abl/o$b
  mkdir -p $d
  /opt/rocm/bin/hipcc $FLAGS -DDK_ABL=$b -c $CS/kernels/conv_igemm.hip -o $d/conv_igemm.o 2>/dev/null &
  /opt/rocm/bin/hipcc $FLAGS -DDK_ABL=$b -c $CS/kernels/conv3x3_direct.hip -o $d/conv3x3_direct.o 2>/dev/null &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_abl/libdk_abl$b.so $d/conv_igemm.o $d/conv3x3_direct.o $OTHERS -ldl -lpthread
  echo built $b
}
n=0
for b in $1; do
  one $b &
  n=$((n+1))
  if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
ls -la $R/build_abl/*.so
