#!/bin/bash
# Dev tool: stamped diagnostic build of ONE kernel file -> build_abl/libdk_<name>_stamp.so (select with DK_LIB=...)
# usage: tools/build_stamp.sh conv_igemm DK_GSTAMP     (file under csrc/kernels without .hip, the macro that enables its stamps)
set -e
R=$(cd $(dirname $0)/.. && pwd)
CS=$R/darknet_amd/csrc
NAME=$1; MACRO=$2
make -s -C $CS -j8
mkdir -p $R/build_abl/st_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$R/include -I$CS/kernels -I$CS/host -Wno-unused-result -Wno-return-type-c-linkage -mllvm -pragma-unroll-threshold=200000 -Wno-pass-failed"
OTHERS=$(find $CS/build -name '*.o' | grep -v -e /$NAME.o)
/opt/rocm/bin/hipcc $FLAGS -D$MACRO=1 -c $CS/kernels/$NAME.hip -o $R/build_abl/st_$NAME/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_abl/libdk_${NAME}_stamp.so $R/build_abl/st_$NAME/$NAME.o $OTHERS -ldl -lpthread
rm -rf $R/build_abl/st_$NAME
echo built $R/build_abl/libdk_${NAME}_stamp.so
