#!/bin/bash
# builds tools/exp/ab/NAME.so = the tree's libtengine_amd.so with ONE source recompiled with extra flags (A/B builds for ab_lib.py)
# usage: [REPLACES=other.hip] build_variant.sh NAME source.hip [-DFLAG ...]        (run after tengine_amd/build.py has built the tree;
#        REPLACES: the tree object the new one stands in for when `source.hip` is another revision under another name)
set -e
R=$(cd $(dirname $0)/../.. && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p $R/tools/exp/ab/obj_$NAME
VG=""
case $SRC in pw_stream.hip|pw_rows.hip|conv_igemm.hip|conv_igemm2.hip|conv_pgemm.hip|conv_pgemm_w.hip|conv_first.hip|conv_first_pool.hip|dwpw.hip|gemm_direct.hip|u8i_kernels.hip) VG="-mllvm --amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-value $VG "$@" -c $R/tengine_amd/csrc/$SRC -o $R/tools/exp/ab/obj_$NAME/$SRC.o 2>&1 | grep -v warning | head -5
OBJS=$(ls $R/tengine_amd/lib/obj/*.o | grep -v "/${REPLACES:-$SRC}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/ab/$NAME.so $OBJS $R/tools/exp/ab/obj_$NAME/$SRC.o -lhsa-runtime64
ls -la $R/tools/exp/ab/$NAME.so
