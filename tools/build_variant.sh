#!/bin/bash
# Build a variant of libmpfmt.so with extra compiler flags into build_ab/libmpfmt_<NAME>.so (for tools/ab_multi.sh):
#   bash tools/build_variant.sh NAME "-DORD_OPT=3" "kernels_order"        (third argument: the translation units the flags touch;
#   the other objects are taken from the tree's own build -- run make there first; default: everything is rebuilt)
set -e
NAME=$1; FLAGS=$2; UNITS=$3     # (MAKEVARS="PEEPHOLE=0" in the environment: the build without the assembler peephole)
cd "$(dirname "$0")/.."
B=/tmp/mpfmt_variant_$NAME
rm -rf $B; mkdir -p $B/motionplanning.jl_amd $B/include build_ab
cp -rp motionplanning.jl_amd/csrc $B/motionplanning.jl_amd/
cp -p include/mpfmt.h $B/include/
if [ -z "$UNITS" ]; then rm -f $B/motionplanning.jl_amd/csrc/*.o; else for u in $UNITS; do rm -f $B/motionplanning.jl_amd/csrc/$u.o; done; fi
make -s -C $B/motionplanning.jl_amd/csrc -j8 $MAKEVARS CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-s-branch-bits=15 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $FLAGS" OUT=$B/libmpfmt.so >/dev/null
cp $B/libmpfmt.so build_ab/libmpfmt_$NAME.so
echo built build_ab/libmpfmt_$NAME.so
