#!/bin/bash
# A/B builds of the library: one source recompiled with extra -D flags, linked with the product's
# other objects into sporco_amd/variants/libsporco_amd_<tag>.so (git-ignored; travels with gpurun;
# selected with SPORCO_AMD_LIBRARY=...).
#   tools/build_variant.sh <tag> <source.hip> [-DFLAG=V ...]
cd "$(dirname "$0")/../sporco_amd/csrc" || exit 1
TAG=$1; SRC=$2; shift 2
make -s >/dev/null || exit 1
mkdir -p ../variants
EXTRA=""
# (the register-kernel translation units take the Makefile's REGFLAGS; pass -ffp-contract=on after them to A/B that)
case $SRC in csc_fused.hip|csc_fused_mc.hip|csc_rows.hip|csc_pgm.hip|csc_rows_mr.hip|csc_rows_mr2.hip|csc_pgm_mr.hip|csc_pgm_mr2.hip) EXTRA="-fno-slp-vectorize -ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -Wall -Wno-unused-function $EXTRA "$@" -c $SRC -o ../variants/${SRC%.hip}_$TAG.o || exit 1
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../variants/${SRC%.hip}_$TAG.o -o ../variants/libsporco_amd_$TAG.so && rm -f ../variants/${SRC%.hip}_$TAG.o
ls -la ../variants/libsporco_amd_$TAG.so
