#!/bin/bash
# A/B builds of libsvsx.so with extra hipcc defines:  tools/build_variant.sh <name> [-DFOO=1 ...]
# -> variants/<name>/libsvsx.so (git-ignored; travels with gpurun).  Select at run time with SVSX_LIB=variants/<name>/libsvsx.so
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/variants/$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
cd $R/scalablevectorsearch_amd/csrc
for f in svsx_search_f32 svsx_search_f16 svsx_search_i8 svsx_search_u8; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $OUT/$f.o & done
/opt/rocm/bin/hipcc $FLAGS "$@" -c svsx_api.hip -o $OUT/svsx_api.o &
wait
# every other object of the in-tree build (run `make` first), whatever the Makefile's source list is
OTHERS=$(ls *.o | grep -v -E '^svsx_(search_(f32|f16|i8|u8)|api)\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsvsx.so $OUT/svsx_search_f32.o $OUT/svsx_search_f16.o $OUT/svsx_search_i8.o $OUT/svsx_search_u8.o $OUT/svsx_api.o $OTHERS
rm -f $OUT/*.o
echo built $OUT/libsvsx.so
