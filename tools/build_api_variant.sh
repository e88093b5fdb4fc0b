#!/bin/bash
# A/B builds of the host file only:  tools/build_api_variant.sh <name> [-DSVSX_...=...]  -> variants/<name>/libsvsx.so (in-tree objects + svsx_api.o recompiled)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/variants/$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
cd $R/scalablevectorsearch_amd/csrc
/opt/rocm/bin/hipcc $FLAGS "$@" -c svsx_api.hip -o $OUT/svsx_api.o
OTHERS=$(ls *.o | grep -v -E '^svsx_api\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsvsx.so $OUT/svsx_api.o $OTHERS
rm -f $OUT/*.o
echo built $OUT/libsvsx.so
