#!/bin/bash
# A/B builds of the flat-search kernels only:  tools/build_flat_variant.sh <name> [-DSVSX_FLAT_...=...]
# -> variants/<name>/libsvsx.so = the in-tree objects (run `make` first) with svsx_flat.o recompiled under the given defines.
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/variants/$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
cd $R/scalablevectorsearch_amd/csrc
/opt/rocm/bin/hipcc $FLAGS "$@" -c svsx_flat.hip -o $OUT/svsx_flat.o
OTHERS=$(ls *.o | grep -v -E '^svsx_flat\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsvsx.so $OUT/svsx_flat.o $OTHERS
rm -f $OUT/*.o
echo built $OUT/libsvsx.so
