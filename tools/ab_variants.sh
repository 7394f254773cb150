#!/bin/bash
# Build variants of the library that differ in geodesic.hip's -D macros, side by side under tools/_diag (build container; they travel
# to the GPU box with the snapshot).   bash tools/ab_variants.sh name1 "-DPOPE_AHEAD=0" name2 "-DPOPE_AHEAD=2 -DPOPE_TILE_PREFETCH=0" ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/graphpope_amd/csrc
make -s -C $C
mkdir -p $R/tools/_diag
others=$(ls $C/build/*.o | grep -v geodesic.o)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wall -Wno-unused-result $flags -c $C/geodesic.hip -o /tmp/geo_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $R/tools/_diag/lib_$name.so $others /tmp/geo_$name.o && echo "built lib_$name.so ($flags)" ) &
done
wait
