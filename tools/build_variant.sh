#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]
# experiment build: the product library with ONE translation unit recompiled
# with extra flags -> tools/_exp/lib_<name>.so (load with XRD_LIB=<path>)
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/tools/_exp
obj=/tmp/_variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc \
  -I$root/include -I$root/xrdslam_amd/csrc "$@" -x hip -c $root/xrdslam_amd/csrc/${XRD_SRC_OVERRIDE:-$src} -o $obj
others=$(ls $root/xrdslam_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/_exp/lib_$name.so $others $obj
echo built tools/_exp/lib_$name.so
