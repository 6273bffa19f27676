#!/bin/bash
# usage: tools/ab_build.sh <tag> <source.hip> [extra hipcc flags...]  -> abtest/lib_<tag>.so = the current library with
# <source.hip> recompiled under the extra flags (A/B probes via AIDE_HIP_LIB=abtest/lib_<tag>.so)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p abtest
obj=abtest/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c aide_amd/csrc/$src -o $obj
others=$(ls aide_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abtest/lib_$tag.so $obj $others
echo built abtest/lib_$tag.so
