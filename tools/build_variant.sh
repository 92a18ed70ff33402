#!/bin/bash
# Kernel tuning: build a variant of libgsplat_mi355.so with extra -D flags for ONE translation unit.
#   tools/build_variant.sh <name> <unit.hip> <flags...>   ->  tools/_variants/lib_<name>.so
# Run a bench against it with GSPLAT_MI355_LIB=tools/_variants/lib_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../gaussian-splatterer_amd/csrc"
name=$1; unit=$2; shift 2
mkdir -p ../../tools/_variants/_obj_$name
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $unit -o ../../tools/_variants/_obj_$name/${unit%.hip}.o
objs=""
for o in _obj/*.o; do b=$(basename $o); if [ "$b" = "${unit%.hip}.o" ]; then objs="$objs ../../tools/_variants/_obj_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_variants/lib_$name.so $objs -ldl
echo built tools/_variants/lib_$name.so
