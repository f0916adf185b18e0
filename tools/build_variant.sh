#!/bin/bash
# Experiment builds: tools/build_variant.sh NAME [-DMACRO=..]...  ->  sea-current_amd/variants/NAME.so (astar.hip rebuilt
# with the given macros, the other objects as they are); select with SC_LIB_PATH.  ASTAR_SRC=file: another astar.hip
# (e.g. `git show HEAD:sea-current_amd/csrc/astar.hip > sea-current_amd/csrc/astar_prev.hip`), for A/B runs on one box.
set -e
cd "$(dirname "$0")/../sea-current_amd"
name=$1; shift
mkdir -p variants
make -s all
unit=${VARIANT_UNIT:-astar}     # which source is rebuilt with the macros (astar, edt, bezier, ...)
src=csrc/$unit.hip
[ $unit = astar ] && src=${ASTAR_SRC:-csrc/astar.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $src -o variants/$name.$unit.o
objs=""
for o in ctx edt astar toppra bezier grid fmt gather; do
  if [ $o = $unit ]; then objs="$objs variants/$name.$unit.o"; else objs="$objs csrc/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so $objs -ldl
rm -f variants/$name.$unit.o
echo built variants/$name.so
