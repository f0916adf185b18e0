#!/bin/bash
# Experiment builds: tools/build_variant.sh NAME [-DMACRO=..]...  ->  sea-current_amd/variants/NAME.so (astar.hip rebuilt
# with the given macros, the other objects as they are); select with SC_LIB_PATH.  ASTAR_SRC=file: another astar.hip
# (e.g. `git show HEAD:sea-current_amd/csrc/astar.hip > sea-current_amd/csrc/astar_prev.hip`), for A/B runs on one box.
set -e
cd "$(dirname "$0")/../sea-current_amd"
name=$1; shift
mkdir -p variants
make -s all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c ${ASTAR_SRC:-csrc/astar.hip} -o variants/$name.astar.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so csrc/ctx.o csrc/edt.o variants/$name.astar.o csrc/toppra.o csrc/bezier.o csrc/grid.o csrc/fmt.o csrc/gather.o -ldl
rm -f variants/$name.astar.o
echo built variants/$name.so
