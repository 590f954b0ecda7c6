#!/bin/bash
# build_abl.sh <tag> "<a.hip b.hip ...>" <extra hipcc flags...>: the listed sources recompiled with extra -D flags, linked with the
# in-tree objects of all other sources into tools/_bin/libivlm_<tag>.so (load it with IVLM_LIB_PATH).  Used for the ablation
# libraries of the per-shape GEMM ceiling table (IVLM_ABL_* switches, gemm_common.h) and of the decode x-staging bound.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; srcs=$2; shift 2
mkdir -p $R/tools/_bin/obj_$tag
excl=""
pids=""
for src in $srcs; do
  base=$(basename $src .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$R/include -I$R/interactvlm_amd/csrc "$@" -c $R/interactvlm_amd/csrc/$src -o $R/tools/_bin/obj_$tag/$base.o &
  pids="$pids $!"
  excl="$excl -e /$base.o"
done
for p in $pids; do wait $p; done
objs=$(ls $R/interactvlm_amd/csrc/_obj/*.o | grep -v $excl)
tl=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
g++ -shared -fPIC -o $R/tools/_bin/libivlm_$tag.so $objs $R/tools/_bin/obj_$tag/*.o -L$tl -lamdhip64 -Wl,-rpath,$tl
echo built $R/tools/_bin/libivlm_$tag.so
