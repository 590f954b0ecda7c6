#!/bin/bash
# build_variant.sh <tag> <file.hip> <extra hipcc flags...>: one source recompiled with extra -D flags, linked with the in-tree
# objects of the other sources into tools/_bin/libivlm_<tag>.so (load it with IVLM_LIB_PATH)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; src=$2; shift 2
mkdir -p $R/tools/_bin/obj_$tag
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/interactvlm_amd/csrc "$@" -c $R/interactvlm_amd/csrc/$src -o $R/tools/_bin/obj_$tag/$base.o
objs=$(ls $R/interactvlm_amd/csrc/_obj/*.o | grep -v "/$base.o")
tl=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
g++ -shared -fPIC -o $R/tools/_bin/libivlm_$tag.so $objs $R/tools/_bin/obj_$tag/$base.o -L$tl -lamdhip64 -Wl,-rpath,$tl
echo built $R/tools/_bin/libivlm_$tag.so
