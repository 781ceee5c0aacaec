#!/bin/bash
# Dev tool: variants/lib_<name>.so = the library with ONE source rebuilt with extra flags.  usage: build_variant.sh name file.hip flags...
set -e
R=$(cd $(dirname $0)/.. && pwd); name=$1; src=$2; shift 2
mkdir -p $R/variants/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $R/tacotron2_subword_amd/csrc/$src -o $R/variants/obj/$name.o
objs=""
for o in $R/tacotron2_subword_amd/build/*.o; do [ "$(basename $o)" = "$src.o" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/lib_$name.so $objs $R/variants/obj/$name.o
echo built variants/lib_$name.so
