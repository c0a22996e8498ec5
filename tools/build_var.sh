#!/bin/bash
# A/B builds of the library with extra -D flags: tools/build_var.sh NAME "-DHB_G=4 ..." -> var/libdr_NAME.so
# (select with DEEPREC_AMD_LIB=var/libdr_NAME.so; var/ is scratch, not committed)
set -e
cd "$(dirname "$0")/../deeprec_amd/csrc"
name=$1; shift
mkdir -p build/var_$name ../../var
for f in abi gather wire segment shard pairwise interaction mlp attention seqloss score_topk tower_score; do
  [ -f build/$f.o ] || { echo "build the library first"; exit 1; }
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $@ -c ${SRC:-caser_conv}.hip -o build/var_$name/${SRC:-caser_conv}.o
objs=""
for f in abi gather wire segment shard pairwise interaction mlp attention caser_conv seqloss score_topk tower_score; do
  if [ -f build/var_$name/$f.o ]; then objs="$objs build/var_$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../../var/libdr_$name.so
echo var/libdr_$name.so
