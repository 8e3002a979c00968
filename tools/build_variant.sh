#!/bin/bash
# tools/build_variant.sh NAME FILE "EXTRA FLAGS" -- A/B builds: recompiles ONE translation unit with extra -D flags and links it
# with the stock objects into tools/_variants/NAME/libkzg_hip.so (select with KZG_HIP_LIB=... at run time).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; shift 2
OUT=$R/tools/_variants/$NAME; mkdir -p $OUT
make -C $R/go-kzg_amd/csrc -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKZG_FP_MUL_NOINLINE -Wno-unused-value -fvisibility=hidden"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/go-kzg_amd/csrc/$FILE.hip -o $OUT/$FILE.o
OBJS=""
for o in $R/go-kzg_amd/_build/*.o; do f=$(basename $o .o); if [ $f = $FILE ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libkzg_hip.so $OBJS
echo $OUT/libkzg_hip.so
