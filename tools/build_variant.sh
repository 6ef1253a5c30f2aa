#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFOO=1 -DBAR=2"   -> mc-cnn-python_amd/build/variants/libmccnn_NAME.so
# A/B builds of the same library with different compile-time kernel parameters; select one at run time with
# MCCNN_HIP_LIB=<path> (see src/_hipabi.py).  Only the sources are rebuilt that the flags can affect (all of them).
set -e
cd "$(dirname "$0")/../mc-cnn-python_amd"
NAME=$1; FLAGS=$2
OUT=build/variants/$NAME; mkdir -p $OUT
for f in csrc/*.hip; do
  o=$OUT/$(basename ${f%.hip}).o
  X=""; [ "$(basename $f)" = cbca_hwd.hip ] && [ -z "$SLP" ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $X --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $FLAGS -I../include -Icsrc -c $f -o $o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libmccnn_$NAME.so $OUT/*.o
echo build/variants/libmccnn_$NAME.so
