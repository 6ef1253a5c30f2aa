#!/bin/bash
# usage: tools/build_src_variant.sh NAME SRC.hip "-DFOO=1"   -> mc-cnn-python_amd/build/variants/libmccnn_NAME.so
# A/B build of ONE source file with extra flags; every other object comes from the regular build (make first).
# Select at run time with MCCNN_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../mc-cnn-python_amd"
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p build/variants/$NAME
X=""; [ "$SRC" = cbca_hwd.hip ] && X="-fno-slp-vectorize"
/opt/rocm/bin/hipcc $X --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  $FLAGS -Ibuild/asm -I../include -Icsrc -c csrc/$SRC -o build/variants/$NAME/${SRC%.hip}.o
OBJS=$(ls build/*.o | grep -v "build/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libmccnn_$NAME.so $OBJS build/variants/$NAME/${SRC%.hip}.o
echo build/variants/libmccnn_$NAME.so
