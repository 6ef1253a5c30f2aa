#!/bin/bash
# usage: tools/build_prog_variant.sh NAME "generator flags" ["-D flags for cbca_prog.hip"]
#   -> mc-cnn-python_amd/build/variants/libmccnn_NAME.so : the library with the program-driven aggregation kernels
# generated with other parameters (--k --w --minvgpr --order ..., csrc/asm/cbca_prog_gen.py); every other object is
# taken from the regular build (run `make -C mc-cnn-python_amd` first).  Select at run time with MCCNN_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../mc-cnn-python_amd"
NAME=$1; GEN=$2; DEFS=$3
LLVM=/opt/rocm/lib/llvm/bin
A=build/variants/$NAME/asm; mkdir -p $A
for v in 2 3 4; do
  python3 csrc/asm/cbca_prog_gen.py --experimental --vpl $v $GEN -o $A/cbca_prog_v$v.s --header $A/cbca_prog_layout_v$v.h
  python3 csrc/asm/cbca_prog_gen.py --experimental --vpl $v $GEN --wta -o $A/cbca_prog_v${v}w.s
  python3 csrc/asm/cbca_prog_gen.py --experimental --vpl $v $GEN --skip -o $A/cbca_prog_v${v}s.s
  for s in v$v v${v}w v${v}s; do
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $A/cbca_prog_$s.s -o $A/cbca_prog_$s.o
    $LLVM/ld.lld -shared $A/cbca_prog_$s.o -o $A/cbca_prog_$s.hsaco
    python3 csrc/asm/bin2inc.py $A/cbca_prog_$s.hsaco $A/cbca_prog_$s.inc
  done
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  $DEFS -I$A -I../include -Icsrc -c csrc/cbca_prog.hip -o build/variants/$NAME/cbca_prog.o
OBJS=$(ls build/*.o | grep -v cbca_prog.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libmccnn_$NAME.so $OBJS build/variants/$NAME/cbca_prog.o
echo build/variants/libmccnn_$NAME.so
