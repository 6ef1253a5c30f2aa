#!/bin/bash
# usage (on the GPU box): VARIANTS="a b c" [ONLY=cbca_iter] [CFG=cfg2] bash tools/run_variants.sh  -> one line per variant
ONLY=${ONLY:-cbca_iter}; CFG=${CFG:-cfg2}
mkdir -p gpurun_out
for v in $VARIANTS; do
  r=$(MCCNN_HIP_LIB=$PWD/mc-cnn-python_amd/build/variants/libmccnn_$v.so python tools/bench_kernels.py --config $CFG --iters 20 --only $ONLY 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')
  echo "$v: $r" | tee -a gpurun_out/variants.txt
done
