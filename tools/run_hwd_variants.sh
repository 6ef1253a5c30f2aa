#!/bin/bash
# usage (GPU box): VARIANTS="a b" [CFG=cfg2] [CHECK=1] bash tools/run_hwd_variants.sh -> timing of the pixel-major CBCA per variant
CFG=${CFG:-cfg2}
mkdir -p gpurun_out
for v in $VARIANTS; do
  lib=$PWD/mc-cnn-python_amd/build/variants/libmccnn_$v.so
  [ "$v" = base ] && lib=$PWD/mc-cnn-python_amd/lib/libmccnn_hip.so
  if [ -n "$CHECK" ]; then extra=""; else extra="--skip-check"; fi
  r=$(MCCNN_HIP_LIB=$lib python tools/dev_hwd_check.py --config $CFG --iters 10 $extra 2>&1 | grep -v amdgpu.ids | grep "cbca_iter_hwd\|ALL OK\|MISMATCH\|False" | tr '\n' ' ')
  echo "$v [$CFG]: $r" | tee -a gpurun_out/hwd_variants.txt
done
