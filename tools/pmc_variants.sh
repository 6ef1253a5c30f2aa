#!/bin/bash
# usage (GPU box): VARIANTS="a b" [SETS="1 3 4"] [ONLY=cbca_iter_prog_pair] [KSUB=mccnn_cbca_prog] bash tools/pmc_variants.sh
# PMC passes (tools/pmc_kernel.sh) of one micro-benchmarked kernel for several library variants (tools/build_prog_variant.sh)
ONLY=${ONLY:-cbca_iter_prog_pair}; KSUB=${KSUB:-mccnn_cbca_prog}
for v in $VARIANTS; do
  echo "== $v"
  MCCNN_HIP_LIB=$PWD/mc-cnn-python_amd/build/variants/libmccnn_$v.so SETS="${SETS:-1 3 4}" bash tools/pmc_kernel.sh $ONLY $KSUB
done
