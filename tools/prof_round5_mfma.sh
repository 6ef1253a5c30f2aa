#!/bin/bash
# On the GPU box (round 5): counter passes for the matrix-core and cost-volume kernels on the current code
# (tools/pmc_kernel.sh: one --pmc set per run, --kernel-trace only).  bash tools/prof_round5_mfma.sh
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r5pmc; mkdir -p $O
SETS="1 2 7 3 4 5" timeout 500 bash tools/pmc_kernel.sh conv3x3_split conv3x3_split_kernel > $O/pmc_conv3x3_split.txt 2>&1
SETS="1 2 7 3 4 5" timeout 500 bash tools/pmc_kernel.sh cost_volume_mfma cost_volume_mfma_kernel > $O/pmc_cost_volume_mfma.txt 2>&1
SETS="1 2 3 4 5 6" timeout 500 bash tools/pmc_kernel.sh cost_volume_hwd cost_volume_exact_pairs_kernel > $O/pmc_cost_volume_exact_pairs.txt 2>&1
timeout 300 python tools/bench_kernels.py --only conv3x3_split,cost_volume_mfma,cost_volume_hwd,features_split,cost_volume_exact > $O/microbench.txt 2>&1
tail -n 40 $O/*.txt
