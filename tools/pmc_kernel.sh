#!/bin/bash
# usage: bash tools/pmc_kernel.sh <bench_kernels --only name> <kernel-name substring>   (run on the GPU box)
# Collects a few rocprofv3 PMC sets (separate passes: --pmc must not be combined with tracing) and prints per-dispatch means.
ONLY=${1:-cbca_iter}; KSUB=${2:-cbca_pipe}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_$ONLY; rm -rf $OUT; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -o p -- python $R/tools/bench_kernels.py --iters 2 --only $ONLY > /dev/null 2>&1
done
cd $R
python - "$KSUB" <<'PY'
import csv,glob,sys
ksub=sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc_*/s*/*counter_collection.csv")):
    agg={}
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-32s %16.1f  (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
