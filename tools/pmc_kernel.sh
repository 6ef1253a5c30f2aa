#!/bin/bash
# usage: [SETS="1 2 3 4 5 6"] bash tools/pmc_kernel.sh <bench_kernels --only name> <kernel-name substring>   (on the GPU box)
# Collects rocprofv3 PMC sets for one kernel of tools/bench_kernels.py and prints per-dispatch means.  Each set is its
# own pass and --pmc is never combined with API/memory tracing (only --kernel-trace, which the counter CSV needs).
# Sets: 1 instruction mix, 2 pipe activity / waits, 3 L2 hit/miss, 4 FETCH_SIZE, 5 WRITE_SIZE, 6 TA / L1 stalls,
# 7 MFMA utilisation.
ONLY=${1:-cbca_iter}; KSUB=${2:-cbca_pipe}; SETS=${SETS:-"1 2 3 4 5 6"}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_$ONLY; rm -rf $OUT; cd /tmp
declare -A CTR
CTR[1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS"
CTR[2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
CTR[3]="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
CTR[4]="FETCH_SIZE"
CTR[5]="WRITE_SIZE"
CTR[6]="GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
CTR[7]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
for i in $SETS; do
  rocprofv3 --kernel-trace --pmc ${CTR[$i]} --output-format csv -d $OUT/s$i -o p -- python $R/tools/bench_kernels.py --iters 2 --only $ONLY > /dev/null 2>&1
done
cd $R
python - "$KSUB" "$OUT" <<'PY'
import csv,glob,sys
ksub,out=sys.argv[1],sys.argv[2]
for f in sorted(glob.glob(out+"/s*/*counter_collection.csv")):
    agg={}
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%-32s %16.1f  (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
