#!/bin/bash
# usage (GPU box): [SETS="1 2 3 4 5 6"] bash tools/pmc_cmd.sh <tag> <kernel-name substring> <command ...>
# rocprofv3 PMC passes (one counter set per run; --pmc only ever beside --kernel-trace) of an arbitrary command, means
# per dispatch of the kernels whose name contains the substring.  Sets as in tools/pmc_kernel.sh.
TAG=$1; KSUB=$2; shift 2; SETS=${SETS:-"1 2 3 4 5 6"}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; cd /tmp
declare -A CTR
CTR[1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM"
CTR[2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_INSTS_BRANCH"
CTR[3]="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
CTR[4]="FETCH_SIZE"
CTR[5]="WRITE_SIZE"
CTR[6]="GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
CTR[7]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum"
CTR[8]="SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES"
for i in $SETS; do
  rocprofv3 --kernel-trace --pmc ${CTR[$i]} --output-format csv -d $OUT/s$i -o p -- "$@" > $OUT.s$i.log 2>&1
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
