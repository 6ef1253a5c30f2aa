#!/bin/bash
# On the GPU box: HBM-side traffic and L2 hit rate of the aggregation kernel's dev-harness variants, one rocprofv3
# counter pass each (--pmc with --kernel-trace only).  bash tools/pmc_prog_variants.sh > out.txt
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r5pv; rm -rf $O; mkdir -p $O; cd /tmp
run() { # name, harness args...
  n=$1; shift
  for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$n -o p -- python $R/tools/dev_prog_check.py --config cfg2 --iters 3 --skip-small "$@" > $O/$n.log 2>&1
    python - "$O/$n" "$n" <<'PY'
import csv,glob,sys
agg={}
for f in glob.glob(sys.argv[1]+"/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mccnn_cbca_prog" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print(sys.argv[2], " ".join("%s=%.0f(n=%d)"%(k,sum(v[1:])/max(1,len(v)-1),len(v)) for k,v in sorted(agg.items())))
PY
    rm -rf $O/$n
  done
  grep "dropped\|cbca_prog pair" $O/$n.log
}
run base_full --k 4 --w 20 --drop 0,0
run base_noadds --k 4 --w 20 --drop 0,1
run pipe_full --k 4 --w 42 --pipe 21 --drop 0,0
run pipe_noadds --k 4 --w 42 --pipe 21 --drop 0,1
run pipe1w_full --k 4 --w 42 --pipe 21 --drop 0,0 --minvgpr 512
