# On the GPU box: rocprofv3 kernel stats of the bench command, PMC passes for the dominant kernel (each --pmc set its
# own run, never combined with API tracing), the split-operand conv kernel's MFMA counters, and the final bench line.
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof_r02b; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02b -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r02b_bench.json 2> $R/gpurun_out/prof_r02b.err
cd $R
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter_pair cbca_stream > gpurun_out/r2_pmc_cbca_stream.txt 2>&1
grep "FETCH\|WRITE_SIZE\|TA_BUSY\|GRBM" gpurun_out/r2_pmc_cbca_stream.txt
SETS="1 2 4 5 6 7" bash tools/pmc_kernel.sh conv3x3_split conv3x3_split_kernel > gpurun_out/r2_pmc_conv3x3_split.txt 2>&1
grep "MFMA\|GRBM" gpurun_out/r2_pmc_conv3x3_split.txt
python bench.py --steps 20 --warmup 2 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['sgm_stage']['ms'])"
