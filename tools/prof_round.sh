export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof_r02b; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02b -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r02b_bench.json 2> $R/gpurun_out/prof_r02b.err
cd $R
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter cbca_stream > gpurun_out/r2_pmc_cbca_stream.txt 2>&1
cat gpurun_out/r2_pmc_cbca_stream.txt | grep "FETCH\|WRITE_SIZE\|TA_BUSY\|GRBM"
python bench.py --steps 20 --warmup 2 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['sgm_stage']['ms'])"
