export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof_r02a; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02a -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_r02a_bench.json 2> $R/gpurun_out/prof_r02a.err
cd $R
tail -2 gpurun_out/prof_r02a.err
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter cbca_stream > gpurun_out/r2_pmc_cbca_stream.txt 2>&1
cat gpurun_out/r2_pmc_cbca_stream.txt
ls gpurun_out/prof_r02a
