# On the GPU box, final state of round 3: rocprofv3 kernel stats of the bench command (fast headline and the bit-exact
# variant), PMC passes of the dominant kernels and of the SGM passes (each --pmc set its own run, never beside API
# tracing), the kernel micro-benchmarks, and the bench lines at every single-GPU config.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/prof_r03f $O/prof_r03fx; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03f -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/prof_r03f_bench.json 2> $O/prof_r03f.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03fx -o bench -- python $R/bench.py --steps 5 --warmup 2 --exact --no-cpu-baseline --no-parity > $O/prof_r03fx_bench.json 2> $O/prof_r03fx.err
cd $R
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter_hwd_pair cbca_hwd_kernel > $O/r3f_pmc_cbca_hwd.txt 2>&1
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter_pair cbca_stream > $O/r3f_pmc_cbca_stream.txt 2>&1
SETS="4 5" bash tools/pmc_kernel.sh sgm_pass_v sgm_pass_kernel > $O/r3f_pmc_sgm_pass.txt 2>&1
SETS="4 5" bash tools/pmc_kernel.sh sgm_first_pass sgm_first_pass > $O/r3f_pmc_sgm_first_pass.txt 2>&1
python tools/bench_kernels.py --iters 10 > $O/r3f_kernel_microbench.txt 2>&1
python bench.py --steps 20 --warmup 2 > $O/r3f_bench.json 2> $O/r3f_bench.err
python bench.py --steps 20 --warmup 2 --exact --no-cpu-baseline > $O/r3f_bench_exact.json 2> $O/r3f_bench_exact.err
for c in cfg1 cfg3 cfg4; do
  python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $O/r3f_bench_fast_$c.json 2> $O/r3f_bench_fast_$c.err
  python bench.py --config $c --steps 10 --warmup 2 --exact --no-cpu-baseline > $O/r3f_bench_exact_$c.json 2> $O/r3f_bench_exact_$c.err
done
grep "FETCH\|WRITE_SIZE" $O/r3f_pmc_*.txt
