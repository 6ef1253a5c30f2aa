# On the GPU box (round 4): rocprofv3 tables of both variants, PMC passes + traffic stamps, bench lines at every config.
# usage: bash tools/prof_round4.sh [part]   (part 1: profiles + PMC, part 2: bench lines; every command under a timeout)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; PART=${1:-1}
if [ "$PART" = 1 ]; then
  cd /tmp
  rm -rf $O/prof_r04 $O/prof_r04f
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04 -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity > $O/prof_r04_bench.json 2> $O/prof_r04.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04f -o bench -- python $R/bench.py --fast --steps 20 --warmup 2 --no-cpu-baseline --no-parity > $O/prof_r04f_bench.json 2> $O/prof_r04f.err
  cd $R
  SETS="1 2 3 4 5 6" timeout 400 bash tools/pmc_kernel.sh cbca_iter_prog_pair mccnn_cbca_prog_v4 > $O/r4_pmc_cbca_prog.txt 2>&1
  SETS="1 3 4 5" timeout 300 bash tools/pmc_kernel.sh cbca_iter_prog_pair_skip mccnn_cbca_prog_v4_skip > $O/r4_pmc_cbca_prog_skip.txt 2>&1
  SETS="4 5" timeout 200 bash tools/pmc_kernel.sh cbca_iter_hwd_pair cbca_hwd_kernel > $O/r4_pmc_cbca_hwd.txt 2>&1
  SETS="4 5" timeout 200 bash tools/pmc_kernel.sh cbca_iter_pair cbca_stream_kernel > $O/r4_pmc_cbca_stream.txt 2>&1
  SETS="2 3 4 5 6" timeout 300 bash tools/pmc_kernel.sh sgm_pass_h sgm_pass_kernel > $O/r4_pmc_sgm_pass.txt 2>&1
  SETS="4 5" timeout 200 bash tools/pmc_kernel.sh sgm_first_pass sgm_first_pass_kernel > $O/r4_pmc_sgm_first_pass.txt 2>&1
  timeout 300 python tools/bench_kernels.py > $O/r4_kernel_microbench.txt 2>&1
else
  timeout 400 python bench.py > $O/r4_bench.json 2> $O/r4_bench.err
  timeout 300 python bench.py --fast --no-cpu-baseline > $O/r4_bench_fast.json 2> $O/r4_bench_fast.err
  timeout 300 python bench.py --fast --separable-cbca --no-cpu-baseline > $O/r4_bench_fast_separable.json 2> $O/r4_bench_fast_separable.err
  timeout 300 python bench.py --library-features --no-cpu-baseline > $O/r4_bench_library_features.json 2> $O/r4_bench_library_features.err
  for c in cfg1 cfg3 cfg4; do
    timeout 400 python bench.py --config $c --no-cpu-baseline > $O/r4_bench_$c.json 2> $O/r4_bench_$c.err
    timeout 400 python bench.py --config $c --fast --no-cpu-baseline > $O/r4_bench_fast_$c.json 2> $O/r4_bench_fast_$c.err
  done
  timeout 700 python bench.py --steps 20 --no-parity --cpu-sample cfg2 --cpu-cores 1 > $O/r4_bench_cpu_cfg2.json 2> $O/r4_bench_cpu_cfg2.err
fi
ls -la $O | tail -30
