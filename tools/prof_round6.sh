# On the GPU box (round 6): rocprofv3 table of the benchmarked command, PMC passes + traffic stamps, bench lines.
# usage: bash tools/prof_round6.sh [part]   (1: profile + PMC, 2: bench lines at every config; every command under a timeout)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r6; PART=${1:-1}; mkdir -p $O
if [ "$PART" = 1 ]; then
  cd /tmp
  rm -rf $O/prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity --no-bounds > $O/bench_under_rocprof.json 2> $O/prof.err
  cd $R
  SETS="1 2 3 4 5 6" timeout 400 bash tools/pmc_kernel.sh cbca_iter_prog_pair mccnn_cbca_prog_v4 > $O/pmc_cbca_prog.txt 2>&1
  SETS="1 2 3 4 5" timeout 400 bash tools/pmc_kernel.sh cbca_iter_prog_pair_skip mccnn_cbca_prog_v4_skip > $O/pmc_cbca_prog_skip.txt 2>&1
  SETS="3 4 5" timeout 300 bash tools/pmc_kernel.sh cbca_iter_prog mccnn_cbca_prog_v4 > $O/pmc_cbca_prog_one_volume.txt 2>&1
  SETS="1 2 3 4 5" timeout 300 bash tools/pmc_kernel.sh cbca_iter_prog_skip mccnn_cbca_prog_v4_skip > $O/pmc_cbca_prog_skip_one_volume.txt 2>&1
  SETS="3 4 5" timeout 300 bash tools/pmc_kernel.sh sgm_pass_h sgm_pass_kernel > $O/pmc_sgm_pass.txt 2>&1
  SETS="3 4 5" timeout 300 bash tools/pmc_kernel.sh sgm_pass_one_volume_h sgm_pass_kernel > $O/pmc_sgm_pass_one_volume.txt 2>&1
  SETS="1 2 4 5 6" timeout 300 bash tools/pmc_kernel.sh cost_volume_hwd cost_volume_exact_pairs > $O/pmc_cost_volume_exact_pairs.txt 2>&1
  SETS="1 2 7" timeout 300 bash tools/pmc_kernel.sh features_split conv3x3_split > $O/pmc_conv3x3_split.txt 2>&1
  timeout 300 python tools/bench_kernels.py > $O/kernel_microbench.txt 2>&1
else
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
  timeout 300 python bench.py --fast --no-cpu-baseline > $O/bench_fast.json 2> $O/bench_fast.err
  for c in cfg1 cfg3 cfg4; do
    timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  done
fi
ls -la $O | tail -30
