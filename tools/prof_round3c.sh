# On the GPU box, after the fused-WTA change: traffic of the pixel-major CBCA, rocprofv3 stats of the bit-exact variant,
# its bench lines at every single-GPU config, the split-features line, and the headline line again.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/prof_r03gx; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03gx -o bench -- python $R/bench.py --steps 5 --warmup 2 --exact --no-cpu-baseline --no-parity > $O/prof_r03gx_bench.json 2> $O/prof_r03gx.err
cd $R
SETS="1 2 3 4 5 6" bash tools/pmc_kernel.sh cbca_iter_hwd_pair cbca_hwd_kernel > $O/r3g_pmc_cbca_hwd.txt 2>&1
python bench.py --steps 20 --warmup 2 --exact --no-cpu-baseline > $O/r3g_bench_exact.json 2> $O/r3g_bench_exact.err
python bench.py --steps 20 --warmup 2 --exact --split-features --no-cpu-baseline > $O/r3g_bench_exact_split.json 2> $O/r3g_bench_exact_split.err
for c in cfg1 cfg3 cfg4; do
  python bench.py --config $c --steps 10 --warmup 2 --exact --no-cpu-baseline > $O/r3g_bench_exact_$c.json 2> $O/r3g_bench_exact_$c.err
done
python bench.py --steps 20 --warmup 2 > $O/r3g_bench.json 2> $O/r3g_bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
