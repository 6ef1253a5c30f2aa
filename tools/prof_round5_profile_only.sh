export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O; cd /tmp; rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity --no-bounds > $O/bench_under_rocprof.json 2> $O/prof.err
cd $R; python tools/summarize_stats.py $O/prof/bench_kernel_stats.csv | head -20
