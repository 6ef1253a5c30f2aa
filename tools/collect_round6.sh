# In the dev container, after `gpurun -- bash tools/prof_round6.sh {1,2}`: copies the round's evidence from the scratch
# directory gpurun_out/r6 into profiles/ (tracked) and re-stamps profiles/pmc_traffic.json.   bash tools/collect_round6.sh
set -e
S=gpurun_out/r6; P=profiles
for f in bench bench_fast bench_cfg1 bench_cfg3 bench_cfg4 bench_under_rocprof; do
  [ -s $S/$f.json ] && cp $S/$f.json $P/r06_$f.json
done
[ -s $S/prof/bench_kernel_stats.csv ] && cp $S/prof/bench_kernel_stats.csv $P/r06_bench_kernel_stats.csv && \
  python tools/summarize_stats.py $P/r06_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity --no-bounds (round 6, cfg2, one MI355X)" > $P/r06_bench_kernel_stats_summary.md
for f in pmc_cbca_prog pmc_cbca_prog_skip pmc_cbca_prog_one_volume pmc_cbca_prog_skip_one_volume pmc_sgm_pass pmc_sgm_pass_one_volume pmc_cost_volume_exact_pairs pmc_conv3x3_split kernel_microbench; do
  [ -s $S/$f.txt ] && grep -v "amdgpu.ids" $S/$f.txt > $P/r06_$f.txt
done
G=mc-cnn-python_amd/csrc/asm/cbca_prog_gen.py
python tools/stamp_traffic.py cbca_iter_prog_pair $P/r06_pmc_cbca_prog.txt $G cbca_iter_prog_pair_skip $P/r06_pmc_cbca_prog_skip.txt $G \
  cbca_iter_prog $P/r06_pmc_cbca_prog_one_volume.txt $G cbca_iter_prog_skip $P/r06_pmc_cbca_prog_skip_one_volume.txt $G \
  sgm_pass $P/r06_pmc_sgm_pass.txt mc-cnn-python_amd/csrc/sgm.hip sgm_pass_one_volume $P/r06_pmc_sgm_pass_one_volume.txt mc-cnn-python_amd/csrc/sgm.hip
