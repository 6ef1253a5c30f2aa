# On the GPU box: rocprofv3 kernel stats of the bench command (fast headline and the bit-exact variant), PMC passes of
# the dominant kernels (each --pmc set its own run, never beside API tracing), and the final bench lines.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; rm -rf $O/prof_r03 $O/prof_r03x; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/prof_r03_bench.json 2> $O/prof_r03.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03x -o bench -- python $R/bench.py --steps 5 --warmup 2 --exact --no-cpu-baseline --no-parity > $O/prof_r03x_bench.json 2> $O/prof_r03x.err
cd $R
SETS="1 2 3 4 5 6" bash tools/pmc_cmd.sh hwd cbca_hwd python $R/tools/dev_hwd_check.py --skip-check --iters 2 > $O/r3_pmc_cbca_hwd.txt 2>&1
python bench.py --steps 20 --warmup 2 > $O/r3_bench_final.json 2> $O/r3_bench_final.err
python bench.py --steps 20 --warmup 2 --exact --no-cpu-baseline > $O/r3_bench_exact_final.json 2> $O/r3_bench_exact_final.err
python - <<PY
import json
for f in ("gpurun_out/r3_bench_final.json","gpurun_out/r3_bench_exact_final.json"):
    d=json.load(open(f)); print(f, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["sgm_stage"]["ms"], d["sgm_stage"]["frac_of_hbm_peak"], d.get("exact_variant_ms_per_step"), d["parity"] and (d["parity"]["wta_flips_left"], d["parity"]["frac_within_1e-3_px"], d["parity"]["timed_path_equals_kernel_by_kernel"]), d["cpu_baseline"] and d["cpu_baseline"]["value"])
    print(d["stage_ms_per_step"])
PY
