for c in cfg1 cfg3 cfg4; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['sgm_stage']['ms'], d['ms_per_step_host_in_host_out'])"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --exact 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 exact', d['ms_per_step'], d['value'], d['stage_ms_per_step'])"
python bench.py --steps 10 --warmup 2 > gpurun_out/r2_bench_full.json 2>gpurun_out/r2_bench_full.err; cat gpurun_out/r2_bench_full.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['value'], d['roofline'], d['sgm_stage'], d['cpu_baseline'], d['stage_ms_per_step'], d['ms_per_step_kernel_by_kernel'], d['ms_per_step_host_in_host_out'])"
python tools/bench_kernels.py --iters 20 2>&1 | grep -v amdgpu
