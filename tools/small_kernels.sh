# On the GPU box: average duration of the small kernels of a pair (rocprofv3 kernel stats of a short bench run).
export TMPDIR=/tmp; R=$PWD; rm -rf /tmp/p1; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python $R/bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/p1/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("interpolate", "lr_status", "fill_kernel", "cross_", "flags", "bilateral", "median", "subpixel",
                            "conv1_split", "wta", "transpose_f4")):
        print("%-60s calls %4s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
