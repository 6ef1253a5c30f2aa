#!/usr/bin/env python3
"""Turns a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into the markdown table kept under
profiles/.

    python tools/summarize_stats.py gpurun_out/prof/bench_kernel_stats.csv "title line" > profiles/rNN_..._summary.md
"""
import csv
import sys


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s\n" % title)
    print("| kernel | calls | avg us | total ms | % |")
    print("|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"]
        if len(name) > 96:
            name = name[:93] + "..."
        t = float(r["TotalDurationNs"])
        if t / total < 2e-4:
            continue
        print("| `%s` | %s | %.1f | %.2f | %.1f |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, t / 1e6,
                                                  100.0 * t / total))


if __name__ == "__main__":
    main()
