"""Rewrites profiles/pmc_traffic.json entries from PMC summaries (tools/pmc_kernel.sh output: FETCH_SIZE / WRITE_SIZE
in KiB per dispatch) and stamps them with the current hash of the kernel source, so that bench.py can tell whether
the kernel it runs is the one that was profiled.  usage: python tools/stamp_traffic.py KEY PMC.txt SOURCE.hip [KEY ...]
HBM bytes = FETCH_SIZE x 2 (gfx950 counts 64-byte requests in 32-byte units for these coalesced streams; checked on
wta_kernel's pure read, MI355X_MICROARCH.md) + WRITE_SIZE."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    with open(path) as f:
        table = json.load(f)
    args = sys.argv[1:]
    for key, pmc, src in zip(args[0::3], args[1::3], args[2::3]):
        vals = {}
        with open(os.path.join(ROOT, pmc)) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2 and parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[parts[0]] = float(parts[1])
        if len(vals) != 2:
            raise SystemExit("%s: FETCH_SIZE / WRITE_SIZE missing" % pmc)
        with open(os.path.join(ROOT, src), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        table[key] = {"fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                      "traffic_bytes": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024),
                      "source": src, "source_sha256_16": sha, "profile": pmc.replace("gpurun_out/r2_", "profiles/r02_")}
        print(key, table[key])
    with open(path, "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
