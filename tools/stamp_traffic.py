"""Rewrites profiles/pmc_traffic.json entries from PMC summaries (tools/pmc_kernel.sh output: FETCH_SIZE / WRITE_SIZE
in KiB per dispatch) and stamps them with the current hash of the kernel source, so that bench.py can tell whether
the kernel it runs is the one that was profiled.  usage: python tools/stamp_traffic.py KEY PMC.txt SOURCE.hip [KEY ...]
HBM bytes = FETCH_SIZE x f + WRITE_SIZE with f = 2 by default: gfx950 reports the 64-byte requests of coalesced
16-byte-per-lane streams at half their size (checked on wta_kernel's pure read, whose bytes are known;
MI355X_MICROARCH.md).  Kernels whose loads ask for whole 128-byte lines (sgm_first_pass's plane gathers: its FETCH_SIZE
already equals the volume it reads) are stamped with KEY:1."""
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
        key, _, fac = key.partition(":")
        fac = float(fac) if fac else 2.0
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
                      "fetch_factor": fac,
                      "traffic_bytes": int(vals["FETCH_SIZE"] * 1024 * fac + vals["WRITE_SIZE"] * 1024),
                      "source": src, "source_sha256_16": sha, "profile": pmc.replace("gpurun_out/r2_", "profiles/r02_")}
        print(key, table[key])
    with open(path, "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
