#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -S --cuda-device-only) for the pattern that produced wrong results in cbca_hwd_kernel's
epilogue: a 16-byte (or 12-byte) vector store whose data registers are overwritten by a VALU instruction within the
next few instructions.  The compiler leaves 2 wait states; the observed failure had exactly that distance.
    python tools/scan_store_hazard.py file.s [max_distance]"""
import re, sys
path = sys.argv[1]; maxd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
store = re.compile(r"^\s+(buffer_store_dwordx[34]|global_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
reg_range = re.compile(r"v\[(\d+):(\d+)\]")
kernel = None; lines = []
for ln in open(path):
    if ln.startswith("_Z") and ln.rstrip().endswith(":"):
        kernel = ln.strip()[:-1]
    s = ln.split(";")[0].rstrip()
    if re.match(r"^\s+[a-z]", s): lines.append((kernel, s))
    elif re.match(r"^\.LBB", s): lines.append((kernel, "LABEL"))
hits = 0
for i, (k, s) in enumerate(lines):
    m = store.match(s)
    if not m: continue
    ops = m.group(2)
    name = m.group(1)
    # data operand: first operand for buffer_store, second for global_store
    parts = [p.strip() for p in ops.split(",")]
    data = parts[0] if name.startswith("buffer") else parts[1]
    r = reg_range.match(data)
    if not r: continue
    regs = set(range(int(r.group(1)), int(r.group(2)) + 1))
    d = 0
    for j in range(i + 1, min(i + 1 + 40, len(lines))):
        t = lines[j][1]
        if t == "LABEL" or t.lstrip().startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")): break
        op = t.split()[0]
        # wait states: s_nop N counts N+1, everything else 1
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dst = t.split()[1].rstrip(",")
            w = set()
            rr = reg_range.match(dst)
            if rr: w = set(range(int(rr.group(1)), int(rr.group(2)) + 1))
            elif re.match(r"v\d+$", dst): w = {int(dst[1:])}
            if w & regs:
                if d <= maxd:
                    hits += 1
                    print("%s: '%s' -> overwritten after %d wait states by '%s'" % (k, s.strip(), d, t.strip()))
                break
        d += (int(t.split()[1], 0) + 1) if op == "s_nop" else 1
print("%s: %d store(s) with data overwritten within %d wait states" % (path, hits, maxd))
