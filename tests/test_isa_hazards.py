"""CPU: the built gfx950 code is free of the store-data hazard the compiler does not pad (tools/probe/storehazard.hip).

Measured on MI355X: when a VALU instruction writes the first data register of a `buffer_store_dwordx4` / `dwordx3` in
the issue slot right behind the store, the store picks up the NEW value in lanes 12-15 of every 16 - about 1 % of the
stores when the data was itself produced by VALU instructions, 4e-7 of them when it came out of LDS - if the store's
soffset is an SGPR.  LLVM pads only the immediate-soffset form.  One wait state is enough; 8- and 4-byte stores are
not affected.  cbca_hwd_kernel's epilogue once produced wrong first components that way; common.h's
buffer_store_b128 / _b96 add the wait state.  This test disassembles lib/libmccnn_hip.so and fails on any 12- or
16-byte buffer store with an SGPR soffset whose data registers are written in the very next instruction."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mc-cnn-python_amd")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

STORE = re.compile(r"^(buffer_store_dwordx[34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def _regs(tok):
    m = VREG.match(tok.rstrip(","))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def _is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))


def scan(lines):
    """lines: [(kernel, text)] in program order; returns the offending stores."""
    hits = []
    for i, (kernel, text) in enumerate(lines):
        m = STORE.match(text)
        if not m or not m.group(5).startswith("s"):          # immediate soffset: the compiler pads it itself
            continue
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        if i + 1 >= len(lines) or lines[i + 1][0] != kernel:
            continue
        nxt = lines[i + 1][1].split()
        if not (_is_valu(nxt[0]) and len(nxt) > 1 and _regs(nxt[1]) & data):
            continue                                           # at least one wait state behind the store
        hits.append("%s: '%s' right behind '%s'" % (kernel, lines[i + 1][1], text))
    return hits


def disassemble(lib, workdir):
    shutil.copy(lib, os.path.join(workdir, "lib.so"))
    subprocess.check_call([OBJDUMP, "--offloading", "lib.so"], cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = []
    for f in sorted(os.listdir(workdir)):
        if "gfx950" not in f:
            continue
        out = subprocess.check_output([OBJDUMP, "-d", "--mcpu=gfx950", f], cwd=workdir, text=True)
        kernel = None
        for ln in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:$", ln)
            if m:
                kernel = m.group(1)
                continue
            if ln.startswith("\t") and kernel:
                lines.append((kernel, ln.split("//")[0].strip()))
    return lines


def test_scanner_finds_the_measured_pattern():
    bad = [("k", "v_div_fixup_f32 v0, v0, v3, v164"), ("k", "buffer_store_dwordx4 v[0:3], v124, s[0:3], s40 offen nt"),
           ("k", "v_fma_f32 v0, -v9, v12, 1.0")]
    assert len(scan(bad)) == 1
    padded = bad[:2] + [("k", "s_nop 0")] + bad[2:]
    assert scan(padded) == []
    from_lds = [("k", "ds_read_b128 v[0:3], v163"), ("k", "s_waitcnt lgkmcnt(0)"),
                ("k", "buffer_store_dwordx4 v[0:3], v8, s[40:43], s99 offen"), ("k", "v_or_b32_e32 v0, s69, v150")]
    assert len(scan(from_lds)) == 1                            # rarer (4e-7 per store), not safe
    narrow = [bad[0], ("k", "buffer_store_dwordx2 v[0:1], v124, s[0:3], s40 offen"), bad[2]]
    assert scan(narrow) == []
    imm = [bad[0], ("k", "buffer_store_dwordx4 v[0:3], v124, s[0:3], 0 offen"), bad[2]]
    assert scan(imm) == []                                     # the compiler's own two wait states cover this form


@pytest.mark.skipif(not os.path.isfile(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_built_library_has_no_unpadded_store_data_hazard(tmp_path):
    import _hipabi
    if not os.path.isfile(_hipabi.LIB_PATH):
        subprocess.check_call(["make", "-C", PKG, "-j4"])
    lines = disassemble(_hipabi.LIB_PATH, str(tmp_path))
    stores = [t for _, t in lines if t.startswith("buffer_store_dwordx4")]
    assert len(stores) > 50, "disassembly did not find the kernels' 16-byte stores"
    hits = scan(lines)
    assert not hits, "\n".join(hits)
