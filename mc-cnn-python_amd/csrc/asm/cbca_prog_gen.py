#!/usr/bin/env python3
"""Generator of the gfx950 assembly of `cbca_prog_kernel`: a4 cost_volume_aggregation
(/root/reference/src/process_functional.py:149-163) in the reference's summation order on pixel-major volumes, run as a
PRE-COMPILED PER-PATCH PROGRAM.

Why assembly, and why a program.  cbca_hwd_kernel (cbca_hwd.hip, round 3) walks the support regions of a K x G patch of
anchors with scalar code: per region row it decodes arms, tests which anchors take part, dispatches the window loads
slot by slot and guards every chain element with a scalar bit test + branch - ~240 scalar instructions and ~25 taken
branches per row for ~120 vector adds, in 165 registers (a statically indexed 31-slot register window), three waves
per SIMD.  None of that control depends on the disparity or on the iteration: it is a function of the image alone and
is the same for all 18 aggregation iterations of a pair and for every disparity chunk.  So it is compiled ONCE per
image (cbca_prog_build, cbca_prog.hip) into a linear program per patch, and the kernel here is a threaded-code
interpreter of that program:

  op (32 bit) = [15:0] byte offset of its handler from `code_base`, [31:16] parameter (lands in M0)
  LOAD  n, p   n consecutive pixels ending at pixel index p (relative to the patch's first region row) -> window
               slots n-1 .. 0: computed entry into a straight line of buffer_load_dwordx{VPL}, then s_waitcnt vmcnt(0)
  WAIT  k      s_waitcnt vmcnt(k) (not emitted by the current builder)
  ADD   s, n   one arm of one anchor column (for anchor row 0, 1 or both): n window slots added in order into the
               anchors' accumulators.  The window is addressed RELATIVELY (VGPR index mode, M0 = parameter), so one
               straight line of v_add_f32 per (column, anchor set, direction) serves every arm: descending arms
               (self, left 1, 2, ...) enter a line whose register numbers fall, ascending arms (right 1, 2, ...) one
               whose numbers rise, at the block that leaves exactly n adds to the end of the line; M0 shifts the line
               onto the slots of this arm.  No test or branch per element, no instruction for anchors that sit a row out.
  REFILL       the next 64 ops (ops live one per lane in a VGPR; v_readlane fetches op i)
  END          divide by the region sizes, store (pf:161)

Every op ends in the same 7-instruction dispatcher (fetch, decode, s_setpc).  The same additions in the same order per
(pixel, disparity) as the reference: bit-exact.  Relative addressing shrinks the window from 31 statically indexed
slots to the W the program is built for (12): 96 registers, five waves per SIMD.

The file doubles as the description of the code layout for the program builder (`layout()` -> cbca_prog_layout.h) and
for the instruction-level simulator the CPU tests run the kernel on (tools/asm_sim.py).

    python cbca_prog_gen.py --vpl 4 -o cbca_prog_v4.s --header cbca_prog_layout_v4.h
"""
import argparse
import re
import sys

R = 13                     # longest arm (distance threshold L <= 14)
KDROP = 0x7ffffff0         # byte offset past every buffer: the range check drops the access
M0_SRC1 = 0x2000           # M0[15:12] = 2: VGPR index mode applies to SRC1 only (the window operand)


class Ins:
    __slots__ = ("op", "args", "mods", "comment")

    def __init__(self, op, args, mods=None, comment=None):
        self.op, self.args, self.mods, self.comment = op, list(args), dict(mods or {}), comment

    def __repr__(self):
        return self.render()

    def render(self):
        if self.op == "label":
            return "%s:" % self.args[0]
        a = ", ".join(_fmt(x) for x in self.args)
        m = ""
        for k, v in self.mods.items():
            if k in ("sim_skip", "dpp"):
                continue
            if v is True:
                m += " " + k
            elif v is not None and v is not False:
                m += " %s:%s" % (k, v)
        s = "  %s %s%s" % (self.op, a, m)
        if self.comment:
            s = "%-72s // %s" % (s, self.comment)
        return s

    def size(self):
        return ins_size(self)


def _fmt(x):
    if isinstance(x, int):
        return str(x) if -16 <= x <= 64 else "0x%x" % (x & 0xffffffff)
    if isinstance(x, float):
        return repr(x)
    return str(x)


SOPP = {"s_waitcnt", "s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_nop",
        "s_endpgm", "s_set_gpr_idx_off", "s_barrier", "s_cbranch_execz"}
SOPK = {"s_movk_i32"}
# (s_bfe_u32 with its literal is 8 bytes by the generic rule)
SMEM = {"s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16"}
VOP3 = {"v_cmp_lt_f32_e64", "v_cmp_eq_f32_e64", "v_pk_add_f32", "v_readlane_b32", "v_fma_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mul_lo_u32",
        "v_mad_u32_u24", "v_cndmask_b32_e64", "v_cmp_lt_f32_e64", "v_cmp_eq_f32_e64", "v_cmp_gt_u32_e64",
        "v_min3_f32", "v_lshl_add_u32"}
DS = {"ds_read_b32", "ds_read_b64", "ds_read_b96", "ds_read_b128"}
MUBUF = {"buffer_load_dword", "buffer_load_dwordx2", "buffer_load_dwordx3", "buffer_load_dwordx4",
         "buffer_store_dword", "buffer_store_dwordx2", "buffer_store_dwordx3", "buffer_store_dwordx4"}
FLOAT_INLINE = {0.0, 0.5, 1.0, 2.0, 4.0, -0.5, -1.0, -2.0, -4.0}


def _is_literal(x):
    if isinstance(x, int):
        return not (-16 <= x <= 64)
    if isinstance(x, float):
        return x not in FLOAT_INLINE
    return False


def _is_label_diff(x):
    """An assemble-time constant `labelA-labelB` (a 32-bit literal in the encoding)."""
    return isinstance(x, str) and re.match(r"^[A-Za-z_]\w*-[A-Za-z_]\w*$", x) is not None


def ins_size(i):
    """Encoded size in bytes (checked against llvm-objdump by the build, see check_layout)."""
    if i.op == "label" or i.op.startswith("pseudo_"):
        return 0
    if i.op in SOPP or i.op in SOPK:
        return 4
    if i.op in SMEM or i.op in VOP3 or i.op in MUBUF or i.op in DS:
        return 8
    if i.mods.get("dpp"):
        return 8
    if i.op == "s_set_gpr_idx_on":
        return 4
    return 8 if any(_is_literal(a) or _is_label_diff(a) for a in i.args) else 4


class Params:
    def __init__(self, vpl=4, K=2, G=5, W=12, NB=1, PF=0, wta=False, debug=0, order=0, minvgpr=0, skip=False, ring=0, persist=False,
                 pipe=0, ntload=False, early=False, tile=0, refresh=False):
        assert K in (1, 2, 4, 8), "anchor sets are aligned power-of-two groups of anchor rows"
        # refresh (round 6): the kernel of an aggregation's FIRST iteration when later ones run the skip programs.  It is
        # the full kernel, and it also stores the quotient of every anchor whose support region is the pixel itself -
        # v1 = (0 + v0) / 1 - back into the INPUT buffer.  After it both buffers of the ping-pong pair hold v1 at those
        # pixels, so EVERY later iteration may leave them alone, whichever buffer the last one writes (without it the
        # input buffer keeps v0, which differs from v1 where v0 is -0.0, and an aggregation with an even number of
        # iterations had to end with a full launch).  The race with waves that read such a pixel as a neighbour is
        # benign: as an operand of a sum that began as 0 + x, v0 and v1 give the same bits.
        self.refresh = refresh
        assert not (refresh and (skip or wta)), "refresh is a variant of the plain full kernel"
        # tile (round 6): a workgroup of `tile` waves owns `tile` horizontally adjacent patches and sweeps their region rows
        # in lock step: a STEP op (every wave of the tile has the same ones at the same places) brings the union of the
        # tile's horizontal arms in one region row into LDS ONCE - every wave requests its share with buffer_load ... lds,
        # between two barriers - and the LOADL ops that follow copy a patch's window out of LDS into the register window
        # (ds_read_b128), where the unchanged ADD lines find it.  The bytes through the vector memory pipe per anchor
        # fall to (5 tile + 2 a) / (tile (5 + 2 a)) for arms of length a; the window loads become LDS reads.
        self.tile = tile
        assert not tile or not (pipe or ring or persist or NB != 1 or PF or early or ntload), "tile: the plain one-window kernel"
        # ntload: a second LOAD line (n <= G) whose loads carry the non-temporal hint: for region rows that consist of
        # unit-region pixels only (nobody else's region holds them, so keeping them in L2 only evicts lines that
        # neighbouring patches will read again)
        self.ntload = ntload
        # early (round 5, experimental: measured 1.5-2.5 % SLOWER on all four kernels, profiles/r05_cbca_prog_early.txt;
        # off by default): the first op of a program - its first LOAD, or END when the program is empty - is fetched by a
        # scalar load beside the vector load of the program's first 64 ops and dispatched as soon as it is there: the
        # first window is requested while the rest of the program is still on its way (a wave no longer waits for its
        # program AND THEN for its first window), and the wave of an empty skip program ends without waiting for
        # anything but that one word.  The first LOAD runs on a copy of the load line whose dispatcher waits BEFORE it
        # reads the next op out of the program register.
        self.early = bool(early) and not (pipe or ring or persist or NB != 1)
        assert K * G <= 20, "the region sizes of the K x G anchors live in s[16:35]"
        self.VPL, self.K, self.G, self.W, self.NB, self.PF, self.wta, self.debug = vpl, K, G, W, NB, PF, wta, debug
        self.order = order                          # 0: column-group-major dispatch inside an XCD's band, 1: row-group-major
        # skip: the kernel for the "skip" programs (cbca_prog_build.h, unit_region): an anchor whose support region is the
        # pixel itself - all four arms 0 in its support word - is neither divided nor stored (its value already stands in
        # the output buffer: the caller's promise, include/mccnn.h)
        self.skip = skip
        assert not (skip and wta), "the last iteration needs every pixel's values for the WTA: it runs the full programs"
        # pipe (round 5): the window is a ring of W slots that the PROGRAM manages.  A LOAD op names its slots (entry
        # block = highest slot, count, byte offset of the last pixel in the op's second word) and does not wait; the
        # builder issues the loads of as many upcoming units as the ring holds and puts a WAIT k (k = slots requested
        # after the unit that is due) in front of each unit's arms: several region rows in flight per wave, the memory
        # round trips of narrow rows overlap instead of following one another.  `pipe` = the widest unit the builder may
        # plan (a unit wider than about half the ring would serialise the pipeline again).
        self.pipe = pipe
        assert not pipe or (NB == 1 and not ring and not PF and pipe <= W), "pipe: one ring of W slots"
        self.UW = pipe if pipe else W               # widest window one unit may ask for
        self.MAXD = min(self.UW, R + 1)             # longest descending run one op can carry (self + left arm)
        self.MAXA = min(self.UW, R)                 # longest ascending run
        self.RS = vpl + (vpl & 1)                   # registers per slot / accumulator: gfx950 wants even-aligned tuples
        self.nacc = K * G * self.RS
        # VGPR map: accumulators, four service registers (six with the ops' second words), window
        self.v_voff, self.v_progA, self.v_progB, self.v_lane4 = self.nacc, self.nacc + 1, self.nacc + 2, self.nacc + 3
        self.v_auxA, self.v_auxB = self.nacc + 4, self.nacc + 5
        self.PHYS_WIN = max(self.nacc + (6 if pipe else 4), self.RS * (self.MAXA - 1))
        self.PHYS_WIN = (self.PHYS_WIN + 3) & ~3
        self.nvgpr = self.PHYS_WIN + NB * W * self.RS      # NB windows: the next one loads under the current one's adds
        self.nvgpr_alloc = max(self.nvgpr, minvgpr)        # experiments: a larger allocation = fewer waves per SIMD
        self.NWAIT = W + 1                          # WAIT handlers 0 .. W
        # ring (experimental, tools/dev_prog_check.py --ring S): the window rows travel through an LDS ring of S 1-KiB
        # slots per wave - PF ops stream upcoming units into it with buffer_load ... lds (no registers), CP ops copy a
        # unit into the register window when its turn comes: several windows in flight per wave
        self.ring = ring
        # persist (experimental, with ring): 8 x NWC x band_groups waves, each walks the column groups cg0, cg0 + NWC, ..
        # of its own row group for every (image, chunk); the next patch's program is fetched while the current one ends
        self.persist = persist
        assert not persist or (ring and not wta), "the persistent loop is built on the ring kernels"
        if ring:
            assert NB == 1 and ring >= W, "a unit must fit the ring"
            self.NWAIT = 64
            self.v_lane16, self.v_lds = self.nvgpr, self.nvgpr + 1
            self.nvgpr += 2
            if persist:
                self.v_nextA, self.v_nextB = self.nvgpr, self.nvgpr + 1
                self.nvgpr += 2
            self.nvgpr_alloc = max(self.nvgpr, minvgpr)
        if tile:
            self.SLOTS = tile * G + 2 * R                 # LDS slots of a region row: the tile's columns + both longest arms
            self.SB = 256 * vpl                           # bytes of a slot (64 lanes x vpl floats)
            # buffer_load ... lds exists for 4, 12 and 16 bytes per lane: a request is always 64 lanes x 16 bytes = 1 KiB,
            # i.e. SPD = 1 slot at four disparities per lane, 2 consecutive slots (pixels) at two
            assert vpl in (2, 4), "tile: 16-byte LDS requests cover whole slots at 2 or 4 disparities per lane"
            self.SPD = 1024 // self.SB
            self.LDS_BYTES = (self.SLOTS + self.SPD - 1) * self.SB    # (an odd row's last request writes one slot more)
            assert self.LDS_BYTES <= 65536, "the LDS address of a slot travels in M0[15:0]"
            self.v_lane16, self.v_lds, self.v_dma = self.nvgpr, self.nvgpr + 1, self.nvgpr + 2
            self.nvgpr += 3 if self.SPD > 1 else 2
            self.nvgpr_alloc = max(self.nvgpr, minvgpr)

    def acc(self, k, j, c=0):
        return (k * self.G + j) * self.RS + c

    def sets(self):
        """Anchor-row sets an ADD line exists for: aligned groups of 1, 2, 4 .. K rows (bit k = anchor row k)."""
        out, s = [], 1
        while s <= self.K:
            out += [((1 << s) - 1) << st for st in range(0, self.K, s)]
            s *= 2
        return out

    def name(self):
        if self.tile:
            return "mccnn_cbca_tile%d_v%d%s" % (self.tile, self.VPL, "_wta" if self.wta else "_skip" if self.skip else "")
        return "mccnn_cbca_prog_v%d%s%s" % (self.VPL, "p" if self.pipe else "",
                                            "_wta" if self.wta else "_skip" if self.skip else "_refresh" if self.refresh else "")


# ---- scalar register map ------------------------------------------------------------------------------------------
S = dict(
    karg=0,            # s[0:1] kernarg segment
    bx=2, by=3, bz=4,  # workgroup ids
    # kernarg block 1 (0x40..0x5f): Dp H W nchunks band_rows band_groups prog_stride_bytes ngroups
    Dp=8, H=9, W=10, nchunks=11, band_rows=12, band_groups=13, prog_stride=14, ngroups=15,
    # kernarg block 0 (0x00..0x3f): in0 in1 out0 out1 prog0 prog1 sup0 sup1 - dead once the job's pointers are picked
    ka=16,             # s[16:31]
    cnt=16,            # s[16 : 16 + K*G] in the END handler: region sizes of the anchors (K*G <= 20)
    y0=40, x0=41, job=42, chunk=43,
    inp=44,            # s[44:45]
    outp=46,           # s[46:47]
    supp=48,           # s[48:49]
    progp=50,          # s[50:51]
    rs_in=52,          # s[52:55]
    rs_prog=56,        # s[56:59]
    pix=60, so=61, i=62, op=63, t=64,
    base=66,           # s[66:67]
    pc=68,             # s[68:69]
    safe_m0=70, progoff=71,
    t0=72, t1=73, t2=74, t3=75, t4=76, t5=77,
    rs_out=80,         # s[80:83]
    # kernarg block 2 (0x60..0x7f, WTA kernel): disp0 disp1 D store1
    disp=84,           # s[84:87]
    D=88, store1=89,
    dispp=96,          # s[96:97]
    dump=98, pfoff=99,
    ring_head=90, ring_tail=91, ring_t=92,
    nwc=93, cg=94, z=95, rg=36, cg0=37, prg=38, guard=39, band=78, cgn=79,
    rs_pn=84,          # s[84:87] (persist): descriptor of the next patch's program
    pfa=100,           # s[100:101]
    wave=90, st_n=91, st_lds=92, st_s=93, st_str=94,      # (tile) wave of the workgroup; STEP's loop state
    rs_ref=84,         # s[84:87] (refresh): descriptor over the anchors' row of the INPUT buffer
)
NSGPR = 102


def decompose(aset, K):
    """An arbitrary set of anchor rows as the aligned power-of-two groups the kernel has lines for, largest first."""
    out, s = [], K
    while s >= 1 and aset:
        for st in range(0, K, s):
            m = ((1 << s) - 1) << st
            if aset & m == m:
                out.append(m)
                aset &= ~m
        s //= 2
    return out


def sreg(n, cnt=1):
    return "s%d" % n if cnt == 1 else "s[%d:%d]" % (n, n + cnt - 1)


def vreg(n, cnt=1):
    return "v%d" % n if cnt == 1 else "v[%d:%d]" % (n, n + cnt - 1)


class Gen:
    def __init__(self, P):
        self.P = P
        self.ins = []
        self.labels = {}

    def e(self, op, *args, comment=None, **mods):
        self.ins.append(Ins(op, args, mods, comment))

    def label(self, name):
        self.ins.append(Ins("label", [name]))

    # -- the dispatcher every handler ends in -------------------------------------------------------------------------
    def tail(self, wait=None):
        """wait: an s_waitcnt that may sit behind the decode of the next op (the LOAD handler's vmcnt(0))."""
        e = self.e
        e("v_readlane_b32", sreg(S["op"]), vreg(self.P.v_progA), sreg(S["i"]))
        e("s_add_u32", sreg(S["i"]), sreg(S["i"]), 1)
        if self.P.ring:       # (experimental kernels are larger than 32 KiB: handler offsets in dwords, unsigned)
            e("s_bfe_u32", sreg(S["t"]), sreg(S["op"]), 0x100000)
            e("s_lshl_b32", sreg(S["t"]), sreg(S["t"]), 2)
        else:
            e("s_sext_i32_i16", sreg(S["t"]), sreg(S["op"]))
        e("s_lshr_b32", "m0", sreg(S["op"]), 16)
        e("s_add_u32", sreg(S["pc"]), sreg(S["base"]), sreg(S["t"]))
        e("s_addc_u32", sreg(S["pc"] + 1), sreg(S["base"] + 1), 0)
        if wait:
            e("s_waitcnt", wait)
        e("s_setpc_b64", sreg(S["pc"], 2))

    def vload(self, dst, voff, rs, soff, **mods):
        n = self.P.VPL
        op = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 3: "buffer_load_dwordx3", 4: "buffer_load_dwordx4"}[n]
        self.e(op, vreg(dst, n), vreg(voff), sreg(rs, 4), soff, offen=True, **mods)

    def vstore(self, src, voff, rs, soff, **mods):
        n = self.P.VPL
        op = {1: "buffer_store_dword", 2: "buffer_store_dwordx2", 3: "buffer_store_dwordx3", 4: "buffer_store_dwordx4"}[n]
        self.e(op, vreg(src, n), vreg(voff), sreg(rs, 4), soff, offen=True, **mods)

    # -- kernel ----------------------------------------------------------------------------------------------------------
    def build(self):
        P, e = self.P, self.e
        K, G, VPL, W = P.K, P.G, P.VPL, P.W
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        self.label("entry")
        if P.persist:
            self.prologue_persist()
        else:
            self.prologue_once()
        self.label("code_base")
        self.handlers()
        self.end_handler()
        return self

    def prologue_persist(self):
        P, e = self.P, self.e
        K, G, VPL, W = P.K, P.G, P.VPL, P.W
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        e("s_load_dwordx8", s("Dp", 8), s("karg", 2), 0x40)
        e("s_load_dword", s("nwc"), s("karg", 2), 0x60)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_max_u32", s("nwc"), s("nwc"), 1)
        e("s_and_b32", s("band"), s("bx"), 7)
        e("s_lshr_b32", s("rg"), s("bx"), 3)
        e("s_mov_b32", s("cg0"), 0)
        self.label("p_div")                                              # rg = wl % band_groups, cg0 = wl / band_groups
        e("s_cmp_lt_u32", s("rg"), s("band_groups"))
        e("s_cbranch_scc1", "p_divd")
        e("s_sub_u32", s("rg"), s("rg"), s("band_groups"))
        e("s_add_u32", s("cg0"), s("cg0"), 1)
        e("s_branch", "p_div")
        self.label("p_divd")
        e("s_mul_i32", s("y0"), s("band"), s("band_rows"))
        e("s_mul_i32", s("t2"), s("rg"), K)
        e("s_add_u32", s("y0"), s("y0"), s("t2"))
        e("s_cmp_ge_i32", s("y0"), s("H"))
        e("s_cbranch_scc1", "done")
        e("s_mul_i32", s("prg"), s("band"), s("band_groups"))
        e("s_add_u32", s("prg"), s("prg"), s("rg"), comment="patch row group")
        e("v_lshlrev_b32", vreg(P.v_lane4), 2, "v0")
        e("v_lshlrev_b32", vreg(P.v_lane16), 4, "v0")
        e("s_lshl_b32", s("pix"), s("Dp"), 2)
        e("s_getpc_b64", s("base", 2))
        self.label("after_getpc")
        e("s_add_u32", s("base"), s("base"), "code_base-after_getpc")
        e("s_addc_u32", sreg(S["base"] + 1), sreg(S["base"] + 1), 0)
        e("s_mov_b32", s("safe_m0"), M0_SRC1)
        e("s_mov_b32", s("z"), 0)
        # a hard bound on the patches a wave may walk: (ngroups + 1) * 2 * nchunks
        e("s_add_u32", s("guard"), s("ngroups"), 1)
        e("s_mul_i32", s("guard"), s("guard"), s("nchunks"))
        e("s_lshl_b32", s("guard"), s("guard"), 1)
        # ---- per (image, chunk) ------------------------------------------------------------------------------------------
        self.label("p_z")
        e("s_lshl_b32", s("t0"), s("nchunks"), 1)
        e("s_cmp_ge_u32", s("z"), s("t0"))
        e("s_cbranch_scc1", "done")
        e("s_load_dwordx16", s("ka", 16), s("karg", 2), 0x0)
        e("s_waitcnt", "lgkmcnt(0)")
        e("s_cmp_ge_u32", s("z"), s("nchunks"))
        e("s_cselect_b32", s("job"), 1, 0)
        e("s_cselect_b32", s("t1"), s("nchunks"), 0)
        e("s_cselect_b64", s("inp", 2), sreg(S["ka"] + 2, 2), sreg(S["ka"] + 0, 2))
        e("s_cselect_b64", s("outp", 2), sreg(S["ka"] + 6, 2), sreg(S["ka"] + 4, 2))
        e("s_cselect_b64", s("progp", 2), sreg(S["ka"] + 10, 2), sreg(S["ka"] + 8, 2))
        e("s_cselect_b64", s("supp", 2), sreg(S["ka"] + 14, 2), sreg(S["ka"] + 12, 2))
        e("s_sub_u32", s("chunk"), s("z"), s("t1"))
        # voff = d0 * 4 with d0 = (chunk * 64 + lane) * VPL, or kDrop past the disparity range
        e("s_lshl_b32", s("t1"), s("chunk"), 6)
        vt = P.PHYS_WIN
        e("v_lshrrev_b32", vreg(vt), 2, vreg(P.v_lane4), comment="lane id")
        e("v_add_u32", vreg(vt), s("t1"), vreg(vt))
        e("v_mul_u32_u24", vreg(vt), 4 * VPL, vreg(vt))
        e("v_mov_b32", vreg(P.v_voff), KDROP)
        e("v_cmp_gt_u32", "vcc", s("pix"), vreg(vt))
        e("v_cndmask_b32", vreg(P.v_voff), vreg(P.v_voff), vreg(vt), "vcc")
        # input rows any arm of this wave's patches can reach (the same for every column group)
        e("s_sub_u32", s("t0"), s("y0"), R)
        e("s_max_i32", s("t0"), s("t0"), 0, comment="row0")
        e("s_sub_u32", s("t3"), s("H"), 1)
        e("s_add_u32", s("t1"), s("y0"), K - 1)
        e("s_min_i32", s("t1"), s("t1"), s("t3"))
        e("s_add_u32", s("t1"), s("t1"), R)
        e("s_min_i32", s("t1"), s("t1"), s("t3"), comment="row1")
        e("s_sub_u32", s("t1"), s("t1"), s("t0"))
        e("s_add_u32", s("t1"), s("t1"), 1, comment="rows")
        e("s_mul_i32", s("t2"), s("W"), s("pix"), comment="bytes per image row")
        e("s_mul_i32", sreg(S["rs_in"] + 2), s("t1"), s("t2"))
        e("s_mul_hi_u32", s("t4"), s("t0"), s("t2"))
        e("s_mul_i32", s("t3"), s("t0"), s("t2"))
        e("s_add_u32", s("rs_in"), s("inp"), s("t3"))
        e("s_addc_u32", sreg(S["rs_in"] + 1), sreg(S["inp"] + 1), s("t4"))
        e("s_and_b32", sreg(S["rs_in"] + 1), sreg(S["rs_in"] + 1), 0xffff)
        e("s_mov_b32", sreg(S["rs_in"] + 3), 0x00020000)
        # the first patch of this (image, chunk): its program is fetched here, the later ones at the end of their forerunner
        e("s_mov_b32", s("cg"), s("cg0"))
        e("s_cmp_ge_u32", s("cg"), s("ngroups"))
        e("s_cbranch_scc1", "p_nextz")
        e("s_mov_b32", s("cgn"), s("cg"))
        self.next_program_fetch()
        self.label("p_patch")                                             # rs_pn / cgn describe the patch that starts now
        e("s_sub_u32", s("guard"), s("guard"), 1)
        e("s_cmp_le_i32", s("guard"), 0)
        e("s_cbranch_scc1", "done")
        e("s_mov_b32", s("cg"), s("cgn"))
        e("s_mul_i32", s("x0"), s("cg"), G)
        for q in range(4):
            e("s_mov_b32", sreg(S["rs_prog"] + q), sreg(S["rs_pn"] + q))
        self.cnt_loads()
        for r in range(P.nacc):
            e("v_mov_b32", vreg(r), 0)
        e("s_movk_i32", s("progoff"), 512)
        e("s_mov_b32", s("ring_head"), 0)
        e("s_mov_b32", s("ring_tail"), 0)
        e("s_mov_b32", s("i"), 0)
        e("s_waitcnt", "vmcnt(0)")
        e("v_mov_b32", vreg(P.v_progA), vreg(P.v_nextA))
        e("v_mov_b32", vreg(P.v_progB), vreg(P.v_nextB))
        e("s_set_gpr_idx_on", s("i"), "gpr_idx(SRC1)")
        self.tail()
        self.label("p_nextz")
        e("s_add_u32", s("z"), s("z"), 1)
        e("s_branch", "p_z")

    def next_program_fetch(self):
        """Descriptor rs_pn of patch (prg, cgn) and its first two 64-op chunks into nextA / nextB."""
        P, e = self.P, self.e
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        e("s_mul_i32", s("t0"), s("prg"), s("ngroups"))
        e("s_add_u32", s("t0"), s("t0"), s("cgn"), comment="patch index")
        e("s_mul_hi_u32", s("t2"), s("t0"), s("prog_stride"))
        e("s_mul_i32", s("t1"), s("t0"), s("prog_stride"))
        e("s_add_u32", s("rs_pn"), s("progp"), s("t1"))
        e("s_addc_u32", sreg(S["rs_pn"] + 1), sreg(S["progp"] + 1), s("t2"))
        e("s_and_b32", sreg(S["rs_pn"] + 1), sreg(S["rs_pn"] + 1), 0xffff)
        e("s_mov_b32", sreg(S["rs_pn"] + 2), s("prog_stride"))
        e("s_mov_b32", sreg(S["rs_pn"] + 3), 0x00020000)
        e("buffer_load_dword", vreg(P.v_nextA), vreg(P.v_lane4), s("rs_pn", 4), 0, offen=True)
        e("buffer_load_dword", vreg(P.v_nextB), vreg(P.v_lane4), s("rs_pn", 4), 0, offen=True, offset=256)

    def cnt_loads(self):
        P, e = self.P, self.e
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        e("s_sub_u32", s("t3"), s("H"), 1)
        for k in range(P.K):
            e("s_add_u32", s("t0"), s("y0"), k)
            e("s_min_i32", s("t0"), s("t0"), s("t3"))
            e("s_mul_i32", s("t0"), s("t0"), s("W"))
            e("s_add_u32", s("t0"), s("t0"), s("x0"))
            e("s_lshl_b32", s("t0"), s("t0"), 2)
            e("s_add_u32", s("t4"), s("supp"), s("t0"))
            e("s_addc_u32", s("t5"), sreg(S["supp"] + 1), 0)
            for j in range(P.G):
                e("s_load_dword", sreg(S["cnt"] + k * P.G + j), sreg(S["t4"], 2), 4 * j)

    def prologue_once(self):
        P, e = self.P, self.e
        K, G, VPL, W = P.K, P.G, P.VPL, P.W
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        e("s_load_dwordx16", s("ka", 16), s("karg", 2), 0x0)
        e("s_load_dwordx8", s("Dp", 8), s("karg", 2), 0x40)
        if P.wta:
            e("s_load_dwordx8", s("disp", 8), s("karg", 2), 0x60)
        if P.tile:            # v0 = work-item id of the workgroup: wave = v0 >> 6 (uniform), v0 = lane from here on
            e("v_lshrrev_b32", vreg(P.v_lds), 6, "v0")
            e("s_nop", 4, comment="gfx940+: a v_readlane / v_readfirstlane may not follow the VALU write of its source at once")
            e("v_readfirstlane_b32", s("wave"), vreg(P.v_lds))
            e("v_and_b32", "v0", 63, "v0")
        e("s_waitcnt", "lgkmcnt(0)")
        # y0 = (bx & 7) * band_rows + (bx >> 3) * K ; patch row group = (bx & 7) * band_groups + (bx >> 3)
        e("s_and_b32", s("t0"), s("bx"), 7)
        if P.order == 0:      # an XCD sweeps its band column group by column group: x = (band, row group), y = column group
            e("s_lshr_b32", s("t1"), s("bx"), 3)
            e("s_mov_b32", s("t3"), s("by"))
        else:                 # ... row group by row group: x = (band, column group), y = row group
            e("s_lshr_b32", s("t3"), s("bx"), 3)
            e("s_mov_b32", s("t1"), s("by"))
        if P.tile:            # the grid counts tiles: this wave's column group = tile * waves + wave (ngroups is padded to whole tiles)
            e("s_mul_i32", s("t3"), s("t3"), P.tile)
            e("s_add_u32", s("t3"), s("t3"), s("wave"))
        e("s_mul_i32", s("y0"), s("t0"), s("band_rows"))
        e("s_mul_i32", s("t2"), s("t1"), K)
        e("s_add_u32", s("y0"), s("y0"), s("t2"))
        e("s_cmp_ge_i32", s("y0"), s("H"))
        e("s_cbranch_scc1", "done")
        e("s_mul_i32", s("t0"), s("t0"), s("band_groups"))
        e("s_add_u32", s("t0"), s("t0"), s("t1"))                        # patch row group
        e("s_mul_i32", s("t0"), s("t0"), s("ngroups"))
        e("s_add_u32", s("t0"), s("t0"), s("t3"), comment="patch index")
        e("s_mul_i32", s("x0"), s("t3"), G)
        # job = bz >= nchunks, chunk = bz - job * nchunks; pick the job's pointers
        e("s_cmp_ge_u32", s("bz"), s("nchunks"))
        e("s_cselect_b32", s("job"), 1, 0)
        e("s_cselect_b32", s("t1"), s("nchunks"), 0)
        e("s_cselect_b64", s("inp", 2), sreg(S["ka"] + 2, 2), sreg(S["ka"] + 0, 2))
        e("s_cselect_b64", s("outp", 2), sreg(S["ka"] + 6, 2), sreg(S["ka"] + 4, 2))
        e("s_cselect_b64", s("progp", 2), sreg(S["ka"] + 10, 2), sreg(S["ka"] + 8, 2))
        e("s_cselect_b64", s("supp", 2), sreg(S["ka"] + 14, 2), sreg(S["ka"] + 12, 2))
        if P.wta:
            e("s_cselect_b64", s("dispp", 2), sreg(S["disp"] + 2, 2), sreg(S["disp"] + 0, 2))
            e("s_cselect_b32", s("store1"), s("store1"), 1, comment="job 0 is always stored, job 1 if store1")
        e("s_sub_u32", s("chunk"), s("bz"), s("t1"))
        # program of this patch: one dword per lane, two 64-op chunks in flight
        e("s_mul_hi_u32", s("t2"), s("t0"), s("prog_stride"))
        e("s_mul_i32", s("t1"), s("t0"), s("prog_stride"))
        e("s_add_u32", s("rs_prog"), s("progp"), s("t1"))
        e("s_addc_u32", sreg(S["rs_prog"] + 1), sreg(S["progp"] + 1), s("t2"))
        e("s_and_b32", sreg(S["rs_prog"] + 1), sreg(S["rs_prog"] + 1), 0xffff)
        e("s_mov_b32", sreg(S["rs_prog"] + 2), s("prog_stride"))
        e("s_mov_b32", sreg(S["rs_prog"] + 3), 0x00020000)
        e("v_lshlrev_b32", vreg(P.v_lane4), 2, "v0", comment="lane * 4 (v0 becomes an accumulator)")
        if P.pipe:        # a 64-op chunk is 512 bytes: the ops, then their second words
            e("buffer_load_dword", vreg(P.v_progA), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True)
            e("buffer_load_dword", vreg(P.v_auxA), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True, offset=256)
            e("buffer_load_dword", vreg(P.v_progB), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True, offset=512)
            e("buffer_load_dword", vreg(P.v_auxB), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True, offset=768)
            e("s_movk_i32", s("progoff"), 1024)
        else:
            e("buffer_load_dword", vreg(P.v_progA), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True)
            e("buffer_load_dword", vreg(P.v_progB), vreg(P.v_lane4), s("rs_prog", 4), 0, offen=True, offset=256)
            e("s_movk_i32", s("progoff"), 512)
            if P.early:
                e("s_load_dword", s("op"), s("rs_prog", 2), 0, comment="the program's first op (descriptor words 0-1 = its address)")
        if P.ring:
            e("v_lshlrev_b32", vreg(P.v_lane16), 4, "v0", comment="lane * 16: this lane's bytes of an LDS slot")
            e("s_mov_b32", s("ring_head"), 0)
            e("s_mov_b32", s("ring_tail"), 0)
        if P.tile:
            e("v_mul_u32_u24", vreg(P.v_lane16), 4 * VPL, "v0", comment="this lane's bytes of an LDS slot")
        # region sizes of the K x G anchors for the END handler (rows clamped to the image; words past the right edge are
        # never used), requested here so that they arrive under the program: the kernarg registers they land in are dead
        e("s_sub_u32", s("t3"), s("H"), 1)
        for k in range(K):
            e("s_add_u32", s("t0"), s("y0"), k)
            e("s_min_i32", s("t0"), s("t0"), s("t3"))
            e("s_mul_i32", s("t0"), s("t0"), s("W"))
            e("s_add_u32", s("t0"), s("t0"), s("x0"))
            e("s_lshl_b32", s("t0"), s("t0"), 2)                            # H * W * 4 < 2^31 (checked by the host)
            e("s_add_u32", s("t4"), s("supp"), s("t0"))
            e("s_addc_u32", s("t5"), sreg(S["supp"] + 1), 0)
            for j in range(G):
                e("s_load_dword", sreg(S["cnt"] + k * G + j), sreg(S["t4"], 2), 4 * j)
        # pix = Dp * 4 ; voff = d0 * 4 with d0 = (chunk * 64 + lane) * VPL, or kDrop past the disparity range
        e("s_lshl_b32", s("pix"), s("Dp"), 2)
        e("s_lshl_b32", s("t1"), s("chunk"), 6)
        vt = P.PHYS_WIN                                                 # a window register as scratch
        e("v_add_u32", vreg(vt), s("t1"), "v0")
        e("v_mul_u32_u24", vreg(vt), 4 * VPL, vreg(vt))
        e("v_mov_b32", vreg(P.v_voff), KDROP)
        e("v_cmp_gt_u32", "vcc", s("pix"), vreg(vt))
        e("v_cndmask_b32", vreg(P.v_voff), vreg(P.v_voff), vreg(vt), "vcc")
        if P.tile and P.SPD > 1:
            # LDS requests are 16 bytes per lane: lanes 0-31 fetch pixel p, lanes 32-63 pixel p + 1 (two consecutive slots)
            e("v_and_b32", vreg(vt), 31, "v0")
            e("v_lshlrev_b32", vreg(vt), 4, vreg(vt))
            e("s_mul_i32", s("t1"), s("chunk"), 64 * 4 * VPL)
            e("v_add_u32", vreg(vt), s("t1"), vreg(vt), comment="byte offset inside the pixel's record")
            e("v_cmp_gt_u32", "vcc", s("pix"), vreg(vt))
            e("v_lshrrev_b32", vreg(P.v_dma), 5, "v0")
            e("v_mul_u32_u24", vreg(P.v_dma), s("pix"), vreg(P.v_dma))
            e("v_add_u32", vreg(P.v_dma), vreg(P.v_dma), vreg(vt))
            e("v_mov_b32", vreg(vt), KDROP)
            e("v_cndmask_b32", vreg(P.v_dma), vreg(vt), vreg(P.v_dma), "vcc")
        if P.PF:
            self.prefetch()
        # input rows row0 .. row1 any arm of this patch can reach: descriptor base = in + row0 * W * pix
        e("s_sub_u32", s("t0"), s("y0"), R)
        e("s_max_i32", s("t0"), s("t0"), 0, comment="row0")
        e("s_sub_u32", s("t3"), s("H"), 1)
        e("s_add_u32", s("t1"), s("y0"), K - 1)
        e("s_min_i32", s("t1"), s("t1"), s("t3"))
        e("s_add_u32", s("t1"), s("t1"), R)
        e("s_min_i32", s("t1"), s("t1"), s("t3"), comment="row1")
        e("s_sub_u32", s("t1"), s("t1"), s("t0"))
        e("s_add_u32", s("t1"), s("t1"), 1, comment="rows")
        e("s_mul_i32", s("t2"), s("W"), s("pix"), comment="bytes per image row")
        e("s_mul_i32", sreg(S["rs_in"] + 2), s("t1"), s("t2"))
        e("s_mul_hi_u32", s("t4"), s("t0"), s("t2"))
        e("s_mul_i32", s("t3"), s("t0"), s("t2"))
        e("s_add_u32", s("rs_in"), s("inp"), s("t3"))
        e("s_addc_u32", sreg(S["rs_in"] + 1), sreg(S["inp"] + 1), s("t4"))
        e("s_and_b32", sreg(S["rs_in"] + 1), sreg(S["rs_in"] + 1), 0xffff)
        e("s_mov_b32", sreg(S["rs_in"] + 3), 0x00020000)
        for r in range(P.nacc):
            e("v_mov_b32", vreg(r), 0)                                  # pf:156: the sum starts at 0
        # code_base for the op offsets
        e("s_getpc_b64", s("base", 2))
        self.label("after_getpc")
        e("s_add_u32", s("base"), s("base"), "code_base-after_getpc")
        e("s_addc_u32", sreg(S["base"] + 1), sreg(S["base"] + 1), 0)
        e("s_mov_b32", s("i"), 0)
        e("s_mov_b32", s("safe_m0"), M0_SRC1)
        if not P.early:
            e("s_waitcnt", "vmcnt(0)")
            e("s_set_gpr_idx_on", s("i"), "gpr_idx(SRC1)", comment="index mode on for the whole program: M0 = 0x2000 | idx")
            self.tail()
        else:
            e("s_set_gpr_idx_on", s("i"), "gpr_idx(SRC1)", comment="index mode on for the whole program: M0 = 0x2000 | idx")
            e("s_mov_b32", s("i"), 1, comment="op 0 comes from the scalar load, the dispatchers go on with op 1")
            e("s_waitcnt", "lgkmcnt(0)")
            e("s_sext_i32_i16", s("t"), s("op"))
            e("s_cmp_eq_u32", s("t"), "h_end-code_base")
            e("s_cbranch_scc1", "first_is_end")
            e("s_add_u32", s("t"), s("t"), "h_loadf1-h_load1", comment="the same LOAD on the first-load copy of the line")
            self.label("first_is_end")
            e("s_lshr_b32", "m0", s("op"), 16)
            e("s_add_u32", s("pc"), s("base"), s("t"))
            e("s_addc_u32", sreg(S["pc"] + 1), sreg(S["base"] + 1), 0)
            e("s_setpc_b64", s("pc", 2))

    def handlers(self):
        P, e = self.P, self.e
        K, G, VPL, W = P.K, P.G, P.VPL, P.W
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        # ---- ADD lines: (column j, anchor set 1 = row 0 / 2 = row 1 / 3 = both, direction) ---------------------------
        self.lines = {}
        for j in range(G):
            for aset in P.sets():
                for d, maxn in (("d", P.MAXD), ("a", P.MAXA)):
                    name = "add_j%d_s%d_%s" % (j, aset, d)
                    self.label(name)
                    self.lines[(j, aset, d)] = name
                    for blk in range(maxn, 0, -1):                       # block `blk` leaves blk adds to the end
                        # descending: static slot blk - 1 (numbers fall along the line); ascending: 1 - blk (rise)
                        slot = (blk - 1) if d == "d" else (1 - blk)
                        for k in range(K):
                            if aset & (1 << k):
                                # packed adds: in VGPR index mode a v_add_f32 issues at half rate (4.3 clocks per wave
                                # instruction against 2.4 without; profiles/r04_probe_gpridx_rate.txt), a
                                # v_pk_add_f32 at its full 4.7 for two adds - the same IEEE sums either way
                                c = 0
                                while c + 1 < VPL:
                                    e("v_pk_add_f32", vreg(P.acc(k, j, c), 2), vreg(P.acc(k, j, c), 2),
                                      vreg(P.PHYS_WIN + P.RS * slot + c, 2))
                                    c += 2
                                if c < VPL:
                                    e("v_add_f32", vreg(P.acc(k, j, c)), vreg(P.acc(k, j, c)),
                                      vreg(P.PHYS_WIN + P.RS * slot + c))
                    self.tail()
        # ---- LOAD: entry per (window, n) (parameter -> soffset), then the straight line of loads -----------------------
        # one window: the handler ends with s_waitcnt vmcnt(0) behind the next op's decode; two windows: no wait here,
        # the program says WAIT k (k = loads of the window that was requested last)
        if P.pipe:
            # LOAD k: count - 1 in the parameter, byte offset of the LAST pixel (from the patch's first region row) in the
            # op's second word -> slots k, k - 1, .. (pixels in descending order, like the plain kernel's line); no wait
            for k in range(W):                                          # (the last stub falls into its block)
                self.label("loadk_%d" % k)
                e("s_sub_u32", s("t0"), s("i"), 1, comment="(i already points at the next op)")
                e("v_readlane_b32", s("so"), vreg(P.v_auxA), s("t0"))
                e("s_and_b32", s("t1"), "m0", 0xff)
                e("s_mov_b32", "m0", s("safe_m0"))
                if k != W - 1:
                    e("s_branch", "loadblk_%d" % k)
            for k in range(W - 1, -1, -1):
                self.label("loadblk_%d" % k)
                self.vload(P.PHYS_WIN + P.RS * k, P.v_voff, S["rs_in"], s("so"))
                if k > 0:
                    e("s_sub_u32", s("t1"), s("t1"), 1, comment="SCC = borrow: that was the last pixel")
                    e("s_cbranch_scc1", "load_done")
                    e("s_sub_u32", s("so"), s("so"), s("pix"))
            self.label("load_done")
            self.tail()
        if P.tile:
            SB = P.SB
            # ---- LOADL n: LDS slots p .. p + n - 1 (p = the parameter) -> window slots 0 .. n - 1 --------------------------
            for n in range(1, W + 1):
                self.label("loadl_n%d" % n)
                e("s_lshr_b32", s("t0"), s("op"), 16)
                e("s_mov_b32", "m0", s("safe_m0"), comment="index 0: the v_add below must read v_lane16 itself")
                e("s_mul_i32", s("t0"), s("t0"), SB)
                e("v_add_u32", vreg(P.v_lds), s("t0"), vreg(P.v_lane16))
                if n != W:
                    e("s_branch", "loadl_blk%d" % n)
            rd = {1: "ds_read_b32", 2: "ds_read_b64", 3: "ds_read_b96", 4: "ds_read_b128"}[VPL]
            for n in range(W, 0, -1):
                self.label("loadl_blk%d" % n)
                if not (P.debug & 64):                                   # debug 64 (fault hunting): no LDS reads
                    e(rd, vreg(P.PHYS_WIN + P.RS * (n - 1), VPL), vreg(P.v_lds), offset=(n - 1) * SB)
                else:
                    e("s_nop", 0)
                    e("s_nop", 0)
            self.tail(wait="lgkmcnt(0)")
            # ---- STEP nslots (parameter), p (the next program word: pixel index of LDS slot 0 relative to the patch's first
            # region row): every wave of the workgroup arrives here at the same op of its own program.  Barrier (nobody
            # reads the previous row any more: a LOADL waits for its reads before its arms run), this wave's share of the
            # row - slots wave, wave + NW, .. - straight into LDS, wait, barrier.
            self.label("step")
            e("s_lshr_b32", s("st_n"), s("op"), 16)
            e("v_readlane_b32", s("so"), vreg(P.v_progA), s("i"))
            e("s_add_u32", s("i"), s("i"), 1)
            SPD = P.SPD                                                  # slots one request fills
            e("s_mul_i32", s("st_s"), s("wave"), SPD)
            e("s_add_u32", s("so"), s("so"), s("st_s"))
            e("s_mul_i32", s("so"), s("so"), s("pix"))
            e("s_mul_i32", s("st_lds"), s("st_s"), SB)
            e("s_mul_i32", s("st_str"), s("pix"), P.tile * SPD)
            e("s_set_gpr_idx_off")
            e("s_barrier")
            self.label("step_loop")
            e("s_cmp_ge_u32", s("st_s"), s("st_n"))
            e("s_cbranch_scc1", "step_done")
            e("s_mov_b32", "m0", s("st_lds"))
            if not (P.debug & 32):                                       # debug 32 (fault hunting): no requests
                e("buffer_load_dwordx4", vreg(P.v_voff if SPD == 1 else P.v_dma), s("rs_in", 4), s("so"), offen=True, lds=True)
            else:
                e("s_nop", 0)
                e("s_nop", 0)
            e("s_add_u32", s("st_s"), s("st_s"), P.tile * SPD)
            e("s_add_u32", s("st_lds"), s("st_lds"), P.tile * SPD * SB)
            e("s_add_u32", s("so"), s("so"), s("st_str"))
            e("s_branch", "step_loop")
            self.label("step_done")
            e("s_waitcnt", "vmcnt(0)")
            e("s_barrier")
            e("s_mov_b32", "m0", s("safe_m0"))
            e("s_set_gpr_idx_on", s("st_n"), "gpr_idx(SRC1)", comment="(the dispatcher sets M0 from the next op)")
            e("s_mov_b32", "m0", s("safe_m0"))
            self.tail()
        for b in range(0 if (P.pipe or P.tile) else P.NB):
            for n in range(1, W + 1):
                if b == 0 and n == 1:
                    self.label("h_load1")
                self.label("load_b%d_n%d" % (b, n))
                e("s_mul_i32", s("so"), "m0", s("pix"))
                e("s_mov_b32", "m0", s("safe_m0"))
                if n != W:
                    e("s_branch", "load_b%d_blk%d" % (b, n))
            for n in range(W, 0, -1):
                self.label("load_b%d_blk%d" % (b, n))
                self.vload(P.PHYS_WIN + P.RS * (b * W + n - 1), P.v_voff, S["rs_in"], s("so"))
                if n > 1:
                    e("s_sub_u32", s("so"), s("so"), s("pix"))
            self.tail(wait="vmcnt(0)" if P.NB == 1 else None)
        if P.early:
            # the first LOAD of a program (dispatched by the prologue, see Params.early): same stubs at the same spacing,
            # same line; its dispatcher waits first - the program register it reads op 1 from may still be in flight
            for n in range(1, W + 1):
                self.label("h_loadf1" if n == 1 else "loadf_n%d" % n)
                e("s_mul_i32", s("so"), "m0", s("pix"))
                e("s_mov_b32", "m0", s("safe_m0"))
                if n != W:
                    e("s_branch", "loadf_blk%d" % n)
            for n in range(W, 0, -1):
                self.label("loadf_blk%d" % n)
                self.vload(P.PHYS_WIN + P.RS * (n - 1), P.v_voff, S["rs_in"], s("so"))
                if n > 1:
                    e("s_sub_u32", s("so"), s("so"), s("pix"))
            e("s_waitcnt", "vmcnt(0)")
            self.tail()
        if P.ntload and not P.pipe:
            for n in range(1, G + 1):
                self.label("loadnt_n%d" % n)
                e("s_mul_i32", s("so"), "m0", s("pix"))
                e("s_mov_b32", "m0", s("safe_m0"))
                if n != G:
                    e("s_branch", "loadnt_blk%d" % n)
            for n in range(G, 0, -1):
                self.label("loadnt_blk%d" % n)
                self.vload(P.PHYS_WIN + P.RS * (n - 1), P.v_voff, S["rs_in"], s("so"), nt=True)
                if n > 1:
                    e("s_sub_u32", s("so"), s("so"), s("pix"))
            self.tail(wait="vmcnt(0)")
        # ---- WAIT k ----------------------------------------------------------------------------------------------------------
        for k in range(P.NWAIT):
            self.label("wait_%d" % k)
            e("s_waitcnt", "vmcnt(%d)" % k)
            self.tail()
        if P.ring:
            SB = 256 * VPL                                               # bytes of a ring slot
            RB = P.ring * SB
            # ---- PF n: n consecutive pixels ending at pixel index p (the parameter) -> the next n ring slots.  The ring
            # head is kept here exactly as the builder keeps it: a unit that does not fit behind the head starts at 0.
            for n in range(1, W + 1):
                self.label("pf_n%d" % n)
                e("s_mul_i32", s("so"), "m0", s("pix"))
                e("s_add_u32", s("ring_t"), s("ring_head"), n * SB)
                e("s_cmp_gt_u32", s("ring_t"), RB)
                e("s_cselect_b32", s("ring_head"), 0, s("ring_head"))
                e("s_set_gpr_idx_off")
                e("s_add_u32", "m0", s("ring_head"), (n - 1) * SB, comment="LDS address of slot n - 1")
                e("s_add_u32", s("ring_head"), s("ring_head"), n * SB)
                if n != W:
                    e("s_branch", "pf_blk%d" % n)
            for n in range(W, 0, -1):
                self.label("pf_blk%d" % n)
                op = {1: "buffer_load_dword", 2: "buffer_load_dwordx2", 3: "buffer_load_dwordx3", 4: "buffer_load_dwordx4"}[VPL]
                e(op, vreg(P.v_voff), s("rs_in", 4), s("so"), offen=True, lds=True)
                if n > 1:
                    e("s_sub_u32", s("so"), s("so"), s("pix"))
                    e("s_sub_u32", "m0", "m0", SB)
            e("s_mov_b32", "m0", s("safe_m0"))
            e("s_set_gpr_idx_on", s("ring_t"), "gpr_idx(SRC1)", comment="(the dispatcher sets M0 from the next op)")
            e("s_mov_b32", "m0", s("safe_m0"))
            self.tail()
            # ---- CP n: the oldest n ring slots -> window slots 0 .. n - 1 (the WAIT op in front of it has let them arrive)
            for n in range(1, W + 1):
                self.label("cp_n%d" % n)
                e("s_mov_b32", "m0", s("safe_m0"))
                e("s_add_u32", s("ring_t"), s("ring_tail"), n * SB)
                e("s_cmp_gt_u32", s("ring_t"), RB)
                e("s_cselect_b32", s("ring_tail"), 0, s("ring_tail"))
                e("v_add_u32", vreg(P.v_lds), s("ring_tail"), vreg(P.v_lane16))
                e("s_add_u32", s("ring_tail"), s("ring_tail"), n * SB)
                if n != W:
                    e("s_branch", "cp_blk%d" % n)
            rd = {1: "ds_read_b32", 2: "ds_read_b64", 3: "ds_read_b96", 4: "ds_read_b128"}[VPL]
            for n in range(W, 0, -1):
                self.label("cp_blk%d" % n)
                e(rd, vreg(P.PHYS_WIN + P.RS * (n - 1), VPL), vreg(P.v_lds), offset=(n - 1) * SB)
            e("s_waitcnt", "lgkmcnt(0)")
            self.tail()
        # ---- REFILL -----------------------------------------------------------------------------------------------------------
        self.label("refill")
        e("s_waitcnt", "vmcnt(0)")
        e("v_mov_b32", vreg(P.v_progA), vreg(P.v_progB))
        if P.pipe:
            e("v_mov_b32", vreg(P.v_auxA), vreg(P.v_auxB))
        e("buffer_load_dword", vreg(P.v_progB), vreg(P.v_lane4), s("rs_prog", 4), s("progoff"), offen=True)
        if P.pipe:
            e("buffer_load_dword", vreg(P.v_auxB), vreg(P.v_lane4), s("rs_prog", 4), s("progoff"), offen=True, offset=256)
        e("s_add_u32", s("progoff"), s("progoff"), 512 if P.pipe else 256)
        e("s_mov_b32", s("i"), 0)
        self.tail()

    def end_handler(self):
        P, e = self.P, self.e
        K, G, VPL, W = P.K, P.G, P.VPL, P.W
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        # ---- END: pf:161 -------------------------------------------------------------------------------------------------------
        self.label("h_end")
        self.label("end")
        e("s_set_gpr_idx_off")
        if P.persist:     # the next patch of this wave: its program travels under the divisions and the stores
            e("s_add_u32", s("cgn"), s("cg"), s("nwc"))
            e("s_cmp_ge_u32", s("cgn"), s("ngroups"))
            e("s_cbranch_scc1", "e_nofetch")
            self.next_program_fetch()
            self.label("e_nofetch")
        e("s_waitcnt", "lgkmcnt(0)", comment="the region sizes, requested in the prologue")
        T = P.PHYS_WIN                                                  # the window is dead: temporaries
        assert P.W * P.RS >= 1 + 5 * VPL, "the division's temporaries live in the window registers"
        # pf:161 sum / n, correctly rounded like NumPy's float32 division: the compiler's own sequence (v_div_scale x 2,
        # v_rcp, four fma, v_div_fmas, v_div_fixup), the VPL components of an anchor interleaved instruction by
        # instruction so that no instruction waits for the one in front of it (each component's scale flags in an SGPR
        # pair of its own, moved to VCC in front of its v_div_fmas).  (A reciprocal-based division - RN(1/n) once per
        # anchor, then mul + 2 fma + fixup per component - is exact for normal quotients but rounds exact ties among
        # subnormal quotients the wrong way, which the special-value test caught; not adopted.)
        den = T
        tmp = lambda c, i: T + 1 + 5 * c + i                           # a, r, t, b, q of component c
        pairs = [S["t0"], S["t2"], S["t4"], S["dump"]]                  # SGPR pairs for the components' scale flags
        for k in range(K if not (P.debug & 1) else 0):               # debug 1: raw sums instead of quotients
            for j in range(G):
                if P.skip:
                    e("s_and_b32", s("t0"), sreg(S["cnt"] + k * G + j), 0xfffff, comment="all four arms 0: not this kernel's pixel")
                    e("s_cbranch_scc0", "nodiv_%d_%d" % (k, j))
                e("s_lshr_b32", s("t0"), sreg(S["cnt"] + k * G + j), 20)
                e("v_cvt_f32_u32", vreg(den), s("t0"))
                C = range(VPL)
                num = lambda c: P.acc(k, j, c)
                for c in C:
                    e("pseudo_div", vreg(num(c)), vreg(den), 0)       # simulator: the quotient; the real code follows
                A, Rr, Tt, B, Q = 0, 1, 2, 3, 4
                for c in C:
                    e("v_div_scale_f32", vreg(tmp(c, A)), sreg(S["pfa"], 2), vreg(den), vreg(den), vreg(num(c)), sim_skip=True)
                for c in C:
                    e("v_rcp_f32", vreg(tmp(c, Rr)), vreg(tmp(c, A)), sim_skip=True)
                for c in C:   # (also keeps every v_rcp result a slot away from its first reader: gfx940+ trans hazard)
                    e("v_div_scale_f32", vreg(tmp(c, B)), sreg(pairs[c], 2), vreg(num(c)), vreg(den), vreg(num(c)), sim_skip=True)
                for c in C:
                    e("v_fma_f32", vreg(tmp(c, Tt)), "-" + vreg(tmp(c, A)), vreg(tmp(c, Rr)), 1.0, sim_skip=True)
                for c in C:
                    e("v_fmac_f32", vreg(tmp(c, Rr)), vreg(tmp(c, Tt)), vreg(tmp(c, Rr)), sim_skip=True)
                for c in C:
                    e("v_mul_f32", vreg(tmp(c, Q)), vreg(tmp(c, B)), vreg(tmp(c, Rr)), sim_skip=True)
                for c in C:
                    e("v_fma_f32", vreg(tmp(c, Tt)), "-" + vreg(tmp(c, A)), vreg(tmp(c, Q)), vreg(tmp(c, B)), sim_skip=True)
                for c in C:
                    e("v_fmac_f32", vreg(tmp(c, Q)), vreg(tmp(c, Tt)), vreg(tmp(c, Rr)), sim_skip=True)
                for c in C:
                    e("v_fma_f32", vreg(tmp(c, A)), "-" + vreg(tmp(c, A)), vreg(tmp(c, Q)), vreg(tmp(c, B)), sim_skip=True)
                for c in C:
                    e("s_mov_b64", "vcc", sreg(pairs[c], 2), sim_skip=True)
                    e("s_nop", 1, sim_skip=True)
                    e("v_div_fmas_f32", vreg(tmp(c, A)), vreg(tmp(c, A)), vreg(tmp(c, Rr)), vreg(tmp(c, Q)), sim_skip=True)
                for c in C:
                    e("v_div_fixup_f32", vreg(num(c)), vreg(tmp(c, A)), vreg(den), vreg(num(c)), sim_skip=True)
                if P.skip:
                    self.label("nodiv_%d_%d" % (k, j))
        # stores: per anchor row a descriptor that ends with the row / the image (columns past the edge are dropped)
        e("s_sub_u32", s("t5"), s("W"), s("x0"))
        if P.tile:
            e("s_max_i32", s("t5"), s("t5"), 0, comment="a patch of the padded tile that lies right of the image stores nothing")
        e("s_min_i32", s("t5"), s("t5"), G)
        e("s_mul_i32", s("t5"), s("t5"), s("pix"), comment="bytes of the patch's columns inside the image")
        e("s_mov_b32", sreg(S["rs_out"] + 3), 0x00020000)
        for k in range(K):
            e("s_add_u32", s("t0"), s("y0"), k)
            e("s_cmp_lt_i32", s("t0"), s("H"))
            e("s_cselect_b32", sreg(S["rs_out"] + 2), s("t5"), 0)
            if P.debug & 2:                                             # debug 2 (timing only): nothing is stored
                e("s_mov_b32", sreg(S["rs_out"] + 2), 0)
            if P.wta:                                                   # a right volume nothing reads is not written
                e("s_cmp_eq_u32", s("store1"), 0)
                e("s_cselect_b32", sreg(S["rs_out"] + 2), 0, sreg(S["rs_out"] + 2))
            e("s_mul_i32", s("t0"), s("t0"), s("W"))
            e("s_add_u32", s("t0"), s("t0"), s("x0"))
            e("s_mul_hi_u32", s("t2"), s("t0"), s("pix"))
            e("s_mul_i32", s("t1"), s("t0"), s("pix"))
            e("s_add_u32", s("rs_out"), s("outp"), s("t1"))
            e("s_addc_u32", sreg(S["rs_out"] + 1), sreg(S["outp"] + 1), s("t2"))
            e("s_and_b32", sreg(S["rs_out"] + 1), sreg(S["rs_out"] + 1), 0xffff)
            if P.refresh:      # the same window over the input buffer
                e("s_add_u32", s("rs_ref"), s("inp"), s("t1"))
                e("s_addc_u32", sreg(S["rs_ref"] + 1), sreg(S["inp"] + 1), s("t2"))
                e("s_and_b32", sreg(S["rs_ref"] + 1), sreg(S["rs_ref"] + 1), 0xffff)
                e("s_mov_b32", sreg(S["rs_ref"] + 2), sreg(S["rs_out"] + 2))
                e("s_mov_b32", sreg(S["rs_ref"] + 3), 0x00020000)
            e("s_mov_b32", s("so"), 0)
            for j in range(G):
                pol = {0: dict(nt=True), 4: {}, 8: dict(sc1=True), 12: dict(sc0=True, sc1=True), 16: dict(sc1=True, nt=True),
                       20: dict(sc0=True, sc1=True, nt=True)}[P.debug & 28]      # debug 4..20: other cache policies
                if P.skip:
                    e("s_and_b32", s("t0"), sreg(S["cnt"] + k * G + j), 0xfffff)
                    e("s_cbranch_scc0", "nostore_%d_%d" % (k, j))
                self.vstore(P.acc(k, j), P.v_voff, S["rs_out"], s("so"), **pol)
                e("s_nop", 0, comment="gfx950 store-data hazard (common.h)")
                if P.refresh:       # a unit region: its value v1 also replaces v0 in the input buffer
                    e("s_and_b32", s("t0"), sreg(S["cnt"] + k * G + j), 0xfffff)
                    e("s_cbranch_scc1", "norefresh_%d_%d" % (k, j))
                    self.vstore(P.acc(k, j), P.v_voff, S["rs_ref"], s("so"), nt=True)   # (nt: -1.3 % of the stage, B.5)
                    e("s_nop", 0, comment="gfx950 store-data hazard (common.h)")
                    self.label("norefresh_%d_%d" % (k, j))
                if P.skip:
                    self.label("nostore_%d_%d" % (k, j))
                if j + 1 < G:
                    e("s_add_u32", s("so"), s("so"), s("pix"))
        if P.wta:
            self.wta_tail()
        if P.persist:
            e("s_cmp_ge_u32", s("cgn"), s("ngroups"))
            e("s_cbranch_scc1", "p_nextz")
            e("s_branch", "p_patch")
        self.label("done")
        e("s_endpgm")

    # a7 fused into the last iteration (pf:245-254): the first strict minimum over d of every result pixel - the two
    # wave reductions of wta_hwd_kernel (cbca_hwd.hip) on the quotients the wave still holds: the minimum, then the
    # lowest index among the lanes that hold it; NaN never wins, -1 when nothing does.  One chunk of disparities.
    def wta_tail(self):
        P, e = self.P, self.e
        K, G, VPL = P.K, P.G, P.VPL
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        T = P.PHYS_WIN
        best, bd, d0v, tmp, lane0off = T, T + 1, T + 2, T + 3, T + 4
        dpp = [("quad_perm:[1,0,3,2]", None), ("quad_perm:[2,3,0,1]", None), ("row_half_mirror", None),
               ("row_mirror", None), ("row_bcast:15", "0xa"), ("row_bcast:31", "0xc")]
        e("pseudo_wta", P.nacc, comment="simulator: evaluates the whole tail below")
        k0 = len(self.ins)
        sk = dict(sim_skip=True)
        # d0 of this lane = lane * VPL; voffset of the map store: 0 in lane 0, out of range elsewhere
        e("v_lshrrev_b32", vreg(d0v), 2, vreg(P.v_lane4), **sk)
        e("v_mul_u32_u24", vreg(d0v), VPL, vreg(d0v), **sk)
        e("v_mov_b32", vreg(lane0off), KDROP, **sk)
        e("v_cmp_eq_u32", "vcc", 0, vreg(P.v_lane4), **sk)
        e("v_cndmask_b32", vreg(lane0off), vreg(lane0off), vreg(P.v_lane4), "vcc", **sk)   # lane 0: lane4 = 0
        # descriptor over the disparity map
        e("s_mov_b32", sreg(S["rs_out"] + 0), s("dispp"), **sk)
        e("s_and_b32", sreg(S["rs_out"] + 1), sreg(S["dispp"] + 1), 0xffff, **sk)
        e("s_mul_i32", s("t5"), s("H"), s("W"), **sk)
        e("s_lshl_b32", s("t5"), s("t5"), 2, **sk)
        e("s_mov_b32", sreg(S["rs_out"] + 3), 0x00020000, **sk)
        for k in range(K):
            for j in range(G):
                e("v_mov_b32", vreg(best), 0x7f800000, **sk)
                e("v_mov_b32", vreg(bd), -1, **sk)
                for c in range(VPL):
                    # if (d0 + c < D && v < best) { best = v; bd = d0 + c; }
                    e("v_add_u32", vreg(tmp), c, vreg(d0v), **sk)
                    e("v_cmp_gt_u32", "vcc", s("D"), vreg(tmp), **sk)
                    e("v_cmp_lt_f32_e64", sreg(S["t2"], 2), vreg(P.acc(k, j, c)), vreg(best), **sk)
                    e("s_and_b64", "vcc", "vcc", sreg(S["t2"], 2), **sk)
                    e("s_nop", 1, **sk)
                    e("v_cndmask_b32", vreg(best), vreg(best), vreg(P.acc(k, j, c)), "vcc", **sk)
                    e("v_cndmask_b32", vreg(bd), vreg(bd), vreg(tmp), "vcc", **sk)
                # wave minimum of best into lane 63 (lanes without a DPP source keep their own value)
                e("v_mov_b32", vreg(tmp), vreg(best), **sk)
                for ctrl, rowmask in dpp:
                    m = dict(row_mask=rowmask) if rowmask else {}
                    e("s_nop", 1, **sk)
                    e("v_min_f32_dpp", vreg(tmp), vreg(tmp), vreg(tmp), ctrl, dpp=True, **m, **sk)
                e("s_nop", 1, **sk)
                e("v_readlane_b32", s("t0"), vreg(tmp), 63, **sk)
                # candidates: lanes holding the minimum offer their index, the others INT_MAX
                e("v_cmp_eq_f32_e64", sreg(S["t2"], 2), s("t0"), vreg(best), **sk)
                e("v_cmp_le_i32", "vcc", 0, vreg(bd), **sk)
                e("s_and_b64", "vcc", "vcc", sreg(S["t2"], 2), **sk)
                e("v_mov_b32", vreg(tmp), 0x7fffffff, **sk)
                e("s_nop", 1, **sk)
                e("v_cndmask_b32", vreg(tmp), vreg(tmp), vreg(bd), "vcc", **sk)
                for ctrl, rowmask in dpp:
                    m = dict(row_mask=rowmask) if rowmask else {}
                    e("s_nop", 1, **sk)
                    e("v_min_i32_dpp", vreg(tmp), vreg(tmp), vreg(tmp), ctrl, dpp=True, **m, **sk)
                e("s_nop", 1, **sk)
                e("v_readlane_b32", s("t0"), vreg(tmp), 63, **sk)
                # disp[(y0 + k) * W + x0 + j] = idx == INT_MAX ? -1 : idx, by lane 0, for pixels inside the image
                e("s_cmp_eq_u32", s("t0"), 0x7fffffff, **sk)
                e("s_cselect_b32", s("t0"), -1, s("t0"), **sk)
                e("v_cvt_f32_i32", vreg(tmp), s("t0"), **sk)
                e("s_add_u32", s("t1"), s("y0"), k, **sk)
                e("s_add_u32", s("t3"), s("x0"), j, **sk)
                e("s_cmp_lt_i32", s("t1"), s("H"), **sk)
                e("s_cselect_b32", s("t4"), s("t5"), 0, **sk)
                e("s_cmp_lt_i32", s("t3"), s("W"), **sk)
                e("s_cselect_b32", sreg(S["rs_out"] + 2), s("t4"), 0, **sk)
                e("s_mul_i32", s("t1"), s("t1"), s("W"), **sk)
                e("s_add_u32", s("t1"), s("t1"), s("t3"), **sk)
                e("s_lshl_b32", s("t1"), s("t1"), 2, **sk)
                e("buffer_store_dword", vreg(tmp), vreg(lane0off), sreg(S["rs_out"], 4), s("t1"), offen=True, **sk)
        self.ins[k0 - 1].args.append(len(self.ins) - k0)

    def prefetch(self):
        """L2 warm-up through the SCALAR cache path.  The vector memory pipe of a CU keeps ~32 KiB of requests in
        flight, and a request that has to go to HBM holds its place four times as long as one that hits L2: the
        first touch of every voxel is what most of that capacity is spent on.  Scalar loads travel another way
        (scalar data cache -> L2), so every wave touches - one s_load_dword per 128-byte line, results discarded -
        the K x G pixels PF columns to the right of its own: the waves that will need them first are dispatched a few
        microseconds later on this XCD and then find them in L2."""
        P, e = self.P, self.e
        s = lambda n, c=1: sreg(S[n], c) if isinstance(n, str) else sreg(n, c)
        e("s_add_u32", s("t0"), s("x0"), P.PF)
        e("s_cmp_ge_i32", s("t0"), s("W"))
        e("s_cbranch_scc1", "pf_done")
        e("s_sub_u32", s("t5"), s("W"), s("t0"))
        e("s_min_i32", s("t5"), s("t5"), P.G)
        e("s_mul_i32", s("t5"), s("t5"), s("pix"), comment="bytes of the prefetched columns of one row")
        for k in range(P.K):
            e("s_add_u32", s("t1"), s("y0"), k)
            e("s_cmp_ge_i32", s("t1"), s("H"))
            e("s_cbranch_scc1", "pf_done")
            e("s_mul_i32", s("t1"), s("t1"), s("W"))
            e("s_add_u32", s("t1"), s("t1"), s("t0"))
            e("s_mul_hi_u32", s("t3"), s("t1"), s("pix"))
            e("s_mul_i32", s("t2"), s("t1"), s("pix"))
            e("s_add_u32", s("pfa"), s("inp"), s("t2"))
            e("s_addc_u32", sreg(S["pfa"] + 1), sreg(S["inp"] + 1), s("t3"))
            e("s_mov_b32", s("pfoff"), 0)
            self.label("pf_loop%d" % k)
            e("s_load_dword", s("dump"), s("pfa", 2), s("pfoff"))
            e("s_add_u32", s("pfoff"), s("pfoff"), 128)
            e("s_cmp_lt_u32", s("pfoff"), s("t5"))
            e("s_cbranch_scc1", "pf_loop%d" % k)
        self.label("pf_done")

    # ---- layout ----------------------------------------------------------------------------------------------------------
    def offsets(self):
        """label -> byte offset from `entry`."""
        off, out = 0, {}
        for i in self.ins:
            if i.op == "label":
                out[i.args[0]] = off
            else:
                off += i.size()
        out["__end"] = off
        return out

    def layout(self):
        P = self.P
        o = self.offsets()
        base = o["code_base"]
        u = 4 if P.ring else 1          # unit of the handler offsets in an op: dwords in the experimental ring kernels
        L = dict(VPL=P.VPL, RS=P.RS, K=P.K, G=P.G, W=P.W, MAXD=P.MAXD, MAXA=P.MAXA, R=R, NWAIT=P.NWAIT, BLK=4 * P.VPL // u,
                 M0_SRC1=M0_SRC1, code_bytes=o["__end"], nvgpr=P.nvgpr)
        L["add"] = {key: (o[name] - base) // u for key, name in self.lines.items()}
        L["tile"] = P.tile
        if P.tile:
            L["load"] = [[0] + [-1] * P.W]
            L["loadl"] = [0] + [(o["loadl_n%d" % n] - base) // u for n in range(1, P.W + 1)]
            L["step"] = (o["step"] - base) // u
            L["SLOTS"], L["SB"] = P.SLOTS, P.SB
        elif P.pipe:
            L["load"] = [[0] + [-1] * P.W]
            L["loadk"] = [(o["loadk_%d" % k] - base) // u for k in range(P.W)]
        else:
            L["load"] = [[0] + [(o["load_b%d_n%d" % (b, n)] - base) // u for n in range(1, P.W + 1)] for b in range(P.NB)]
        L["pipe"], L["UW"] = P.pipe, P.UW
        L["loadnt"] = ([0] + [(o["loadnt_n%d" % n] - base) // u for n in range(1, P.G + 1)]) if (P.ntload and not P.pipe) else None
        L["NB"] = P.NB
        L["wait"] = [(o["wait_%d" % k] - base) // u for k in range(P.NWAIT)]
        L["ring"] = P.ring
        if P.ring:
            L["pf"] = [0] + [(o["pf_n%d" % n] - base) // u for n in range(1, P.W + 1)]
            L["cp"] = [0] + [(o["cp_n%d" % n] - base) // u for n in range(1, P.W + 1)]
        L["refill"] = (o["refill"] - base) // u
        L["end"] = (o["end"] - base) // u
        assert max(L["wait"] + [L["end"], L["refill"]]) < (65536 if P.ring else 32768), "op offsets are 16 bits"
        return L

    def render(self):
        P = self.P
        name = P.name()
        out = ['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".text", ".globl %s" % name, ".p2align 8",
               ".type %s,@function" % name, "%s:" % name]
        for i in self.ins:
            if i.op == "label":
                out.append(".L%s_%s:" % (name, i.args[0]))
            elif i.op.startswith("pseudo_"):
                continue
            else:
                line = i.render()
                # local labels: branch targets and the code_base difference
                line = re.sub(r"\b(code_base|after_getpc|done|pf_done|pf_loop\d+|load_b\d+_blk\d+|loadblk_\d+|loadnt_blk\d+|load_done|loadf_blk\d+|h_end|h_load1|h_loadf1|first_is_end|nodiv_\d+_\d+|nostore_\d+_\d+|norefresh_\d+_\d+|pf_blk\d+|cp_blk\d+|loadl_blk\d+|step_loop|step_done|p_div|p_divd|p_z|p_patch|p_nextz|e_nofetch)\b", lambda m: ".L%s_%s" % (name, m.group(1)), line)
                out.append(line)
        kargs = 0x80 if (P.wta or P.persist) else 0x60
        lds_bytes = P.LDS_BYTES if P.tile else P.ring * 256 * P.VPL
        wg = 64 * P.tile if P.tile else 64
        out += [".Lfunc_end_%s:" % name, ".size %s, .Lfunc_end_%s-%s" % (name, name, name), "",
                ".rodata", ".p2align 6", ".amdhsa_kernel %s" % name,
                "  .amdhsa_group_segment_fixed_size %d" % lds_bytes, "  .amdhsa_private_segment_fixed_size 0",
                "  .amdhsa_kernarg_size %d" % kargs, "  .amdhsa_user_sgpr_count 2",
                "  .amdhsa_user_sgpr_kernarg_segment_ptr 1", "  .amdhsa_system_sgpr_workgroup_id_x 1",
                "  .amdhsa_system_sgpr_workgroup_id_y 1", "  .amdhsa_system_sgpr_workgroup_id_z 1",
                "  .amdhsa_system_vgpr_workitem_id 0", "  .amdhsa_next_free_vgpr %d" % P.nvgpr_alloc,
                "  .amdhsa_next_free_sgpr %d" % NSGPR, "  .amdhsa_accum_offset %d" % ((P.nvgpr_alloc + 3) & ~3),
                "  .amdhsa_reserve_vcc 1", "  .amdhsa_float_round_mode_32 0", "  .amdhsa_float_round_mode_16_64 0",
                "  .amdhsa_float_denorm_mode_32 3", "  .amdhsa_float_denorm_mode_16_64 3", "  .amdhsa_dx10_clamp 1",
                "  .amdhsa_ieee_mode 1", ".end_amdhsa_kernel", "",
                ".amdgpu_metadata", "---", "amdhsa.version:", "  - 1", "  - 2", "amdhsa.kernels:",
                "  - .name: %s" % name, "    .symbol: %s.kd" % name, "    .kernarg_segment_size: %d" % kargs,
                "    .kernarg_segment_align: 8", "    .group_segment_fixed_size: %d" % lds_bytes, "    .private_segment_fixed_size: 0",
                "    .wavefront_size: 64", "    .sgpr_count: %d" % (NSGPR + 6), "    .vgpr_count: %d" % P.nvgpr_alloc,
                "    .agpr_count: 0", "    .max_flat_workgroup_size: %d" % wg, "    .args:"]
        off = 0
        while off < kargs:
            out += ["      - .offset: %d" % off, "        .size: 8", "        .value_kind: by_value"]
            off += 8
        out += ["amdhsa.target: amdgcn-amd-amdhsa--gfx950", "...", ".end_amdgpu_metadata", ""]
        return "\n".join(out)


def header(L, P):
    """cbca_prog_layout_v{VPL}.h: the code layout the program builder encodes ops against."""
    v = P.VPL
    pre = "CBCA_PROG_V%d_" % v
    out = ["// generated by csrc/asm/cbca_prog_gen.py - do not edit", "#pragma once", "#include <stdint.h>"]
    for k in ("VPL", "RS", "K", "G", "W", "NB", "MAXD", "MAXA", "NWAIT", "BLK", "M0_SRC1"):
        out.append("#define %s%s %d" % (pre, k, L[k]))
    out.append("#define %sREFILL %d" % (pre, L["refill"]))
    out.append("#define %sEND %d" % (pre, L["end"]))
    # add[j][aset][dir] = offset of the line's first block (-1: no line for that set, see decompose())
    rows = []
    for j in range(P.G):
        cells = []
        for a in range(1 << P.K):
            cells.append("{%d, %d}" % ((L["add"][(j, a, "d")], L["add"][(j, a, "a")]) if (j, a, "d") in L["add"] else (-1, -1)))
        rows.append("{" + ", ".join(cells) + "}")
    out.append("#define %sADD_INIT {%s}" % (pre, ", ".join(rows)))
    out.append("#define %sLOAD_INIT {%s}" % (pre, ", ".join("{" + ", ".join(str(x) for x in row) + "}" for row in L["load"])))
    out.append("#define %sLOAD0_INIT {%s}" % (pre, ", ".join(str(x) for x in L["load"][0])))
    out.append("#define %sWAIT_INIT {%s}" % (pre, ", ".join(str(x) for x in L["wait"])))
    # a mccnn::prog::Layout (csrc/cbca_prog_build.h) for this kernel
    out.append("#define %sLAYOUT {%sVPL, %sRS, %sK, %sG, %sW, %sMAXD, %sMAXA, %sBLK, %sM0_SRC1, %sREFILL, %sEND, "
               "%sADD_INIT, %sLOAD0_INIT}" % ((pre,) * 14))
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vpl", type=int, default=4)
    ap.add_argument("--w", type=int, default=12)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--order", type=int, default=0)
    ap.add_argument("--nb", type=int, default=1)
    ap.add_argument("--pf", type=int, default=0)
    ap.add_argument("--wta", action="store_true")
    ap.add_argument("--persist", action="store_true", help="experimental (with --ring): persistent waves")
    ap.add_argument("--ring", type=int, default=0, help="experimental: window rows through an LDS ring of this many slots")
    ap.add_argument("--skip", action="store_true", help="the kernel of the skip programs (unit regions neither divided nor stored)")
    ap.add_argument("--refresh", action="store_true",
                    help="the full kernel that also writes unit-region pixels back into its input buffer (first iteration)")
    ap.add_argument("--minvgpr", type=int, default=0, help="experiments: allocate at least this many VGPRs (occupancy)")
    ap.add_argument("--ntload", action="store_true", help="non-temporal loads for region rows of unit-region pixels")
    ap.add_argument("--early", action="store_true", help="experimental: the program's first op dispatched from a scalar load")
    ap.add_argument("--pipe", type=int, default=0, help="the window is a program-managed ring of --w slots; the widest unit")
    ap.add_argument("--tile", type=int, default=0, help="waves per workgroup that share their region rows through LDS (0: none)")
    ap.add_argument("-o", default=None)
    ap.add_argument("--header", default=None)
    ap.add_argument("--experimental", action="store_true",
                    help="required for every variant that was measured and NOT adopted (--nb 2, --pf, --ring, --persist, "
                         "--pipe, --ntload, --early, --tile, --minvgpr): the library's build (Makefile) never passes it, so a "
                         "shipped code object cannot pick one up by accident")
    a = ap.parse_args()
    experimental = [n for n, on in (("--nb", a.nb != 1), ("--pf", a.pf), ("--ring", a.ring), ("--persist", a.persist),
                                    ("--pipe", a.pipe), ("--ntload", a.ntload), ("--early", a.early), ("--tile", a.tile),
                                    ("--minvgpr", a.minvgpr), ("--order", a.order)) if on]
    if experimental and not a.experimental:
        ap.error("%s: measured and not adopted - pass --experimental to generate it anyway" % ", ".join(experimental))
    P = Params(vpl=a.vpl, K=a.k, W=a.w, NB=a.nb, PF=a.pf, wta=a.wta, order=a.order, minvgpr=a.minvgpr, skip=a.skip, ring=a.ring, persist=a.persist, pipe=a.pipe, ntload=a.ntload, early=a.early, tile=a.tile, refresh=a.refresh)
    g = Gen(P).build()
    if a.o:
        open(a.o, "w").write(g.render())
    if a.header:
        open(a.header, "w").write(header(g.layout(), P))
    if not a.o and not a.header:
        sys.stdout.write(g.render())


if __name__ == "__main__":
    main()
