"""Plain-Python statement of the per-patch aggregation programs (csrc/cbca_prog.hip builds them on the GPU, the
assembly kernel of csrc/asm/cbca_prog_gen.py interprets them) + an op-level interpreter on numpy.

Test infrastructure: `build_program` is the specification the device builder is compared with word for word,
`run_program` executes a program's LOAD / ADD ops on a pixel-major volume so that a program can be checked against the
CPU oracle of pf:149-163 without any GPU.
"""
import numpy as np

R = 13


def decompose(aset, K):
    """= cbca_prog_gen.decompose: a set of anchor rows as aligned power-of-two groups, largest first."""
    out, s = [], K
    while s >= 1 and aset:
        for st in range(0, K, s):
            m = ((1 << s) - 1) << st
            if aset & m == m:
                out.append(m)
                aset &= ~m
        s //= 2
    return out


def arms_of(word):
    w = int(word)
    return w & 31, (w >> 5) & 31, (w >> 10) & 31, (w >> 15) & 31     # up, down, left, right


def band_rows_of(H, K):
    per = -(-H // 8)
    return -(-per // K) * K


def prog_stride_dwords(L):
    """= mccnn::prog::stride_dwords (csrc/cbca_prog_build.h): upper bound of a patch's program in dwords."""
    K, G, W = L["K"], L["G"], L.get("UW", L["W"])
    rows = 2 * K + 2 * R - 1
    groups = K // 2 if K >= 2 else 1
    pieces = -(-(R + 1) // W)
    if L.get("pipe"):      # every unit: LOAD + WAIT + its arm runs; every op has a second word (64 ops + 64 words per chunk)
        n = rows * pieces * (4 * G + 2 * G * groups) + 2
        n += n // 63 + 1
        return 2 * (-(-n // 64) * 64)
    n = rows * pieces * (2 * G + 2 * G * groups) + 2
    n += n // 63 + 1
    return -(-n // 64) * 64


def unit_region(word):
    """True for a pixel whose support region is the pixel itself (all four arms 0): (0 + x) / 1 = x, so from the third
    consecutive iteration of a ping-pong pair on both buffers already hold its value (pf:149-163 with aver_num = 1)."""
    return (int(word) & 0xfffff) == 0


def step_sets(K, G, ok, up, dn, nd, t):
    """Anchor rows of every column that take part in sweep step t (pf:155: self, up 1.., then down 1..)."""
    aset = np.zeros(G, int)
    for k in range(K):
        for j in range(G):
            if not ok[k, j]:
                continue
            if t < nd:
                on = K - 1 - k <= t <= K - 1 - k + up[k, j]
            else:
                on = k <= t - nd <= k + dn[k, j] - 1
            if on:
                aset[j] |= 1 << k
    return aset


def plan_row(sup0, W, yq, row0, x0, aset, L, units):
    """The window units of ONE region row (yq) for the patch columns x0 .. x0 + G - 1 whose anchor-row sets are `aset`,
    appended to `units` in execution order (see plan_units)."""
    G, WW = L["G"], L.get("UW", L["W"])
    MAXD, MAXA = L["MAXD"], L["MAXA"]
    H = sup0.shape[0]
    cols = [j for j in range(G) if aset[j]]
    if not cols:
        return
    assert 0 <= yq < H
    lr = {}
    for j in cols:
        _, _, l, r = arms_of(sup0[yq, x0 + j])
        lr[j] = (min(l, R), min(r, R))
    c = {j: j + R for j in cols}                                  # virtual slot of the column's own pixel

    def unit(lo, hi, runs):
        assert 1 <= hi - lo + 1 <= WW
        p = (yq - row0) * W + (x0 - R + hi)
        assert 0 <= p < 65536 or L.get("pipe") or L.get("tile"), "pixel index exceeds the op's 16-bit field"
        # single-use row: every pixel of the window is a unit-region pixel (and lies in the patch's own columns)
        single = R <= lo and hi < R + G and all(unit_region(sup0[yq, x0 - R + v]) for v in range(lo, hi + 1))
        units.append((lo, hi, p, runs, single))

    desc = lambda j, first, n: ("d", j, int(aset[j]), first, n)
    asc = lambda j, first, n: ("a", j, int(aset[j]), first, n)
    lo = min(c[j] - lr[j][0] for j in cols)
    hi = max(c[j] + lr[j][1] for j in cols)
    if hi - lo + 1 <= WW and max(lr[j][0] + 1 for j in cols) <= MAXD and max(lr[j][1] for j in cols) <= MAXA:
        unit(lo, hi, [desc(j, c[j], lr[j][0] + 1) for j in cols] + [asc(j, c[j] + 1, lr[j][1]) for j in cols if lr[j][1]])
        return
    # wide row: windows over groups of columns, descending arms first
    i = 0
    while i < len(cols):
        j = cols[i]
        glo, ghi = c[j] - lr[j][0], c[j]
        if ghi - glo + 1 > min(WW, MAXD):                          # one arm longer than the window: in pieces
            first, left = c[j], lr[j][0] + 1
            while left:
                n = min(left, WW, MAXD)
                unit(first - n + 1, first, [desc(j, first, n)])
                first, left = first - n, left - n
            i += 1
            continue
        grp = [j]
        while i + len(grp) < len(cols):
            j2 = cols[i + len(grp)]
            nlo, nhi = min(glo, c[j2] - lr[j2][0]), max(ghi, c[j2])
            if nhi - nlo + 1 > WW or lr[j2][0] + 1 > MAXD:
                break
            glo, ghi = nlo, nhi
            grp.append(j2)
        unit(glo, ghi, [desc(jj, c[jj], lr[jj][0] + 1) for jj in grp])
        i += len(grp)
    acols = [j for j in cols if lr[j][1]]
    i = 0
    while i < len(acols):
        j = acols[i]
        glo, ghi = c[j] + 1, c[j] + lr[j][1]
        if ghi - glo + 1 > min(WW, MAXA):
            first, left = c[j] + 1, lr[j][1]
            while left:
                n = min(left, WW, MAXA)
                unit(first, first + n - 1, [asc(j, first, n)])
                first, left = first + n, left - n
            i += 1
            continue
        grp = [j]
        while i + len(grp) < len(acols):
            j2 = acols[i + len(grp)]
            nlo, nhi = min(glo, c[j2] + 1), max(ghi, c[j2] + lr[j2][1])
            if nhi - nlo + 1 > WW or lr[j2][1] > MAXA:
                break
            glo, ghi = nlo, nhi
            grp.append(j2)
        unit(glo, ghi, [asc(jj, c[jj] + 1, lr[jj][1]) for jj in grp])
        i += len(grp)


def plan_units(sup0, H, W, y0, x0, L, skip_unit=False):
    """The patch's work as a list of window units in execution order: (lo, hi, p_last, runs) - load virtual slots
    lo .. hi of one region row (p_last = pixel index of slot hi relative to the patch's first region row), then the arm
    runs (dir, column j, anchor set, first slot, n) that read them.  skip_unit: anchors whose region is the pixel itself
    take no part (the kernel's skip variant neither divides nor stores them)."""
    K, G, WW = L["K"], L["G"], L.get("UW", L["W"])
    MAXD, MAXA = L["MAXD"], L["MAXA"]
    units = []
    up, dn, ok = np.zeros((K, G), int), np.zeros((K, G), int), np.zeros((K, G), bool)
    lowest = highest = y0
    for k in range(K):
        y = y0 + k
        for j in range(G):
            x = x0 + j
            if x < W and y < H and not (skip_unit and unit_region(sup0[y, x])):
                u, d, _, _ = arms_of(sup0[y, x])
                u, d = min(u, y), min(d, H - 1 - y)
                up[k, j], dn[k, j], ok[k, j] = u, d, True
                lowest = min(lowest, y - u)
                highest = max(highest, y + d if d > 0 else y0)
    nd = y0 + K - 1 - lowest + 1
    na = highest - y0
    row0 = max(y0 - R, 0)
    for t in range(nd + na):
        yq = y0 + K - 1 - t if t < nd else y0 + 1 + (t - nd)
        aset = step_sets(K, G, ok, up, dn, nd, t)
        plan_row(sup0, W, yq, row0, x0, aset, L, units)
    return units


def build_program(sup0, H, W, y0, x0, L, skip_unit=False):
    """sup0: [H, W] uint32 support words (plane 0).  Returns the uint32 ops of the patch at rows y0.., columns x0...
    One window (NB = 1): LOAD (which waits), the unit's arms, next unit.  Two windows: the next unit's LOAD is issued
    before the current unit's arms, WAIT k (k = slots of that newest LOAD) lets exactly the current window arrive."""
    WW, RS, NB = L["W"], L["RS"], L["NB"]
    MAXD, MAXA, BLK = L["MAXD"], L["MAXA"], L["BLK"]
    M0 = L["M0_SRC1"]
    ops, aux = [], []

    def emit(op, second=0):
        if len(ops) % 64 == 63:
            ops.append(L["refill"] | (M0 << 16))
            aux.append(0)
        ops.append(op & 0xffffffff)
        aux.append(second & 0xffffffff)

    units = plan_units(sup0, H, W, y0, x0, L, skip_unit)

    def load(i):
        lo, hi, p, _, single = units[i]
        if single and L.get("loadnt"):
            emit(L["loadnt"][hi - lo + 1] | (p << 16))
        else:
            emit(L["load"][i % NB][hi - lo + 1] | (p << 16))

    def arms(i, wb=None):
        lo, hi, p, runs, _ = units[i]
        if wb is None:
            wb = (i % NB) * WW                                         # first physical slot of this unit's window
        for d, j, aset_all, first, n in runs:
            sf = first - lo
            for aset in decompose(aset_all, L["K"]):                   # aligned groups of anchor rows the kernel has lines for
                nk = bin(aset).count("1")
                if d == "d":
                    assert 1 <= n <= MAXD and 0 <= sf - n + 1 and sf < L.get("UW", WW)
                    emit((L["add"][(j, aset, "d")] + (MAXD - n) * BLK * nk) | ((M0 | (RS * (wb + sf - n + 1))) << 16))
                else:
                    assert 1 <= n <= MAXA and 0 <= sf and sf + n - 1 < L.get("UW", WW)
                    emit((L["add"][(j, aset, "a")] + (MAXA - n) * BLK * nk) | ((M0 | (RS * (wb + sf + n - 1))) << 16))

    if L.get("pipe"):
        # The window is a ring of WW slots managed HERE (cbca_prog_gen.py, pipe): a unit takes the next free run of
        # consecutive slots behind the head (a unit that does not fit before the end of the ring starts at slot 0), the
        # loads of as many upcoming units as fit are issued ahead, and WAIT k - k = slots requested after the unit that
        # is due - lets exactly that unit arrive before its arms read it.  A slot is requested again only after the
        # arms that read it (program order), so no load can overtake a reader.
        pix = L["pix"]
        head, issued, live = 0, 0, []

        def try_issue():
            nonlocal head, issued
            while issued < len(units):
                lo, hi, p, _, _ = units[issued]
                n = hi - lo + 1
                start = 0 if head + n > WW else head
                if any(start < a + m and a < start + n for a, m in live):
                    break
                assert p * pix < 2 ** 31
                emit(L["loadk"][start + n - 1] | ((n - 1) << 16), p * pix)
                live.append((start, n))
                head = start + n
                issued += 1
        try_issue()
        for i in range(len(units)):
            assert live, "unit %d was never requested" % i
            emit(L["wait"][sum(m for _, m in live[1:])] | (M0 << 16))
            arms(i, live[0][0])
            live.pop(0)
            try_issue()
        emit(L["end"] | (M0 << 16))
        out = np.zeros(-(-len(ops) // 64) * 128, np.uint32)
        o, a = np.array(ops, np.uint32), np.array(aux, np.uint32)
        for c in range(0, len(ops), 64):
            m = min(64, len(ops) - c)
            out[2 * c:2 * c + m] = o[c:c + m]
            out[2 * c + 64:2 * c + 64 + m] = a[c:c + m]
        return out
    elif L.get("ring"):
        # experimental (cbca_prog_gen.py --ring S): the units travel through an LDS ring of S slots.  PF streams a unit into
        # the next free slots (behind the head; a unit that does not fit there starts at slot 0 - the kernel keeps the
        # same head), as many units ahead as the ring holds; WAIT k lets the oldest unit arrive (k = slots requested after
        # it), CP copies it into the register window and frees its slots.
        S = L["ring"]
        head, issued, live = 0, 0, []

        def try_issue():
            nonlocal head, issued
            while issued < len(units):
                lo, hi, p, _, _ = units[issued]
                n = hi - lo + 1
                start = 0 if head + n > S else head
                if any(start < a + m and a < start + n for a, m in live):
                    break
                emit(L["pf"][n] | (p << 16))
                live.append((start, n))
                head = start + n
                issued += 1
        try_issue()
        for i in range(len(units)):
            n = units[i][1] - units[i][0] + 1
            assert live, "unit %d was never requested" % i
            emit(L["wait"][sum(m for _, m in live[1:])] | (M0 << 16))
            emit(L["cp"][n] | (M0 << 16))
            live.pop(0)
            try_issue()
            arms(i)
    elif NB == 1:
        for i in range(len(units)):
            load(i)
            arms(i)
    else:
        if units:
            load(0)
        for i in range(len(units)):
            if i + 1 < len(units):
                load(i + 1)
                emit(L["wait"][units[i + 1][1] - units[i + 1][0] + 1] | (M0 << 16))
            else:
                emit(L["wait"][0] | (M0 << 16))
            arms(i)
    emit(L["end"] | (M0 << 16))
    return np.array(ops, np.uint32)


def build_all(sup0, H, W, L, skip_unit=False):
    """[row groups of the 8 bands][column groups][stride] uint32, the layout the kernel indexes."""
    K, G = L["K"], L["G"]
    br = band_rows_of(H, K)
    bg = br // K
    ngroups = -(-W // G)
    stride = prog_stride_dwords(L)
    out = np.zeros((8 * bg, ngroups, stride), np.uint32)
    longest = 0
    for rg in range(8 * bg):
        y0 = rg * K
        if y0 >= H:
            continue
        for cg in range(ngroups):
            p = build_program(sup0, H, W, y0, cg * G, L, skip_unit)
            assert len(p) <= stride, (len(p), stride)
            longest = max(longest, len(p))
            out[rg, cg, :len(p)] = p
    return out, dict(band_rows=br, band_groups=bg, ngroups=ngroups, stride=stride, longest=longest)


def tile_shape(H, W, L):
    """Launch geometry of the tile kernels (cbca_prog_gen.py, tile): row groups of the 8 bands, tiles per row, padded
    column groups, words per program."""
    K, G, NW = L["K"], L["G"], L["tile"]
    br = band_rows_of(H, K)
    ntx = -(-W // (G * NW))
    return dict(band_rows=br, band_groups=br // K, ntx=ntx, ngroups=ntx * NW, stride=tile_stride_dwords(L))


def tile_stride_dwords(L):
    """Upper bound of a tile wave's program: a patch program's ops (prog_stride_dwords) + per sweep step the two words
    of a STEP op and a pad word."""
    K = L["K"]
    n = prog_stride_dwords(L) + 3 * (2 * K + 2 * R - 1)
    n += n // 63 + 1
    return -(-n // 64) * 64


def build_tile_programs(sup0, H, W, y0, tx, L, skip_unit=False):
    """The programs of the NW = L["tile"] waves of one tile (rows y0 .., columns tx * NW * G ..): every sweep step of the
    TILE (the schedule of patch_setup / plan_units over all NW x G columns) is a STEP op in every wave's program - the
    union of the tile's horizontal arms in that region row goes to LDS slots 0 .. - followed by the wave's own window
    units as LOADL (LDS slot of the window's first pixel) + the unchanged ADD ops."""
    K, G, NW, WW, RS = L["K"], L["G"], L["tile"], L["W"], L["RS"]
    MAXD, MAXA, BLK, M0 = L["MAXD"], L["MAXA"], L["BLK"], L["M0_SRC1"]
    TW = NW * G
    x0t = tx * TW
    up, dn, ok = np.zeros((K, TW), int), np.zeros((K, TW), int), np.zeros((K, TW), bool)
    lowest = highest = y0
    for k in range(K):
        y = y0 + k
        for c in range(TW):
            x = x0t + c
            if x < W and y < H and not (skip_unit and unit_region(sup0[y, x])):
                u, d, _, _ = arms_of(sup0[y, x])
                u, d = min(u, y), min(d, H - 1 - y)
                up[k, c], dn[k, c], ok[k, c] = u, d, True
                lowest = min(lowest, y - u)
                highest = max(highest, y + d if d > 0 else y0)
    nd = y0 + K - 1 - lowest + 1
    na = highest - y0
    row0 = max(y0 - R, 0)
    progs = [[] for _ in range(NW)]
    nop = L["wait"][-1] | (M0 << 16)                      # s_waitcnt vmcnt(W): waits for nothing

    def emit(w, op):
        ops = progs[w]
        if len(ops) % 64 == 63:
            ops.append(L["refill"] | (M0 << 16))
        ops.append(op & 0xffffffff)

    stats = dict(steps=0, slots=0, units=0)
    for t in range(nd + na):
        yq = y0 + K - 1 - t if t < nd else y0 + 1 + (t - nd)
        aset = step_sets(K, TW, ok, up, dn, nd, t)
        cols = [c for c in range(TW) if aset[c]]
        if not cols:
            continue
        lo_t, hi_t = 1 << 30, -1
        for c in cols:
            _, _, l, r = arms_of(sup0[yq, x0t + c])
            lo_t, hi_t = min(lo_t, c + R - min(l, R)), max(hi_t, c + R + min(r, R))
        nslots = hi_t - lo_t + 1
        assert 1 <= nslots <= L["SLOTS"]
        p = (yq - row0) * W + (x0t - R + lo_t)            # pixel index of LDS slot 0 relative to the first region row
        assert p >= 0
        stats["steps"] += 1
        stats["slots"] += nslots
        for w in range(NW):
            if len(progs[w]) % 64 >= 62:                  # a STEP's two words stay in one 64-op chunk (63 = its REFILL)
                emit(w, nop)
                if len(progs[w]) % 64 == 63:
                    emit(w, nop)
            emit(w, L["step"] | (nslots << 16))
            emit(w, p)
            units = []
            plan_row(sup0, W, yq, row0, x0t + w * G, aset[w * G:(w + 1) * G], L, units)
            for lo, hi, _, runs, _ in units:
                n = hi - lo + 1
                first = w * G + lo - lo_t                 # LDS slot of the window's first pixel
                assert 0 <= first and first + n <= nslots
                emit(w, L["loadl"][n] | (first << 16))
                stats["units"] += 1
                for d, j, aset_all, firstv, cnt in runs:
                    sf = firstv - lo
                    for a in decompose(aset_all, K):
                        nk = bin(a).count("1")
                        if d == "d":
                            assert 1 <= cnt <= MAXD and 0 <= sf - cnt + 1 and sf < WW
                            emit(w, (L["add"][(j, a, "d")] + (MAXD - cnt) * BLK * nk) | ((M0 | (RS * (sf - cnt + 1))) << 16))
                        else:
                            assert 1 <= cnt <= MAXA and 0 <= sf and sf + cnt - 1 < WW
                            emit(w, (L["add"][(j, a, "a")] + (MAXA - cnt) * BLK * nk) | ((M0 | (RS * (sf + cnt - 1))) << 16))
    for w in range(NW):
        emit(w, L["end"] | (M0 << 16))
    return [np.array(o, np.uint32) for o in progs], stats


def build_all_tiles(sup0, H, W, L, skip_unit=False):
    """[row groups of the 8 bands][padded column groups][stride] uint32: the layout the tile kernels index."""
    K, G, NW = L["K"], L["G"], L["tile"]
    m = tile_shape(H, W, L)
    out = np.zeros((8 * m["band_groups"], m["ngroups"], m["stride"]), np.uint32)
    longest, tot = 0, dict(steps=0, slots=0, units=0)
    for rg in range(8 * m["band_groups"]):
        y0 = rg * K
        for tx in range(m["ntx"]):
            if y0 >= H:
                continue
            progs, st = build_tile_programs(sup0, H, W, y0, tx, L, skip_unit)
            for k in tot:
                tot[k] += st[k]
            for w, pw in enumerate(progs):
                assert len(pw) <= m["stride"], (len(pw), m["stride"])
                longest = max(longest, len(pw))
                out[rg, tx * NW + w, :len(pw)] = pw
    return out, dict(m, longest=longest, **tot)


def decode_tables(L):
    add = {}
    for (j, aset, d), off in L["add"].items():
        nk = bin(aset).count("1")
        maxn = L["MAXD"] if d == "d" else L["MAXA"]
        for n in range(1, maxn + 1):
            add[off + (maxn - n) * L["BLK"] * nk] = (j, aset, d, n)
    load = {off: (b, n) for b, row in enumerate(L["load"]) for n, off in enumerate(row) if n}
    if L.get("loadnt"):
        load.update({off: (0, n) for n, off in enumerate(L["loadnt"]) if n})
    wait = {off: k for k, off in enumerate(L["wait"])}
    return add, load, wait


def run_program(prog, vol, H, W, y0, x0, L, sup0):
    """Executes one patch program on vol [H, W, D] (float32); returns {(y, x): quotient vector}."""
    K, G, WW, VPL = L["K"], L["G"], L["W"], L["RS"]
    add, load, wait = decode_tables(L)
    D = vol.shape[2]
    row0 = max(y0 - R, 0)
    flat = vol.reshape(H * W, D)
    win = np.full((WW * L["NB"], D), np.nan, np.float32)
    acc = np.zeros((K, G, D), np.float32)
    pc = 0
    pipe = L.get("pipe")
    loadk = {off: k for k, off in enumerate(L["loadk"])} if pipe else {}
    while True:
        # (pipe: a 64-op chunk is followed by the 64 second words of its ops)
        op = int(prog[(pc // 64) * 128 + pc % 64]) if pipe else int(prog[pc])
        second = int(prog[(pc // 64) * 128 + 64 + pc % 64]) if pipe else 0
        pc += 1
        off, par = op & 0xffff, op >> 16
        if off in loadk and off != L["end"] and off != L["refill"] and off not in wait:
            k, n = loadk[off], (par & 0xff) + 1
            assert second % L["pix"] == 0
            p = second // L["pix"]
            for e in range(n):
                win[k - e] = flat[row0 * W + p - e]
            continue
        if off == L["end"]:
            break
        if off == L["refill"]:
            assert pc % 64 == 0
            continue
        if off in wait:
            continue
        if off in load:
            b, n = load[off]
            win[b * WW:(b + 1) * WW] = np.nan
            for s in range(n):
                win[b * WW + s] = flat[row0 * W + par - (n - 1 - s)]
            continue
        j, aset, d, n = add[off]
        idx = (par & 0xff) // VPL
        assert (par & 0xff) % VPL == 0 and par >> 12 == 2
        order = [idx + (n - 1) - e for e in range(n)] if d == "d" else [idx - (n - 1) + e for e in range(n)]
        for s in order:
            assert 0 <= s < WW * L["NB"] and not np.isnan(win[s]).all(), "window slot %d not loaded" % s
            for k in range(K):
                if aset & (1 << k):
                    acc[k, j] = acc[k, j] + win[s]
    out = {}
    for k in range(K):
        for j in range(G):
            y, x = y0 + k, x0 + j
            if y < H and x < W:
                out[(y, x)] = acc[k, j] / np.float32(int(sup0[y, x]) >> 20)
    return out
