// Builder of the per-patch aggregation programs that the assembly kernel of csrc/asm/cbca_prog_gen.py interprets
// (a4, /root/reference/src/process_functional.py:149-163 in the reference's summation order).
//
// A patch is K x G anchors (rows y0 .., columns x0 ..).  Its program lists, in execution order, what cbca_hwd_kernel
// (cbca_hwd.hip) decides with scalar code on every launch: which region rows the patch sweeps (one descending sweep
// from its last anchor row up to the highest row any anchor reaches, one ascending sweep below - pf:155: self, up 1..,
// then down 1..), which window of pixels each row needs (LOAD), and per anchor column the arms to add: a descending
// run (the column's own pixel, then left 1, 2, ..) and an ascending run (right 1, 2, ..) - pf:157-160.  All of it is a
// function of the image alone, so it is built once per image and serves every disparity chunk of all 18 iterations.
//
// One function for the device (cbca_prog.hip: one thread per patch) and for the host (the CPU tests compile this
// header with g++ and compare it word for word with the plain-Python statement tests/asmtools/cbca_prog_ref.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MCCNN_PROG_HD __host__ __device__ __forceinline__
#else
#define MCCNN_PROG_HD inline
#endif

namespace mccnn {
namespace prog {

constexpr int R = 13;        // longest arm (distance threshold L <= 14)
constexpr int MAXG = 8;      // anchors per row a layout may have
constexpr int MAXK = 4;      // anchor rows a layout may have
constexpr int MAXW = 32;     // window slots a layout may have

// The code layout of one assembled kernel (filled from the generated cbca_prog_layout_v*.h): byte offsets of the op
// handlers from the kernel's code_base.
struct Layout {
    int VPL, RS, K, G, W, MAXD, MAXA, BLK, M0, refill, end;
    int add[MAXG][1 << MAXK][2];     // [column][anchor set][0 descending, 1 ascending]: first block of the line, or -1
    int load[MAXW + 1];              // [n]: LOAD of n slots
};

MCCNN_PROG_HD int band_rows_of(int H, int K)
{
    const int per = (H + 7) / 8;
    return (per + K - 1) / K * K;
}

// Upper bound of a patch's program in dwords: per region row at most 2 G windows (every column alone, both
// directions; an arm longer than the window comes in ceil(14 / W) pieces with a window each) and as many arm runs,
// each of which splits into at most max(1, K/2) aligned anchor groups; one REFILL per 63 ops; END; rounded to whole
// 64-op chunks.
MCCNN_PROG_HD int stride_dwords(int K, int G, int W)
{
    const int rows = 2 * K + 2 * R - 1;
    const int groups = K >= 2 ? K / 2 : 1;
    const int pieces = (R + 1 + W - 1) / W;
    int n = rows * pieces * (2 * G + 2 * G * groups) + 2;
    n += n / 63 + 1;
    return (n + 63) / 64 * 64;
}

// Raw op i of a program sits at dword i + i / 63: every 64-op chunk ends with a REFILL (the kernel keeps 64 ops per
// lane-register and fetches the next chunk when it meets that op).
struct CountEmitter {
    int n;
    MCCNN_PROG_HD void op(uint32_t) { ++n; }
};
struct WriteEmitter {
    uint32_t *out;
    int i, cap;
    uint32_t refill;
    MCCNN_PROG_HD void op(uint32_t v)
    {
        const int f = i + i / 63;
        if (f < cap) {
            if (i > 0 && i % 63 == 0) out[f - 1] = refill;
            out[f] = v;
        }
        ++i;
    }
};
MCCNN_PROG_HD int dwords_of(int raw_ops) { return raw_ops == 0 ? 0 : (raw_ops - 1) + (raw_ops - 1) / 63 + 1; }

// The anchors of a patch: vertical arms clamped to the image, and the two sweeps they span.
struct Patch {
    int y0, x0, row0, nd, na;
    // bit t: anchor (k, j) takes part in sweep step t.  Descending step s visits row y0 + K - 1 - s: anchor k is in for
    // s in [K-1-k, K-1-k+up]; ascending step a (= step nd + a) visits row y0 + 1 + a: for a in [k, k+dn-1].
    uint64_t sched[MAXK][MAXG];
};
MCCNN_PROG_HD uint64_t sched_of(bool ok, int K, int k, int up, int dn, int nd)
{
    if (!ok) return 0;
    const uint64_t d = ((2ull << up) - 1ull) << (K - 1 - k);
    const uint64_t a = (((1ull << dn) - 1ull) << k) << nd;
    return d | a;
}

// sup0: plane 0 of the support buffer ([H][W] words: bits 0-4 up, 5-9 down, 10-14 left, 15-19 right).
// skip_unit: anchors whose support region is the pixel itself (all four arms 0) take no part.  (0 + x) / 1 = x (pf:156-
// 161 with aver_num = 1) up to the sign of a zero, so after the first iteration of a ping-pong pair both buffers hold
// a value of such a pixel that every later sum treats alike (include/mccnn.h, mccnn_cbca_iter_prog_pair_skip): the
// "skip" programs neither read it for its own sake nor does the skip kernel write it.
MCCNN_PROG_HD bool unit_region(uint32_t word) { return (word & 0xfffffu) == 0u; }

MCCNN_PROG_HD void patch_setup(const Layout &L, const uint32_t *sup0, int H, int W, int y0, int x0, Patch &P,
                               bool skip_unit = false)
{
    const int K = L.K, G = L.G;
    int up[MAXK][MAXG], dn[MAXK][MAXG];
    bool ok[MAXK][MAXG];
    int lowest = y0, highest = y0;
    for (int k = 0; k < K; ++k) {
        const int y = y0 + k;
        for (int j = 0; j < G; ++j) {
            const int x = x0 + j;
            ok[k][j] = x < W && y < H;
            up[k][j] = dn[k][j] = 0;
            if (!ok[k][j]) continue;
            const uint32_t a = sup0[(size_t)y * W + x];
            if (skip_unit && unit_region(a)) {
                ok[k][j] = false;
                continue;
            }
            const int u = (int)(a & 31u), d = (int)((a >> 5) & 31u);
            up[k][j] = u < y ? u : y;
            dn[k][j] = d < H - 1 - y ? d : H - 1 - y;
            if (y - up[k][j] < lowest) lowest = y - up[k][j];
            const int reach = dn[k][j] > 0 ? y + dn[k][j] : y0;
            if (reach > highest) highest = reach;
        }
    }
    P.y0 = y0;
    P.x0 = x0;
    P.nd = y0 + K - 1 - lowest + 1;
    P.na = highest - y0;
    P.row0 = y0 - R > 0 ? y0 - R : 0;
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < G; ++j) P.sched[k][j] = sched_of(ok[k][j], K, k, up[k][j], dn[k][j], P.nd);
}

// Per-row work arrays with run-time indices: plain stack arrays on the host, strided LDS on the device (private
// "scratch" arrays cost a ~300-clock memory round trip per access there).
template <class T>
struct ArrT {
    T *p;
    int s;
    MCCNN_PROG_HD T &operator[](int i) const { return p[i * s]; }
};
// every entry is an anchor-row set (< 2^MAXK), a column index or an arm length: bytes on the device (10 KB of LDS per
// wave as ints kept the builder at two waves per SIMD), ints on the host
template <class T>
struct RowTmpT {
    ArrT<T> aset, cols, l, r, acols;
};
using Arr = ArrT<int>;
using RowTmp = RowTmpT<int>;

// The ops of sweep step t (one region row): its windows and arm runs, in execution order.
template <class E, class TT>
MCCNN_PROG_HD void emit_row(const Layout &L, const Patch &P, const uint32_t *sup0, int W, int t, const RowTmpT<TT> &T, E &e)
{
    // every scalar of the layout and of the patch into registers first: on the device both live in LDS, next to the work
    // arrays this function writes, so the compiler would otherwise reload them after every store
    const int K = L.K, G = L.G, WW = L.W, MAXD = L.MAXD, MAXA = L.MAXA, BLK = L.BLK, RS = L.RS;
    const uint32_t M0 = (uint32_t)L.M0;
    const int y0 = P.y0, x0 = P.x0, nd = P.nd, row0 = P.row0;
    {
        const int yq = t < nd ? y0 + K - 1 - t : y0 + 1 + (t - nd);
        // anchors taking part in this step
        const ArrT<TT> aset = T.aset, cols = T.cols;
        int ncols = 0;
        for (int j = 0; j < G; ++j) {
            int m = 0;
            for (int k = 0; k < K; ++k) m |= (int)((P.sched[k][j] >> t) & 1ull) << k;
            aset[j] = (TT)m;
            if (m) cols[ncols++] = (TT)j;
        }
        if (ncols == 0) return;
        const ArrT<TT> l = T.l, r = T.r;
        // the row's G support words: unconditional, independent loads (words right of the image are never used and lie
        // inside the support buffer, whose derived planes follow plane 0); their arm lengths, clamped to R
        for (int j = 0; j < G; ++j) {
            const uint32_t a = sup0[(size_t)yq * W + x0 + j];
            const int lj = (int)((a >> 10) & 31u), rj = (int)((a >> 15) & 31u);
            l[j] = (TT)(lj > R ? R : lj);
            r[j] = (TT)(rj > R ? R : rj);
        }
        int lo = 1 << 30, hi = -1, maxl = 0, maxr = 0;
        for (int i = 0; i < ncols; ++i) {
            const int j = cols[i];
            const int c = j + R;
            if (c - l[j] < lo) lo = c - l[j];
            if (c + r[j] > hi) hi = c + r[j];
            if (l[j] > maxl) maxl = l[j];
            if (r[j] > maxr) maxr = r[j];
        }
        auto load = [&](int wlo, int whi) {
            const int n = whi - wlo + 1;
            const uint32_t p = (uint32_t)((yq - row0) * W + (x0 - R + whi));
            e.op((uint32_t)L.load[n] | (p << 16));
        };
        // one arm run of column j: `first` = virtual slot of its first element, window starts at virtual slot wlo
        auto run = [&](int j, int wlo, int first, int n, int dir) {
            const int sf = first - wlo;
            int rest = aset[j];
            for (int s = K; s >= 1 && rest; s >>= 1)
                for (int st = 0; st < K; st += s) {
                    const int m = ((1 << s) - 1) << st;
                    if ((rest & m) != m) continue;
                    rest &= ~m;
                    const int maxn = dir ? MAXA : MAXD;
                    const int idx = dir ? sf + n - 1 : sf - n + 1;
                    e.op((uint32_t)(L.add[j][m][dir] + (maxn - n) * BLK * s) | ((M0 | (uint32_t)(RS * idx)) << 16));
                }
        };
        if (hi - lo + 1 <= WW && maxl + 1 <= MAXD && maxr <= MAXA) {
            load(lo, hi);
            for (int i = 0; i < ncols; ++i) run(cols[i], lo, cols[i] + R, l[cols[i]] + 1, 0);
            for (int i = 0; i < ncols; ++i)
                if (r[cols[i]]) run(cols[i], lo, cols[i] + R + 1, r[cols[i]], 1);
            return;
        }
        // wide row: windows over groups of neighbouring columns, descending arms first
        const int capd = WW < MAXD ? WW : MAXD, capa = WW < MAXA ? WW : MAXA;
        for (int i = 0; i < ncols;) {
            const int j = cols[i];
            int glo = j + R - l[j], ghi = j + R;
            if (ghi - glo + 1 > capd) {                       // one arm longer than a window: in pieces
                int first = j + R, left = l[j] + 1;
                while (left) {
                    const int n = left < capd ? left : capd;
                    load(first - n + 1, first);
                    run(j, first - n + 1, first, n, 0);
                    first -= n;
                    left -= n;
                }
                ++i;
                continue;
            }
            int cnt = 1;
            while (i + cnt < ncols) {
                const int j2 = cols[i + cnt];
                const int nlo = glo < j2 + R - l[j2] ? glo : j2 + R - l[j2];
                const int nhi = ghi > j2 + R ? ghi : j2 + R;
                if (nhi - nlo + 1 > WW || l[j2] + 1 > MAXD) break;
                glo = nlo;
                ghi = nhi;
                ++cnt;
            }
            load(glo, ghi);
            for (int q = 0; q < cnt; ++q) run(cols[i + q], glo, cols[i + q] + R, l[cols[i + q]] + 1, 0);
            i += cnt;
        }
        const ArrT<TT> acols = T.acols;
        int nac = 0;
        for (int i = 0; i < ncols; ++i)
            if (r[cols[i]]) acols[nac++] = cols[i];
        for (int i = 0; i < nac;) {
            const int j = acols[i];
            int glo = j + R + 1, ghi = j + R + r[j];
            if (ghi - glo + 1 > capa) {
                int first = j + R + 1, left = r[j];
                while (left) {
                    const int n = left < capa ? left : capa;
                    load(first, first + n - 1);
                    run(j, first, first, n, 1);
                    first += n;
                    left -= n;
                }
                ++i;
                continue;
            }
            int cnt = 1;
            while (i + cnt < nac) {
                const int j2 = acols[i + cnt];
                const int nlo = glo < j2 + R + 1 ? glo : j2 + R + 1;
                const int nhi = ghi > j2 + R + r[j2] ? ghi : j2 + R + r[j2];
                if (nhi - nlo + 1 > WW || r[j2] > MAXA) break;
                glo = nlo;
                ghi = nhi;
                ++cnt;
            }
            load(glo, ghi);
            for (int q = 0; q < cnt; ++q) run(acols[i + q], glo, acols[i + q] + R + 1, r[acols[i + q]], 1);
            i += cnt;
        }
    }
}

// The whole program of a patch, sequentially (host tests; the device kernel deals the rows to lanes, cbca_prog.hip).
// Returns the number of dwords the program has (> cap: it did not fit, which the stride bound excludes).
MCCNN_PROG_HD int build_patch(const Layout &L, const uint32_t *sup0, int H, int W, int y0, int x0, uint32_t *out, int cap,
                              bool skip_unit = false)
{
    Patch P;
    patch_setup(L, sup0, H, W, y0, x0, P, skip_unit);
    WriteEmitter e = {out, 0, cap, (uint32_t)L.refill | ((uint32_t)L.M0 << 16)};
    int buf[5][MAXG];
    const RowTmp T = {{buf[0], 1}, {buf[1], 1}, {buf[2], 1}, {buf[3], 1}, {buf[4], 1}};
    for (int t = 0; t < P.nd + P.na; ++t) emit_row(L, P, sup0, W, t, T, e);
    e.op((uint32_t)L.end | ((uint32_t)L.M0 << 16));
    return dwords_of(e.i);
}


}  // namespace prog
}  // namespace mccnn
