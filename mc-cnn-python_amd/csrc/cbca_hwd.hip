// a4 cost_volume_aggregation (/root/reference/src/process_functional.py:117-183) in the reference's own summation
// order - bit-exact - on the pixel-major "HWD" layout [H][W][Dp], plus the two volume readers that let the whole
// bit-exact pair stay pixel-major (a7 WTA pf:239-272, a9 sub-pixel pf:381-400).
//
//   out[p, d] = ( sum_{q in Vert(p): self, up 1..u, down 1..dn}  sum_{r in Horiz(q): self, left 1..l, right 1..r}  in[r, d] ) / |U(p)|
// as ONE flat float32 running sum per (p, d) starting from 0 (pf:149-163): the order of the additions is part of the
// result, so every output is a dependent chain of |U(p)| adds (mean 30, up to 27 x 27 = 729 on the synthetic pair).
//
// Why disparities on lanes.  The plane-major reference-order kernel (cbca_ref4_kernel, cross_cbca.hip) gives every lane
// its own pixel and pays for it with a divergent walk: a wave waits for its longest region, each region row starts
// with a dependent load of that row's arms, and every element costs loop bookkeeping in vector instructions - 1.87 ms
// per volume iteration at 750x500x256, 5 % of the HBM roofline, for ~0.04 ms worth of float32 adds.  The support
// region does not depend on the disparity.  With the volume pixel-major, a wave takes a pixel and ALL its disparities
// (lane l owns d = 4 l .. 4 l + 3, exactly sgm_pass_kernel's mapping): the walk is wave-uniform, i.e. it runs on the
// scalar unit (arms by s_load, loop counters and addresses in SGPRs, s_cbranch per element), no lane ever waits for
// another, every region element is one coalesced 1 KiB buffer_load_dwordx4 with a wave-uniform offset, and the loads
// are independent of the add chains, so a whole region row is in flight at once.
//
// Why G pixels per wave.  Done pixel by pixel, every add needs its own 1 KiB operand from the vector memory pipe
// (64 B/clk/CU): 375 K pixels x 30 elements x 1 KiB = 11.5 GB per volume iteration, 0.3 ms at the pipe's peak.  G
// horizontally adjacent pixels share most of every region row, and they visit the rows in the same order (self, up
// 1.., down 1..; a pixel simply sits out the rows beyond its own vertical arm).  So a wave owns G neighbours: per
// row it loads the union of their horizontal arms ONCE into a register window (slot = column, statically indexed -
// the chains are fully unrolled and every element is guarded by a scalar branch), then runs each pixel's chain over
// the window in that pixel's own order.  Same additions, same order, per (pixel, d): bit-exact.  Loads per pixel and
// row fall from ~6 to ~9/4.
//
// HBM sees each voxel once (8 B/voxel/iteration, like the streaming kernel): the re-reads are served by L2.  The
// launch is XCD-aware for that: workgroup b runs on XCD b % 8, every XCD owns a band of image rows, and inside a band
// the workgroups sweep column by column (4 rows x G columns per workgroup), so that the ~400 waves an XCD has
// resident cover a compact window of the image whose rows stay in that XCD's 4 MiB L2 while they are being reused.
#include "support.h"

namespace mccnn {
namespace hw {

constexpr int R = HWD_R;               // longest arm served (distance threshold L <= 14)
constexpr int G = HWD_G;               // pixels per wave (support.h: the window masks are built for it)
constexpr int NW = HWD_NW;             // window slots: columns x0 - R .. x0 + G - 1 + R
constexpr int kDrop = 0x7ffffff0;      // byte offset past every buffer: the range check drops the access
typedef hwd_mask_t wm_t;               // window mask: one bit per slot
constexpr wm_t kOne = 1;

#ifndef CBCA_HWD_WPB
#define CBCA_HWD_WPB 1                 // waves (image rows) per workgroup
#endif
#ifndef CBCA_HWD_NTS
#define CBCA_HWD_NTS 2                 // aux bits of the result stores (2 = non-temporal)
#endif
#ifndef CBCA_HWD_ABL
#define CBCA_HWD_ABL 0                 // timing-only ablations (wrong results): 1 no adds, 2 no window loads, 3 neither
#endif
#ifndef CBCA_HWD_ONLY
#define CBCA_HWD_ONLY 0
#endif
#ifndef CBCA_HWD_BLOCK4
#define CBCA_HWD_BLOCK4 0
#endif
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VPL> struct Vec;
template <> struct Vec<4> {
    struct T { float x, y, z, w; };     // four scalars, not a vector type: see add()
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        T v;
        v.x = __uint_as_float(u.x); v.y = __uint_as_float(u.y); v.z = __uint_as_float(u.z); v.w = __uint_as_float(u.w);
        return v;
    }
    static __device__ __forceinline__ void store(T v, __amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        u32x4 u;
        u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
        buffer_store_b128<CBCA_HWD_NTS>(u, rs, voff, soff);
    }
    static __device__ __forceinline__ T zero() { T v = {0.f, 0.f, 0.f, 0.f}; return v; }
    // a += w as four v_add_f32: this file is built with -fno-slp-vectorize (Makefile), because clang otherwise packs the
    // four adds into two v_pk_add_f32, which gfx950 issues at a fraction of the plain rate - same IEEE sums either way.
    static __device__ __forceinline__ void add(T &a, const T &w)
    {
#if (CBCA_HWD_ABL & 1)
        a.x += w.x;                        // one add instead of four
#else
        a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w;
#endif
    }
    static __device__ __forceinline__ T div(const T &a, float n) { T v = {a.x / n, a.y / n, a.z / n, a.w / n}; return v; }
};
// three disparities per lane: D = 192 (KITTI) fills all 64 lanes instead of 48 of them, and the window is a quarter
// smaller (123 registers: four waves per SIMD instead of three)
template <> struct Vec<3> {
    struct T { float x, y, z; };
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        const u32x3 u = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
        T v;
        v.x = __uint_as_float(u.x); v.y = __uint_as_float(u.y); v.z = __uint_as_float(u.z);
        return v;
    }
    static __device__ __forceinline__ void store(T v, __amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        u32x3 u;
        u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z);
        buffer_store_b96<CBCA_HWD_NTS>(u, rs, voff, soff);
    }
    static __device__ __forceinline__ T zero() { T v = {0.f, 0.f, 0.f}; return v; }
    static __device__ __forceinline__ void add(T &a, const T &w) { a.x += w.x; a.y += w.y; a.z += w.z; }
    static __device__ __forceinline__ T div(const T &a, float n) { T v = {a.x / n, a.y / n, a.z / n}; return v; }
};
template <> struct Vec<2> {
    struct T { float x, y; };
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        T v;
        v.x = __uint_as_float(u.x); v.y = __uint_as_float(u.y);
        return v;
    }
    static __device__ __forceinline__ void store(T v, __amdgpu_buffer_rsrc_t rs, int voff, unsigned soff)
    {
        u32x2 u;
        u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y);
        __builtin_amdgcn_raw_buffer_store_b64(u, rs, voff, soff, CBCA_HWD_NTS);
    }
    static __device__ __forceinline__ T zero() { T v = {0.f, 0.f}; return v; }
    static __device__ __forceinline__ void add(T &a, const T &w) { a.x += w.x; a.y += w.y; }
    static __device__ __forceinline__ T div(const T &a, float n) { T v = {a.x / n, a.y / n}; return v; }
};

// The chains.  J (pixel of the group) and Z (distance along the arm) are template parameters, so every window access
// has a compile-time index (the window lives in registers) and the control is scalar bit tests on the pixel's window
// mask m (zero when the pixel sits the row out).  A wave runs this branchy serial code at one instruction per ~13
// cycles, and the scalar test + branch of an element cost as much as its two packed adds, so long arms are taken four
// elements per test (one s_andn2 against a 4-bit pattern) and a pixel whose region row is the pixel itself - the most
// common case - leaves after one compare.
template <int VPL, int J, int Z, int DIR>   // DIR = -1: left arm (slots below J + R), +1: right arm
__device__ __forceinline__ void walk_arm(typename Vec<VPL>::T &a, const typename Vec<VPL>::T (&win)[NW], wm_t m)
{
    if constexpr (Z <= R) {
        if (m & (kOne << (J + R + DIR * Z))) {
            if constexpr (CBCA_HWD_BLOCK4 && Z + 3 <= R) {
                // bits of elements Z .. Z+3
                constexpr wm_t four = DIR > 0 ? ((wm_t)0xF << (J + R + Z)) : ((wm_t)0xF << (J + R - Z - 3));
                if ((~m & four) == 0) {
                    Vec<VPL>::add(a, win[J + R + DIR * Z]);
                    Vec<VPL>::add(a, win[J + R + DIR * (Z + 1)]);
                    Vec<VPL>::add(a, win[J + R + DIR * (Z + 2)]);
                    Vec<VPL>::add(a, win[J + R + DIR * (Z + 3)]);
                    walk_arm<VPL, J, Z + 4, DIR>(a, win, m);
                } else {                                   // the arm ends within the next three elements
                    Vec<VPL>::add(a, win[J + R + DIR * Z]);
                    if (m & (kOne << (J + R + DIR * (Z + 1)))) {
                        Vec<VPL>::add(a, win[J + R + DIR * (Z + 1)]);
                        if (m & (kOne << (J + R + DIR * (Z + 2)))) Vec<VPL>::add(a, win[J + R + DIR * (Z + 2)]);
                    }
                }
            } else {
                Vec<VPL>::add(a, win[J + R + DIR * Z]);
                walk_arm<VPL, J, Z + 1, DIR>(a, win, m);
            }
        }
    }
}
// pf:157-161 for pixel J on one region row: self, left 1.., right 1..
template <int VPL, int J>
__device__ __forceinline__ void walk_rows(typename Vec<VPL>::T (&acc)[G], const typename Vec<VPL>::T (&win)[NW],
                                          const wm_t (&m)[G])
{
    if constexpr (J < G) {
        if (m[J] != 0) {
            Vec<VPL>::add(acc[J], win[J + R]);
            if (m[J] != (kOne << (J + R))) {
                walk_arm<VPL, J, 1, -1>(acc[J], win, m[J]);
                walk_arm<VPL, J, 1, +1>(acc[J], win, m[J]);
            }
        }
        walk_rows<VPL, J + 1>(acc, win, m);
    }
}

// Loads the window slots whose bits are set in `u` (the OR of the row's masks).  Two levels of scalar tests - a
// nibble of four slots, then its slots - because a typical row touches 8-10 of the 30 slots: ~20 scalar instructions
// instead of 3 per slot.  (A jump to the first slot of the run with fall-through from slot to slot would be cheaper
// still, but the compiler structurises that switch into a state machine of 64-bit flags.)
template <int VPL, int N>
__device__ __forceinline__ void load_window(typename Vec<VPL>::T (&win)[NW], wm_t u, __amdgpu_buffer_rsrc_t rs,
                                            int voff, unsigned rowoff, unsigned pix)
{
    if constexpr (4 * N < NW) {
        if (u & ((wm_t)0xF << (4 * N))) {
            unsigned off = rowoff + (unsigned)(4 * N) * pix;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (4 * N + i < NW) {
#if (CBCA_HWD_ABL & 2)
                    if (u & (kOne << (4 * N + i))) {
                        win[4 * N + i] = Vec<VPL>::zero();
                        win[4 * N + i].x = __int_as_float(voff + (int)off);   // a value the compiler cannot fold
                    }
#else
                    if (u & (kOne << (4 * N + i))) win[4 * N + i] = Vec<VPL>::load(rs, voff, off);
#endif
                    off += pix;
                }
            }
        }
        load_window<VPL, N + 1>(win, u, rs, voff, rowoff, pix);
    }
}

// wave-wide reductions of the WTA (also used by the fused last iteration below)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_min_f(float x)
{
    // lanes without a source keep their own value (old = x)
    auto step = [](float v, int o) { return fminf(v, __int_as_float(o)); };
    x = step(x, dpp_i<0xB1>(__float_as_int(x), __float_as_int(x)));         // quad_perm [1,0,3,2]
    x = step(x, dpp_i<0x4E>(__float_as_int(x), __float_as_int(x)));         // quad_perm [2,3,0,1]
    x = step(x, dpp_i<0x141>(__float_as_int(x), __float_as_int(x)));        // row_half_mirror
    x = step(x, dpp_i<0x140>(__float_as_int(x), __float_as_int(x)));        // row_mirror
    x = step(x, dpp_i<0x142, 0xA>(__float_as_int(x), __float_as_int(x)));   // row_bcast:15
    x = step(x, dpp_i<0x143, 0xC>(__float_as_int(x), __float_as_int(x)));   // row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ int wave_min_i(int x)
{
    x = min(x, dpp_i<0xB1>(x, x));
    x = min(x, dpp_i<0x4E>(x, x));
    x = min(x, dpp_i<0x141>(x, x));
    x = min(x, dpp_i<0x140>(x, x));
    x = min(x, dpp_i<0x142, 0xA>(x, x));
    x = min(x, dpp_i<0x143, 0xC>(x, x));
    return __builtin_amdgcn_readlane(x, 63);
}

// One launch aggregates up to two volumes of the same shape (left and right view, each with its own support plane).
struct Jobs {
    const float *in[2];
    float *out[2];
    const Support *sup[2];
    int n;
    // fused WTA of the last iteration (WTA kernels only): the first strict minimum over d of every output pixel
    // (pf:245-254) goes to disp[job]; a volume whose store[job] is 0 is not written at all
    float *disp[2];
    int store[2];
    int D;
};

// A wave owns a patch of K x G anchors: rows y0 .. y0 + K - 1 (y0 a multiple of K), columns x0 .. x0 + G - 1.
// Vertically adjacent anchors walk almost the same rows, but each in its own order (self, up 1.., down 1..), so the
// rows are visited in ONE descending sweep y0 + K - 1 ... lowest row any anchor reaches - anchor k joins at its own
// row y0 + k (its "self") and stays until its upper arm ends - and then ONE ascending sweep y0 + 1 ... highest row -
// anchor k joins at y0 + k + 1, after its upper arm is complete, and stays until its lower arm ends.  Every anchor
// still sees self, up 1.., down 1.. (pf:155), and a row's window is loaded once for the K anchors instead of K times:
// in long-arm zones, where the vector memory pipe (64 B/clk/CU) is what bounds this kernel, that is ~K times fewer
// bytes per output.  The K anchors of a column meet the same row pixel, i.e. the same window mask; bit t of
// sched[k][j] says whether anchor (k, j) takes part in step t of the two sweeps.
//
// grid = (8 * nyb, ngroups, nchunks * jobs): blockIdx.x & 7 = XCD = band of `band_rows` image rows (a multiple of
// K * WPB), blockIdx.x >> 3 = group of K * WPB rows inside the band (K rows per wave), blockIdx.y = group of G
// columns, blockIdx.z = (job, chunk of 64 * VPL disparities).  Dispatch order is x fastest, then y: inside its band
// an XCD sweeps column group by column group.
#ifndef CBCA_HWD_MINW
#define CBCA_HWD_MINW 1
#endif
#ifndef CBCA_HWD_K
#define CBCA_HWD_K 2
#endif
constexpr int K = CBCA_HWD_K;
static_assert(2 * K + 2 * R - 1 <= 32, "the sweep schedule is a 32-bit word");

template <int VPL, int KK>
__device__ __forceinline__ void walk_anchor_rows(typename Vec<VPL>::T (&acc)[K][G], const typename Vec<VPL>::T (&win)[NW],
                                                 const wm_t (&roww)[G], const uint32_t (&sched)[K][G], int t)
{
    if constexpr (KK < K) {
        wm_t m[G];
#pragma unroll
        for (int j = 0; j < G; ++j) m[j] = (sched[KK][j] >> t) & 1u ? roww[j] : (wm_t)0;
        walk_rows<VPL, 0>(acc[KK], win, m);
        walk_anchor_rows<VPL, KK + 1>(acc, win, roww, sched, t);
    }
}

template <int VPL, bool WTA = false>
__global__ __launch_bounds__(64 * CBCA_HWD_WPB, CBCA_HWD_MINW) void cbca_hwd_kernel(const Jobs jobs, int Dp, int H, int W, int nchunks, int band_rows)
{
    typedef typename Vec<VPL>::T vf;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int y0 = (int)(blockIdx.x & 7) * band_rows + ((int)(blockIdx.x >> 3) * CBCA_HWD_WPB + wv) * K;
    if (y0 >= H) return;
    const int x0 = (int)blockIdx.y * G;
    const int job = (int)blockIdx.z / nchunks, chunk = (int)blockIdx.z - job * nchunks;
    const float *const in = job ? jobs.in[1] : jobs.in[0];
    float *const out = job ? jobs.out[1] : jobs.out[0];
    const Support *__restrict__ const sup = job ? jobs.sup[1] : jobs.sup[0];
    const wm_t *__restrict__ const wmask =
        reinterpret_cast<const wm_t *>(reinterpret_cast<const char *>(sup) + wmask_plane_offset(H, W));

    const unsigned pix = (unsigned)Dp * 4u;                       // bytes between neighbouring pixels
    const int ytop = min(y0 + K - 1, H - 1);                       // last anchor row inside the image
    const int row0 = max(y0 - R, 0), row1 = min(ytop + R, H - 1);  // rows any arm of this patch can reach
    const size_t rowf = (size_t)W * Dp;                            // floats per image row
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in + (size_t)row0 * rowf), 0, (int)((size_t)(row1 - row0 + 1) * rowf * 4), 0x00020000);
    const int d0 = (chunk * 64 + lane) * VPL;
    const int voff = d0 < Dp ? d0 * 4 : kDrop;                     // lanes past the disparity range: loads 0, stores dropped

    // anchors: vertical arms (plane 0) -> the sweep schedule.  The words of a patch that straddles the right or the
    // bottom edge are read from clamped rows / inside the support buffer and never used: their schedule is empty.
    uint32_t sched[K][G];
    int lowest = y0, highest = y0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int y = y0 + k;
        const size_t p0 = (size_t)min(y, H - 1) * W + x0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint32_t aw = sup[p0 + j];
            const bool ok = x0 + j < W && y < H;
            const int up = min(arm_up(aw), y), dn = min(arm_down(aw), H - 1 - y);   // clamped to the image: the scalar
            if (ok) {                                                                // loads below stay inside the plane
                lowest = min(lowest, y - up);
                highest = max(highest, dn > 0 ? y + dn : y0);
            }
            // descending step s visits row y0 + K - 1 - s: anchor k takes part for s in [K-1-k, K-1-k+up];
            // ascending step a visits row y0 + 1 + a: for a in [k, k+dn-1].  The ascending bits are placed once the
            // number of descending steps is known (below).
            sched[k][j] = ok ? (((2u << up) - 1u) << (K - 1 - k)) | ((((1u << dn) - 1u) << k) << 16) : 0u;
        }
    }
    const int nd = y0 + K - 1 - lowest + 1;                        // descending steps: rows y0+K-1 .. lowest
    const int na = highest - y0;                                   // ascending steps: rows y0+1 .. highest
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < G; ++j) sched[k][j] = (sched[k][j] & 0xFFFFu) | ((sched[k][j] >> 16) << nd);
    const int nsteps = nd + na;
#if CBCA_HWD_ONLY
    {   // timing-only diagnostic: 1 = only the patches whose anchors all have four zero arms run, 2 = only the others
        uint32_t arms = 0;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < G; ++j) arms |= sup[(size_t)min(y0 + k, H - 1) * W + x0 + j] & 0xFFFFFu;
        if ((arms == 0) != (CBCA_HWD_ONLY == 1)) return;
    }
#endif
    auto row_of = [&](int t) { return t < nd ? y0 + K - 1 - t : y0 + 1 + (t - nd); };

    vf acc[K][G];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[k][j] = Vec<VPL>::zero();  // pf:156 sum starts at 0

    wm_t nxt[G];
    {
        const size_t pn = (size_t)min(row_of(0), H - 1) * W + x0;
#pragma unroll
        for (int j = 0; j < G; ++j) nxt[j] = wmask[pn + j];
    }
    for (int t = 0; t < nsteps; ++t) {
        wm_t roww[G], u = 0;
        uint32_t any[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            roww[j] = nxt[j];
            any[j] = 0u;
#pragma unroll
            for (int k = 0; k < K; ++k) any[j] |= sched[k][j];
            u |= (any[j] >> t) & 1u ? roww[j] : (wm_t)0;
        }
        const int yq = row_of(t);
        // slot k = column x0 - R + k; the offset may wrap below zero for slots left of the image, which no arm reaches
#if (CBCA_HWD_ABL & 4)
        const unsigned rowoff = (unsigned)((((yq - row0) & 1) * W + (x0 % 60) + 16 - R) * (int)pix);   // timing only: every load hits a cache
#else
        const unsigned rowoff = (unsigned)(((yq - row0) * W + x0 - R) * (int)pix);
#endif
        {   // the next row's masks travel while this row is loaded and summed (past the end: a harmless re-read)
            const size_t pn = (size_t)min(row_of(min(t + 1, nsteps - 1)), H - 1) * W + x0;
#pragma unroll
            for (int j = 0; j < G; ++j) nxt[j] = wmask[pn + j];
        }
        vf win[NW];
        load_window<VPL, 0>(win, u, rs_in, voff, rowoff, pix);
        walk_anchor_rows<VPL, 0>(acc, win, roww, sched, t);
    }
    // Epilogue: every region size first (scalar loads: after the first store the compiler no longer trusts `sup` to be
    // unchanged and turns each later read into a vector load + a full wait), then every quotient into registers of
    // its own, then the stores back to back behind a scheduling barrier.  The barrier is load-bearing: with divisions
    // and stores interleaved, the next column's division reuses the registers a buffer_store_dwordx4 is still
    // reading (the compiler leaves the 2 wait states it knows about; on gfx950 the store then picked up the next
    // division's intermediate in lanes 12-15 of every 16 - observed as wrong first components, tools/dev_hwd_where.py).
    float cnt[K][G];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t p0 = (size_t)min(y0 + k, H - 1) * W + x0;
#pragma unroll
        for (int j = 0; j < G; ++j) cnt[k][j] = (float)sup_count(sup[p0 + j]);
    }
    vf res[K][G];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < G; ++j) res[k][j] = Vec<VPL>::div(acc[k][j], cnt[k][j]);   // pf:161
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int y = y0 + k;
        // the descriptor ends with the row (or the patch): stores of columns past the right edge are dropped by its
        // range check, so the loop needs no per-column validity (their counts are whatever word follows: never used);
        // a row below the image gets an empty descriptor instead of a branch (a branch here lets the compiler sink that
        // row's divisions behind it, back between the stores)
        const bool wr = y < H && (!WTA || (job ? jobs.store[1] : jobs.store[0]) != 0);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            out + ((size_t)y * W + x0) * Dp, 0, wr ? (int)((unsigned)min(G, W - x0) * pix) : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < G; ++j) Vec<VPL>::store(res[k][j], rs_out, voff, (unsigned)j * pix);
    }
    if constexpr (WTA) {
        // a7 fused into the last iteration: exactly wta_hwd_kernel's two reductions on the values just stored
        // (the minimum, then the lowest index among the lanes that hold it; NaN never wins; -1 when nothing does)
        float *const dsp = job ? jobs.disp[1] : jobs.disp[0];
        const int D = jobs.D;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < G; ++j) {
                float best = __builtin_huge_valf();
                int bd = -1;
                const float *c = &res[k][j].x;
#pragma unroll
                for (int q = 0; q < VPL; ++q)
                    if (d0 + q < D && c[q] < best) { best = c[q]; bd = d0 + q; }
                const float mn = wave_min_f(best);
                const int idx = wave_min_i((best == mn && bd >= 0) ? bd : 0x7fffffff);
                if (lane == 0 && y0 + k < H && x0 + j < W)
                    dsp[(size_t)(y0 + k) * W + x0 + j] = idx == 0x7fffffff ? -1.f : (float)idx;
            }
    }
}

static int launch(const Jobs &jobs, int D, int H, int W, hipStream_t s, bool wta = false)
{
    const int Dp = mccnn_hwd_pitch(D);
    MCCNN_REQUIRE((size_t)(2 * R + K) * W * Dp * 4 < ((size_t)1 << 31), MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter_hwd: %d columns x %d disparities exceed a buffer descriptor's reach", W, D);
#ifdef CBCA_HWD_VPL
    const int vpl = CBCA_HWD_VPL;
#else
    // 2 disparities per lane up to 128, 3 where that fills the lanes exactly (Dp a multiple of 3 up to 192: a lane
    // must not straddle two pixels), 4 per lane and 256-disparity chunks otherwise
    const int vpl = Dp <= 128 ? 2 : (Dp <= 192 && Dp % 3 == 0) ? 3 : 4;
#endif
    const int nchunks = cdiv(Dp, 64 * vpl);
    const int band_rows = cdiv(cdiv(H, 8), hw::K * CBCA_HWD_WPB) * hw::K * CBCA_HWD_WPB;
    const int ngroups = cdiv(W, G);
    MCCNN_REQUIRE(ngroups <= 65535 && nchunks * jobs.n <= 65535, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter_hwd: %dx%dx%d exceeds the grid", W, H, D);
    const dim3 grid(8 * (band_rows / (hw::K * CBCA_HWD_WPB)), ngroups, nchunks * jobs.n), block(64 * CBCA_HWD_WPB);
    if (wta) {
        MCCNN_REQUIRE(nchunks == 1, MCCNN_E_UNSUPPORTED,
                      "mccnn_cbca_iter_hwd_pair_wta: D=%d spans more than one chunk of a wave (use mccnn_wta_hwd)", D);
        if (vpl == 4)
            hipLaunchKernelGGL((cbca_hwd_kernel<4, true>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
        else if (vpl == 3)
            hipLaunchKernelGGL((cbca_hwd_kernel<3, true>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
        else
            hipLaunchKernelGGL((cbca_hwd_kernel<2, true>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
    } else if (vpl == 4)
        hipLaunchKernelGGL((cbca_hwd_kernel<4, false>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
    else if (vpl == 3)
        hipLaunchKernelGGL((cbca_hwd_kernel<3, false>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
    else
        hipLaunchKernelGGL((cbca_hwd_kernel<2, false>), grid, block, 0, s, jobs, Dp, H, W, nchunks, band_rows);
    return check_launch("mccnn_cbca_iter_hwd");
}

// ---- a7 on the pixel-major volume: first strict minimum over d (pf:245-254) ----------------------------------------
// One pixel per wave step, 4 disparities per lane and 256-disparity group; the lowest index among equal minima is
// found by a second reduction over the candidates' indices.  8 pixels (8 KiB of loads) in flight per wave.
__global__ __launch_bounds__(256) void wta_hwd_kernel(const float *__restrict__ vol, int D, int Dp, long N,
                                                      float *__restrict__ disp, int per_wave)
{
    constexpr int PF = 8;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long n0 = wave * per_wave, n1 = min(n0 + per_wave, N);
    if (n0 >= N) return;
    const int ng = (Dp + 255) / 256;
    for (long nb = n0; nb < n1; nb += PF) {
        float best[PF];
        int bd[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            best[k] = __builtin_huge_valf();
            bd[k] = -1;
        }
        for (int g = 0; g < ng; ++g) {
            const int d = g * 256 + lane * 4;
            float4 v[PF];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const long n = min(nb + k, n1 - 1);
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (d < Dp) v[k] = *reinterpret_cast<const float4 *>(vol + (size_t)n * Dp + d);
            }
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                if (d + 0 < D && v[k].x < best[k]) { best[k] = v[k].x; bd[k] = d; }
                if (d + 1 < D && v[k].y < best[k]) { best[k] = v[k].y; bd[k] = d + 1; }
                if (d + 2 < D && v[k].z < best[k]) { best[k] = v[k].z; bd[k] = d + 2; }
                if (d + 3 < D && v[k].w < best[k]) { best[k] = v[k].w; bd[k] = d + 3; }
            }
        }
        float res = 0.f;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const float m = wave_min_f(best[k]);
            const int cand = (best[k] == m && bd[k] >= 0) ? bd[k] : 0x7fffffff;
            const int idx = wave_min_i(cand);
            if (lane == k) res = idx == 0x7fffffff ? -1.f : (float)idx;   // pf:254: the index as float32
        }
        if (lane < PF && nb + lane < n1) disp[nb + lane] = res;
    }
}

// ---- a9 on the pixel-major volume (pf:387-396): the same arithmetic as subpixel_kernel (post.hip) -------------------
template <bool NUMPY1>
__global__ __launch_bounds__(256) void subpixel_hwd_kernel(const float *__restrict__ dl, const float *__restrict__ vol,
                                                           int D, int Dp, long N, float *__restrict__ out)
{
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float d = dl[n];
    const int im = (int)(d - 1.f), ip = (int)(d + 1.f), ic = (int)d;
    float res = d;
    if (!(im < 0 || ip >= D)) {
        const float *p = vol + (size_t)n * Dp;
        const float cm = p[im], cp = p[ip], c = p[ic];
        const float num = cp - cm;
        if (NUMPY1) {
            double den = (double)cp - 2.0 * (double)c;
            den = den + (double)cm;
            den = 2.0 * den;
            res = (float)((double)d - (double)num / den);
        } else {
            float den = cp - 2.f * c;
            den = den + cm;
            den = 2.f * den;
            res = d - num / den;
        }
    }
    out[n] = res;
}

}  // namespace hw
}  // namespace mccnn

extern "C" int mccnn_cbca_iter_hwd(const float *in_hwd, float *out_hwd, const mccnn_support_t *support, int D, int H,
                                   int W, int L, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in_hwd && out_hwd && support, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd: null pointer");
    MCCNN_REQUIRE(in_hwd != out_hwd, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd: in-place aggregation is not defined (ping-pong)");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 14, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter_hwd: L=%d outside [1,14] (use mccnn_cbca_iter on the plane-major volume)", L);
    if (const int rc = check_support_record(support, H, W, L, "mccnn_cbca_iter_hwd", true)) return rc;
    const hw::Jobs jobs = {{in_hwd, nullptr}, {out_hwd, nullptr}, {support, nullptr}, 1, {nullptr, nullptr}, {1, 1}, D};
    return hw::launch(jobs, D, H, W, (hipStream_t)stream);
}

extern "C" int mccnn_cbca_iter_hwd_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                        const float *in_right, float *out_right, const mccnn_support_t *support_right,
                                        int D, int H, int W, int L, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in_left && out_left && support_left && in_right && out_right && support_right, MCCNN_E_INVALID,
                  "mccnn_cbca_iter_hwd_pair: null pointer");
    MCCNN_REQUIRE(in_left != out_left && in_right != out_right && out_left != out_right && in_left != out_right &&
                      in_right != out_left,
                  MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair: outputs must not alias an input or each other");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 14, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter_hwd_pair: L=%d outside [1,14]", L);
    int rc = check_support_record(support_left, H, W, L, "mccnn_cbca_iter_hwd_pair", true);
    if (rc) return rc;
    rc = check_support_record(support_right, H, W, L, "mccnn_cbca_iter_hwd_pair", true);
    if (rc) return rc;
    const hw::Jobs jobs = {{in_left, in_right}, {out_left, out_right}, {support_left, support_right}, 2,
                           {nullptr, nullptr}, {1, 1}, D};
    return hw::launch(jobs, D, H, W, (hipStream_t)stream);
}

extern "C" int mccnn_cbca_iter_hwd_pair_wta(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                            const float *in_right, float *out_right, const mccnn_support_t *support_right,
                                            int D, int H, int W, int L, float *disparity_left, float *disparity_right,
                                            int store_right, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in_left && out_left && support_left && in_right && support_right && disparity_left && disparity_right,
                  MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair_wta: null pointer");
    MCCNN_REQUIRE(out_right || !store_right, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair_wta: store_right without out_right");
    MCCNN_REQUIRE(in_left != out_left && in_right != out_right && out_left != out_right && in_left != out_right &&
                      in_right != out_left,
                  MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair_wta: outputs must not alias an input or each other");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter_hwd_pair_wta: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 14, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter_hwd_pair_wta: L=%d outside [1,14]", L);
    int rc = check_support_record(support_left, H, W, L, "mccnn_cbca_iter_hwd_pair_wta", true);
    if (rc) return rc;
    rc = check_support_record(support_right, H, W, L, "mccnn_cbca_iter_hwd_pair_wta", true);
    if (rc) return rc;
    // a volume that is not stored still needs a valid (never dereferenced) base for its empty descriptors
    const hw::Jobs jobs = {{in_left, in_right}, {out_left, store_right ? out_right : out_left},
                           {support_left, support_right}, 2, {disparity_left, disparity_right}, {1, store_right ? 1 : 0}, D};
    return hw::launch(jobs, D, H, W, (hipStream_t)stream, true);
}

extern "C" int mccnn_wta_hwd(const float *vol_hwd, int D, int H, int W, float *disparity, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(vol_hwd && disparity, MCCNN_E_INVALID, "mccnn_wta_hwd: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_wta_hwd: non-positive size");
    const long N = (long)H * W;
    const int per_wave = 64;   // pixels per wave: 8 rounds of 8
    const long waves = (N + per_wave - 1) / per_wave;
    hipLaunchKernelGGL(hw::wta_hwd_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, vol_hwd, D,
                       mccnn_hwd_pitch(D), N, disparity, per_wave);
    return check_launch("mccnn_wta_hwd");
}

extern "C" int mccnn_subpixel_hwd(const float *disp, const float *vol_hwd, int D, int H, int W, int numpy1_promotion,
                                  float *out, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp && vol_hwd && out, MCCNN_E_INVALID, "mccnn_subpixel_hwd: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_subpixel_hwd: non-positive size");
    const long N = (long)H * W;
    const dim3 grid(cdiv(N, 256)), block(256);
    if (numpy1_promotion)
        hipLaunchKernelGGL(hw::subpixel_hwd_kernel<true>, grid, block, 0, (hipStream_t)stream, disp, vol_hwd, D,
                           mccnn_hwd_pitch(D), N, out);
    else
        hipLaunchKernelGGL(hw::subpixel_hwd_kernel<false>, grid, block, 0, (hipStream_t)stream, disp, vol_hwd, D,
                           mccnn_hwd_pitch(D), N, out);
    return check_launch("mccnn_subpixel_hwd");
}
