// a3 compute_cross_region (/root/reference/src/process_functional.py:571-657) and
// a4 cost_volume_aggregation (pf:117-183) on gfx950.
//
// The reference materialises, per pixel, the coordinate list of its cross-shaped support region (int32
// [H,W,784,2], 2.35 GB per image at 750x500) and then gathers through it.  Here a pixel carries
// one packed 32-bit word (four 5-bit arm lengths + the 12-bit region size, mccnn_support_t) plus one derived 8-byte
// "emit word" (reciprocal of the size + vertical arms); the region is regenerated from the arms.
//
// Two aggregation kernels, same result set:
//   cbca_pipe_kernel    (MCCNN_CBCA_SEPARABLE, default distance) - O(1) work per output via float64 prefix sums,
//                       four wave-specialised stages stream a column strip of one disparity plane; 8 B/voxel/iteration
//                       of HBM traffic, bound by the latency of a pipeline step (TA, VALU and LDS all ~50-60 % busy).
//   cbca_iter_kernel    LDS-tiled; REFERENCE_ORDER variant walks the region in the reference's list order and is
//                       bit-exact; its separable variant serves distances > 14.
#include <math.h>
#include "common.h"
#ifndef PRIO_EMIT
#define PRIO_EMIT 3
#define PRIO_HSUM 2
#endif

namespace mccnn {

typedef uint32_t Support;  // == mccnn_support_t: bits 0-4 up, 5-9 down, 10-14 left, 15-19 right, 20-31 region size
static_assert(sizeof(mccnn_support_t) == 4, "support record must be 4 bytes");

__device__ __forceinline__ int arm_up(uint32_t a) { return (int)(a & 31u); }
__device__ __forceinline__ int arm_down(uint32_t a) { return (int)((a >> 5) & 31u); }
__device__ __forceinline__ int arm_left(uint32_t a) { return (int)((a >> 10) & 31u); }
__device__ __forceinline__ int arm_right(uint32_t a) { return (int)((a >> 15) & 31u); }
__device__ __forceinline__ int sup_count(uint32_t a) { return (int)(a >> 20); }

// np.linalg.norm of the 1-vector (cur - other): sqrt(x*x), float32 (pf:588,596,615,623)
__device__ __forceinline__ float norm1(float x)
{
    const float sq = x * x;
    return sqrtf(sq);
}

__global__ __launch_bounds__(256) void cross_arms_kernel(const float *__restrict__ img, int H, int W, float tau, int L,
                                                         Support *__restrict__ sup)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const float cur = img[(size_t)h * W + w];
    int up = 0, down = 0, left = 0, right = 0;
    // pf:585-591 / 612-618: bias 0 is the anchor itself (|0| < tau); stop at the first failure
    if (!(norm1(cur - cur) >= tau)) {
        int lim = min(L, h + 1);
        for (int b = 1; b < lim; ++b) {
            if (norm1(cur - img[(size_t)(h - b) * W + w]) >= tau) break;
            ++up;
        }
        lim = min(L, w + 1);
        for (int b = 1; b < lim; ++b) {
            if (norm1(cur - img[(size_t)h * W + (w - b)]) >= tau) break;
            ++left;
        }
    }
    {   // pf:593-599 / 620-626: bias 1 .. min(L, size - pos) - 1
        int lim = min(L, H - h);
        for (int b = 1; b < lim; ++b) {
            if (norm1(cur - img[(size_t)(h + b) * W + w]) >= tau) break;
            ++down;
        }
        lim = min(L, W - w);
        for (int b = 1; b < lim; ++b) {
            if (norm1(cur - img[(size_t)h * W + (w + b)]) >= tau) break;
            ++right;
        }
    }
    sup[(size_t)h * W + w] = (uint32_t)up | ((uint32_t)down << 5) | ((uint32_t)left << 10) | ((uint32_t)right << 15);
}

// pf:640-653: region size = sum over the vertical arm of the horizontal arm sizes.  The size goes into the upper 12
// bits of the same word whose lower 20 bits (the arms, written by the previous kernel and never changed here) the
// neighbours are reading: relaxed atomics make that formally race-free, and any mix of old/new words is correct.
//
// The same kernel fills the second plane of the support buffer, the 8-byte "emit words" the streaming aggregation
// kernel reads once per output: the float64 reciprocal 1/n rounded to 42 mantissa bits, whose 10 freed low bits
// carry the vertical arms (0-4 up, 5-9 down).  One 16-byte load per lane then brings everything the emit stage needs
// for two pixels, instead of a support word plus two table gathers (which cost ~55 cache-line lookups each).
__host__ __device__ __forceinline__ size_t emit_plane_offset(int H, int W)
{
    return ((size_t)H * W * 4 + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(256) void cross_count_kernel(Support *__restrict__ sup, int H, int W)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    auto ld = [&](size_t i) { return __hip_atomic_load(&sup[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const uint32_t a = ld((size_t)h * W + w);
    uint32_t n = 0;
    for (int q = h - arm_up(a); q <= h + arm_down(a); ++q) {
        const uint32_t aq = ld((size_t)q * W + w);
        n += arm_left(aq) + arm_right(aq) + 1;
    }
    __hip_atomic_fetch_or(&sup[(size_t)h * W + w], n << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long bits = (unsigned long long)__double_as_longlong(1.0 / (double)n);
    bits = (bits + 0x200ull) & ~0x3ffull;          // round to nearest at 42 mantissa bits
    unsigned long long *emitw =
        reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(sup) + emit_plane_offset(H, W));
    emitw[(size_t)h * W + w] = bits | (a & 0x3ffu);
}

// pf:637-655: explicit list, order (self, up.., down..) x (self, left.., right..), padded with (-1,-1)
__global__ __launch_bounds__(256) void cross_region_list_kernel(const Support *__restrict__ sup, int H, int W, int maxn,
                                                                int32_t *__restrict__ region)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const uint32_t a = sup[(size_t)h * W + w];
    int2 *r = reinterpret_cast<int2 *>(region) + ((size_t)h * W + w) * maxn;
    int n = 0;
    const int nu = arm_up(a), nv = 1 + nu + arm_down(a);
    for (int v = 0; v < nv; ++v) {
        const int q = v == 0 ? h : (v <= nu ? h - v : h + (v - nu));
        const uint32_t aq = sup[(size_t)q * W + w];
        const int nl = arm_left(aq), nh = 1 + nl + arm_right(aq);
        for (int z = 0; z < nh; ++z) {
            const int x = z == 0 ? w : (z <= nl ? w - z : w + (z - nl));
            r[n++] = make_int2(q, x);
        }
    }
    for (int i = n; i < maxn; ++i) r[i] = make_int2(-1, -1);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled aggregation: a workgroup stages a (TH+2R) x (TW+2R) tile of one disparity plane (R = halo = longest arm).
constexpr int CB_TW = 64;  // output tile width  (one wave = one full output row)

template <int R, int CB_TH, bool REF_ORDER>
__global__ __launch_bounds__(256) void cbca_iter_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const Support *__restrict__ sup, int H, int W)
{
    constexpr int IW = CB_TW + 2 * R;      // staged tile width
    constexpr int IH = CB_TH + 2 * R;      // staged tile height
    constexpr int IP = IW + 1;             // LDS pitch
    __shared__ float tin[IH * IP];
    __shared__ float ths[REF_ORDER ? 1 : IH * CB_TW];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * CB_TW, h0 = blockIdx.y * CB_TH;
    const size_t plane = (size_t)H * W;
    const float *src = in + (size_t)blockIdx.z * plane;
    float *dst = out + (size_t)blockIdx.z * plane;

    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int hh = h0 - R + r, ww = w0 - R + c;
        float v = 0.f;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = src[(size_t)hh * W + ww];
        tin[r * IP + c] = v;
    }
    __syncthreads();

    if constexpr (!REF_ORDER) {
        // horizontal-arm sums of every staged row (rows outside the image are never consumed)
        for (int i = tid; i < IH * CB_TW; i += 256) {
            const int r = i >> 6, c = i & 63;
            const int hh = h0 - R + r, ww = w0 + c;
            float s = 0.f;
            if (hh >= 0 && hh < H && ww < W) {
                const uint32_t a = sup[(size_t)hh * W + ww];
                const float *row = &tin[r * IP + c + R];
                for (int j = -arm_left(a); j <= arm_right(a); ++j) s += row[j];
            }
            ths[r * CB_TW + c] = s;
        }
        __syncthreads();
        const int c = tid & 63;
        const int ww = w0 + c;
        for (int k = 0; k < CB_TH / 4; ++k) {
            const int r = (tid >> 6) + 4 * k;
            const int hh = h0 + r;
            if (hh < H && ww < W) {
                const Support sp = sup[(size_t)hh * W + ww];
                const float *col = &ths[(r + R) * CB_TW + c];
                float s = 0.f;
                for (int i = -arm_up(sp); i <= arm_down(sp); ++i) s += col[i * CB_TW];
                dst[(size_t)hh * W + ww] = s / (float)sup_count(sp);  // pf:161
            }
        }
    } else {
        // the reference's flat running sum (pf:157-161), in its list order
        const int c = tid & 63;
        const int ww = w0 + c;
        for (int k = 0; k < CB_TH / 4; ++k) {
            const int r = (tid >> 6) + 4 * k;
            const int hh = h0 + r;
            if (hh < H && ww < W) {
                const Support sp = sup[(size_t)hh * W + ww];
                float s = 0.f;
                const int nu = arm_up(sp), nv = 1 + nu + arm_down(sp);
                for (int v = 0; v < nv; ++v) {
                    const int dq = v == 0 ? 0 : (v <= nu ? -v : v - nu);
                    const uint32_t aq = sup[(size_t)(hh + dq) * W + ww];
                    const float *row = &tin[(r + R + dq) * IP + c + R];
                    s += row[0];
                    for (int z = 1; z <= arm_left(aq); ++z) s += row[-z];
                    for (int z = 1; z <= arm_right(aq); ++z) s += row[z];
                }
                dst[(size_t)hh * W + ww] = s / (float)sup_count(sp);
            }
        }
    }
}

template <int R, int CB_TH>
static int launch_cbca(const float *in, float *out, const Support *sup, int D, int H, int W, int order, hipStream_t s)
{
    const dim3 grid(cdiv(W, CB_TW), cdiv(H, CB_TH), D), block(256);
    if (order == MCCNN_CBCA_REFERENCE_ORDER)
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, true>), grid, block, 0, s, in, out, sup, H, W);
    else
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, false>), grid, block, 0, s, in, out, sup, H, W);
    return check_launch("mccnn_cbca_iter");
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming separable aggregation: O(1) work per output, independent of the arm lengths (MCCNN_CBCA_SEPARABLE).
//
// The per-pixel loops of cbca_iter_kernel cost (wave-maximum arm length) dependent LDS round trips per output.  Here
// a strip of OUTW = 100 output columns of one disparity plane is streamed down its rows:
//   row y arrives (2 floats per lane, 128 columns = 100 outputs + 14/13-column halos)
//     -> float64 inclusive prefix sum P along the row (DPP scan across the 64 lanes)
//     -> horizontal-arm sum of pixel (y,c) = P[c+right] - P[c-left-1]              (2 LDS reads)
//     -> running float64 column prefix Q[y][c] += that, kept in an LDS ring
//   row y-13 leaves: vertical-arm sum = Q[y'+down] - Q[y'-up-1], times 1/|U|, rounded once to float32.
// float64 differences of prefix sums of float32 data carry ~1e-14 absolute error, so the result is the correctly
// rounded region mean; it differs from the reference's sequential float32 sum only by that sum's own rounding
// (same tolerance as any separable order).  Lanes never diverge.
//
// Wave specialisation.  A single wave per strip is bound by its own serial instruction chain (the float64 ring caps a
// CU at ~5 such strips; measured 0.53-0.80 ms per iteration at 750x500x256).  So FOUR waves share one strip (one
// ring) and each runs one stage of the row pipeline in lock step, one barrier per batch of B = 4 rows:
//     iteration t:   waves 0,3  scan   batch t     (2 rows each)  -> prow[t & 1]
//                    wave  1    hsum   batch t-1   prow lookups, column prefix -> ring rows
//                    wave  2    emit   batch t-2   ring lookups, x 1/|U|, store
// 16 waves per CU, and a strip advances at the pace of its slowest stage instead of the sum of all three (measured
// alone at 750x500x256: scan 0.14 ms, hsum 0.16 ms, emit 0.21 ms; a fifth wave sharing the emit rows, one scan + two
// emit waves, and 8-row batches were all slower).  Roles take s_setprio emit > hsum > scan: the emit wave is the one
// that never waits at the barrier.
//
// Memory side.  The vector-memory (TA) pipe is the busiest unit (a wave instruction keeps it ~16 cycles + bytes/64),
// so a row moves with four instructions: the 2 floats per lane (8 B), the packed support words of the staged row
// (8 B: left/right arms of 2 pixels), the emit words of the leaving row (16 B: float64 reciprocal of the region size
// with the vertical arms in its low mantissa bits, 2 pixels) and the 2-float store - all raw buffer ops: per-lane byte
// offset + wave-uniform row offset, no address arithmetic.  Each descriptor spans one whole plane and every access
// stays inside it by construction (the range check tests voffset + soffset against that span, per dword at the high
// end; tools/probe/bufrange.hip): pairs entirely outside the image are clamped onto valid columns - no arm can reach
// those elements, so any finite value cancels in the prefix differences - the one pair that can straddle the right
// edge (odd W) is fetched one column early and swizzled, and stores of rows outside the chunk are dropped by a
// per-lane offset beyond the span instead of a branch.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

constexpr int CS_IN = 128;  // staged columns per strip (2 per lane)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int R, int RING, bool ODDW>
__global__ __launch_bounds__(256) void cbca_pipe_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const Support *__restrict__ sup, int H, int W, int rows,
                                                        int nstrips, int nchunks, int total)
{
    constexpr int RS = (R + 1) & ~1;           // staged halo on the left, even so that lanes own aligned column pairs
    constexpr int OUTW = (CS_IN - RS - R) & ~1; // output columns per strip
    constexpr int RP = (OUTW + 3) & ~1;        // ring pitch in doubles (even, > OUTW)
    constexpr int B = 4;                       // rows per batch
    constexpr int NPF = 4;                     // batches of loads each role keeps in flight
    constexpr int PRP = CS_IN + 2;             // prow pitch: prow[k+1] = sum of staged elements 0..k, prow[0] = 0
    // emit(t-2) reads logical rows [y0-2R-1, y0+B-1] of the ring while hsum(t-1) writes the next B rows: 2R+1+2B rows
    // must not alias (a power-of-two ring would need 64 rows = 52 KB; 36 rows keep 4 workgroups per CU)
    static_assert(RING >= 2 * R + 1 + 2 * B, "ring must hold the emit window plus the batch being written");
    __shared__ double prow[2 * B * PRP];       // double-buffered by batch parity
    __shared__ double ring[RING * RP];
    const int lane = threadIdx.x & 63;
    // Role of this wave.  The dispatcher puts the four waves of a workgroup on the four SIMDs of a CU with a start
    // SIMD that varies per workgroup, so with roles tied to the wave index a SIMD can end up hosting several emit
    // waves (the longest stage) of the four resident workgroups.  Role = (SIMD id + wave slot) mod 4 gives every
    // SIMD one wave of each role when resident workgroups occupy equal slots; if the four waves do not come out
    // with four distinct roles (placement is not architecturally guaranteed), fall back to the wave index.
    __shared__ int claim[4];
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));   // [3:0] wave slot, [5:4] SIMD
        const int pref = (int)(((hw >> 4) + hw) & 3u);
        if (lane == 0) claim[wave] = pref;
        __syncthreads();
        const int seen = (1 << claim[0]) | (1 << claim[1]) | (1 << claim[2]) | (1 << claim[3]);
        if (seen == 15) wave = pref;
        wave = __builtin_amdgcn_readfirstlane(wave);
    }
    // XCD-aware order: consecutive work items (neighbouring strips of one plane share halo columns) stay on one
    // XCD's L2; the dispatcher places block b on XCD b % 8 (speed only, any placement is correct)
    int id;
    {
        const int b = blockIdx.x, q = total >> 3, r = total & 7, x = b & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int strip = id % nstrips;
    const int chunk = (id / nstrips) % nchunks;
    const int d = id / (nstrips * nchunks);
    const int w0 = strip * OUTW, h0 = chunk * rows, h1 = min(h0 + rows, H);
    const int ys = max(h0 - R, 0), ye = min(h1 - 1 + R, H - 1);
    const int ylast = h1 - 1 + R;
    const int nb = (ylast - ys + B) / B;       // batches; batch k holds rows ys + k*B + (0..B-1)
    const size_t plane = (size_t)H * W;

    const int x0 = w0 - RS + 2 * lane;         // image column of this lane's first staged element (even)
    const int c0 = w0 + 2 * lane;              // this lane's first output column (even)
    const bool oc0 = 2 * lane < OUTW && c0 < W, oc1 = 2 * lane + 1 < OUTW && c0 + 1 < W;
    const int i0 = oc0 ? RS + 2 * lane : RS;   // staged index of output column c0 (idle lanes stay in bounds)
    const bool rl = 2 * lane < RP;             // lane owns two ring columns
    const bool vstr = ODDW && x0 == W - 1, sstr = ODDW && c0 == W - 1;
    const int x0c = vstr ? W - 2 : min(max(x0, 0), W - 2);
    const int c0c = sstr ? W - 2 : min(c0, W - 2);
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in + (size_t)d * plane), 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dst =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)d * plane, 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sup =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<Support *>(sup), 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_emit = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(sup)) + emit_plane_offset(H, W), 0, (int)(plane * 8),
        0x00020000);
    const int vb = 4 * x0c, sb = 4 * c0c, ob = 4 * c0;
    const int rowv = 4 * W;

    if (threadIdx.x < 2 * B) prow[threadIdx.x * PRP] = 0.0;
    if (wave == 1 && rl) {                     // Q of the row above the first staged row is zero
        double *z = &ring[((ys - 1 + RING) % RING) * RP + lane];
        z[0] = 0.0;
        z[RP / 2] = 0.0;
    }

    if (wave == 0 || wave == 3) {
        // ---------------- scan role: rows b0, b0+1 of every batch ----------------
        const int b0 = wave == 0 ? 0 : 2;
        u32x2 vv[NPF][2];
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                vv[slot][j] = __builtin_amdgcn_raw_buffer_load_b64(rs_src, vb, min(ys + k * B + b0 + j, ye) * rowv, 0);
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        for (int tb = 0; tb < nb + 2; tb += NPF) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int t = tb + u;
                if (t >= nb + 2) break;
                if (t < nb) {
                    double a0[2], tt[2], ex[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        a0[j] = (double)__uint_as_float(vstr ? vv[u][j].y : vv[u][j].x);
                        tt[j] = a0[j] + (double)__uint_as_float(vv[u][j].y);
                    }
                    // steps outermost so the two rows' dependent chains interleave
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x111>(tt[j]);        // row_shr:1
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x112>(tt[j]);        // row_shr:2
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x114>(tt[j]);        // row_shr:4
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x118>(tt[j]);        // row_shr:8
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x142, 0xA>(tt[j]);   // row_bcast:15
#pragma unroll
                    for (int j = 0; j < 2; ++j) tt[j] += dpp_f64<0x143, 0xC>(tt[j]);   // row_bcast:31
#pragma unroll
                    for (int j = 0; j < 2; ++j) ex[j] = dpp_f64<0x138>(tt[j]);         // wave_shr:1 -> exclusive
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        double2 pp;
                        pp.x = ex[j] + a0[j];
                        pp.y = tt[j];
                        *reinterpret_cast<double2 *>(&prow[((t & 1) * B + b0 + j) * PRP + 1 + 2 * lane]) = pp;
                    }
                    issue(u, t + NPF);
                }
                __syncthreads();
            }
        }
    } else if (wave == 1) {
        // ---------------- hsum role: batch t-1 ----------------
        __builtin_amdgcn_s_setprio(PRIO_HSUM);
        double q0 = 0.0, q1 = 0.0;
        u32x2 sy[NPF][B];
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int b = 0; b < B; ++b)
                sy[slot][b] = __builtin_amdgcn_raw_buffer_load_b64(rs_sup, sb, min(ys + k * B + b, ye) * rowv, 0);
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        for (int tb = 0; tb < nb + 2; tb += NPF) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int t = tb + u;
                if (t >= nb + 2) break;
                const int k = t - 1;
                const int slot = (u + NPF - 1) % NPF;
                if (k >= 0 && k < nb) {
                    double hs0[B], hs1[B];
                    {
                        double pa0[B], pb0[B], pa1[B], pb1[B];   // all 16 prow reads in flight before the first use
                        const char *prb = reinterpret_cast<const char *>(prow) + (k & 1) * (B * PRP * 8);
#pragma unroll
                        for (int b = 0; b < B; ++b) {
                            const uint32_t a = sstr ? sy[slot][b].y : sy[slot][b].x, c = sy[slot][b].y;
                            const char *pr = prb + b * (PRP * 8) + 8 * i0;
                            // sum over staged elements [i-left, i+right] = prow[i+right+1] - prow[i-left]
                            pa0[b] = *reinterpret_cast<const double *>(pr + 8 * (arm_right(a) + 1));
                            pb0[b] = *reinterpret_cast<const double *>(pr - 8 * arm_left(a));
                            pa1[b] = *reinterpret_cast<const double *>(pr + 8 * (arm_right(c) + 2));
                            pb1[b] = *reinterpret_cast<const double *>(pr + 8 - 8 * arm_left(c));
                        }
#pragma unroll
                        for (int b = 0; b < B; ++b) {
                            hs0[b] = pa0[b] - pb0[b];
                            hs1[b] = pa1[b] - pb1[b];
                        }
                    }
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        // rows past the image bottom are staged from clamped loads and never referenced by an arm,
                        // so they run through unconditionally (straight-line code) into ring rows nobody reads
                        const int y = ys + k * B + b;
                        q0 += hs0[b];
                        q1 += hs1[b];
                        if (rl) {   // even column -> slot lane, odd column -> slot RP/2 + lane
                            double *rr = &ring[(y % RING) * RP + lane];
                            rr[0] = q0;
                            rr[RP / 2] = q1;
                        }
                    }
                    issue(slot, k + NPF);
                }
                __syncthreads();
            }
        }
    } else {
        // ---------------- emit role: batch t-2 ----------------
        __builtin_amdgcn_s_setprio(PRIO_EMIT);
        constexpr int EB = B;      // (splitting the rows over two emit waves, 320-thread workgroups, measured slower)
        constexpr int e0 = 0;
        constexpr int kDrop = 0x7ffffff0;          // byte offset past every plane: the range check drops the store
        const int obm = oc1 ? ob : kDrop;
        u32x4 so[NPF][EB];                         // emit words of the two output pixels: {lo0, hi0, lo1, hi1}
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int b = 0; b < EB; ++b)
                so[slot][b] = __builtin_amdgcn_raw_buffer_load_b128(
                    rs_emit, 2 * sb, min(max(ys + k * B + e0 + b - R, h0), h1 - 1) * (2 * rowv), 0);
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        for (int tb = 0; tb < nb + 2; tb += NPF) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int t = tb + u;
                if (t >= nb + 2) break;
                const int k = t - 2;
                const int slot = (u + NPF - 2) % NPF;
                if (k >= 0 && k < nb) {
                    // Straight-line for the whole batch: all ring reads are issued before the first is consumed
                    // (one LDS round trip per batch, not per row).  Rows outside the chunk compute on clamped
                    // indices and their store is dropped by an out-of-range buffer offset.
                    double qa0[EB], qb0[EB], qa1[EB], qb1[EB];
                    const char *ringb = reinterpret_cast<const char *>(ring);
                    const unsigned colb = 8u * (unsigned)lane;           // byte offset of this lane's even column in a ring row
#pragma unroll
                    for (int b = 0; b < EB; ++b) {
                        const int yoc = min(max(ys + k * B + e0 + b - R, h0), h1 - 1);
                        const uint32_t a = sstr ? so[slot][b].z : so[slot][b].x, c = so[slot][b].z;
                        const int ym = yoc % RING;   // wave-uniform; per-lane wrap by unsigned min
                        auto below = [&](int up) {
                            const int i = ym - up - 1;
                            return __umul24(min((unsigned)i, (unsigned)(i + RING)), RP * 8u);
                        };
                        auto above = [&](int dn) {
                            const int i = ym + dn;
                            return __umul24(min((unsigned)i, (unsigned)(i - RING)), RP * 8u);
                        };
                        qa0[b] = *reinterpret_cast<const double *>(ringb + above(arm_down(a)) + colb);
                        qb0[b] = *reinterpret_cast<const double *>(ringb + below(arm_up(a)) + colb);
                        qa1[b] = *reinterpret_cast<const double *>(ringb + above(arm_down(c)) + colb + 4 * RP);
                        qb1[b] = *reinterpret_cast<const double *>(ringb + below(arm_up(c)) + colb + 4 * RP);
                    }
#pragma unroll
                    for (int b = 0; b < EB; ++b) {
                        const int yo = ys + k * B + e0 + b - R;
                        const bool valid = yo >= h0 && yo < h1;      // wave-uniform
                        const int yoc = min(max(yo, h0), h1 - 1);
                        u32x2 o;
                        const u32x4 e = so[slot][b];
                        const double rn0 = __hiloint2double((int)(sstr ? e.w : e.y), (int)((sstr ? e.z : e.x) & ~0x3ffu));
                        const double rn1 = __hiloint2double((int)e.w, (int)(e.z & ~0x3ffu));
                        o.x = __float_as_uint((float)((qa0[b] - qb0[b]) * rn0));
                        o.y = __float_as_uint((float)((qa1[b] - qb1[b]) * rn1));
                        if (!ODDW) {
                            __builtin_amdgcn_raw_buffer_store_b64(o, rs_dst, valid ? obm : kDrop, yoc * rowv, 0);
                        } else if (valid) {
                            if (oc1)
                                __builtin_amdgcn_raw_buffer_store_b64(o, rs_dst, ob, yoc * rowv, 0);
                            else if (oc0)
                                __builtin_amdgcn_raw_buffer_store_b32(o.x, rs_dst, ob, yoc * rowv, 0);
                        }
                    }
                    issue(slot, k + NPF);
                }
                __syncthreads();
            }
        }
    }
}

template <int R, int RING>
static int launch_cbca_pipe(const float *in, float *out, const Support *sup, int D, int H, int W, hipStream_t s)
{
    constexpr int OUTW = (CS_IN - ((R + 1) & ~1) - R) & ~1;
    MCCNN_REQUIRE(W >= 2, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: W=%d < 2", W);
    const int nstrips = cdiv(W, OUTW);
    // Row chunks.  A strip of one plane can be cut into row chunks (each re-stages 2R halo rows).  The launch wants
    // (a) tall chunks and (b) a workgroup count that fills whole rounds of the chip's 4-per-CU resident slots: with
    // 1.0 < total/slots < 2.0 etc. the last round runs mostly empty (1242x375x192 in one chunk per strip: 2496
    // workgroups = 2.44 rounds, 19 % of the slot-time idle).  Pick the chunk count with the best product of the two
    // efficiencies; chunks are never shorter than 64 rows.  (750x500x256: 2048 workgroups = 2 full rounds in 1 chunk.)
    static const int slots = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        return 4 * (cus > 0 ? cus : 256);
    }();
    int nchunks = 1;
    double best = -1.0;
    for (int n = 1; n <= max(1, min(H / 64, 16)); ++n) {
        const double rounds = (double)nstrips * D * n / slots;
        const double fill = rounds / ceil(rounds);
        const double rows_n = (double)H / n;
        const double eff = fill * rows_n / (rows_n + 2 * R);
        if (eff > best + 1e-9) {
            best = eff;
            nchunks = n;
        }
    }
    const int rows = cdiv(H, nchunks);
    const long total = (long)nstrips * nchunks * D;
    MCCNN_REQUIRE(total <= 0x7fffffffL && (long)H * W * 4 <= 0x7fffffffL, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter: image %dx%d / volume too large for 32-bit buffer offsets", W, H);
    if (W & 1)
        hipLaunchKernelGGL((cbca_pipe_kernel<R, RING, true>), dim3((unsigned)total), dim3(256), 0, s, in, out, sup, H,
                           W, rows, nstrips, nchunks, (int)total);
    else
        hipLaunchKernelGGL((cbca_pipe_kernel<R, RING, false>), dim3((unsigned)total), dim3(256), 0, s, in, out, sup, H,
                           W, rows, nstrips, nchunks, (int)total);
    return check_launch("mccnn_cbca_iter(pipe)");
}

}  // namespace mccnn

extern "C" size_t mccnn_support_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    return mccnn::emit_plane_offset(H, W) + (size_t)H * W * 8;
}

extern "C" int mccnn_cross_arms(const float *image, int H, int W, float tau, int L, mccnn_support_t *support,
                                mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image && support, MCCNN_E_INVALID, "mccnn_cross_arms: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cross_arms: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 32, MCCNN_E_UNSUPPORTED,
                  "mccnn_cross_arms: L=%d outside [1,32] (5-bit arms, 12-bit region size)", L);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipLaunchKernelGGL(cross_arms_kernel, grid, block, 0, s, image, H, W, tau, L, support);
    int rc = check_launch("mccnn_cross_arms");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_count_kernel, grid, block, 0, s, support, H, W);
    return check_launch("mccnn_cross_arms(count)");
}

extern "C" int mccnn_cross_region_list(const mccnn_support_t *support, int H, int W, int L, int32_t *region,
                                       mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(support && region, MCCNN_E_INVALID, "mccnn_cross_region_list: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0 && L >= 1, MCCNN_E_INVALID, "mccnn_cross_region_list: bad size");
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipLaunchKernelGGL(cross_region_list_kernel, grid, block, 0, (hipStream_t)stream, support, H, W, (2 * L) * (2 * L),
                       region);
    return check_launch("mccnn_cross_region_list");
}

extern "C" int mccnn_cbca_iter(const float *in, float *out, const mccnn_support_t *support, int D, int H, int W, int L,
                               int order, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in && out && support, MCCNN_E_INVALID, "mccnn_cbca_iter: null pointer");
    MCCNN_REQUIRE(in != out, MCCNN_E_INVALID, "mccnn_cbca_iter: in-place aggregation is not defined (ping-pong)");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter: non-positive size");
    MCCNN_REQUIRE(D <= 65535, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: D=%d exceeds grid.z", D);
    MCCNN_REQUIRE(order == MCCNN_CBCA_SEPARABLE || order == MCCNN_CBCA_REFERENCE_ORDER, MCCNN_E_INVALID,
                  "mccnn_cbca_iter: unknown order %d", order);
    hipStream_t s = (hipStream_t)stream;
    if (order == MCCNN_CBCA_SEPARABLE && L <= 14) return launch_cbca_pipe<13, 36>(in, out, support, D, H, W, s);
    if (L <= 14) return launch_cbca<13, 32>(in, out, support, D, H, W, order, s);
    if (L <= 32) return launch_cbca<31, 16>(in, out, support, D, H, W, order, s);
    MCCNN_REQUIRE(false, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: L=%d > 32 not built", L);
}
