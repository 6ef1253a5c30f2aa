// a3 compute_cross_region (/root/reference/src/process_functional.py:571-657) and
// a4 cost_volume_aggregation (pf:117-183) on gfx950.
//
// The reference materialises, per pixel, the coordinate list of its cross-shaped support region (int32
// [H,W,784,2], 2.35 GB per image at 750x500) and then gathers through it.  Here a pixel carries one 8-byte record
// (four uint8 arm lengths + the int32 region size, mccnn_support_t); the region is regenerated from the arms.
//
// Two aggregation kernels, same result set:
//   cbca_stream_kernel  (MCCNN_CBCA_SEPARABLE, default distance) - O(1) work per output via float64 prefix sums,
//                       one wavefront streams a column strip of one disparity plane; bound by the vector-memory
//                       issue rate and HBM (8 B/voxel/iteration).
//   cbca_iter_kernel    LDS-tiled; REFERENCE_ORDER variant walks the region in the reference's list order and is
//                       bit-exact; its separable variant serves distances > 14.
#include "common.h"

namespace mccnn {

struct __attribute__((aligned(8))) Support {  // == mccnn_support_t
    uint32_t arms;  // byte 0 up, 1 down, 2 left, 3 right
    int32_t count;
};
static_assert(sizeof(Support) == 8 && sizeof(mccnn_support_t) == 8, "support record must be 8 bytes");

__device__ __forceinline__ int arm_up(uint32_t a) { return (int)(a & 0xff); }
__device__ __forceinline__ int arm_down(uint32_t a) { return (int)((a >> 8) & 0xff); }
__device__ __forceinline__ int arm_left(uint32_t a) { return (int)((a >> 16) & 0xff); }
__device__ __forceinline__ int arm_right(uint32_t a) { return (int)(a >> 24); }

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
    sup[(size_t)h * W + w].arms = (uint32_t)up | ((uint32_t)down << 8) | ((uint32_t)left << 16) | ((uint32_t)right << 24);
}

// pf:640-653: region size = sum over the vertical arm of the horizontal arm sizes
__global__ __launch_bounds__(256) void cross_count_kernel(Support *__restrict__ sup, int H, int W)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const uint32_t a = sup[(size_t)h * W + w].arms;
    int n = 0;
    for (int q = h - arm_up(a); q <= h + arm_down(a); ++q) {
        const uint32_t aq = sup[(size_t)q * W + w].arms;
        n += arm_left(aq) + arm_right(aq) + 1;
    }
    sup[(size_t)h * W + w].count = n;
}

// pf:637-655: explicit list, order (self, up.., down..) x (self, left.., right..), padded with (-1,-1)
__global__ __launch_bounds__(256) void cross_region_list_kernel(const Support *__restrict__ sup, int H, int W, int maxn,
                                                                int32_t *__restrict__ region)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const uint32_t a = sup[(size_t)h * W + w].arms;
    int2 *r = reinterpret_cast<int2 *>(region) + ((size_t)h * W + w) * maxn;
    int n = 0;
    const int nu = arm_up(a), nv = 1 + nu + arm_down(a);
    for (int v = 0; v < nv; ++v) {
        const int q = v == 0 ? h : (v <= nu ? h - v : h + (v - nu));
        const uint32_t aq = sup[(size_t)q * W + w].arms;
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
                const uint32_t a = sup[(size_t)hh * W + ww].arms;
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
                for (int i = -arm_up(sp.arms); i <= arm_down(sp.arms); ++i) s += col[i * CB_TW];
                dst[(size_t)hh * W + ww] = s / (float)sp.count;  // pf:161
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
                const int nu = arm_up(sp.arms), nv = 1 + nu + arm_down(sp.arms);
                for (int v = 0; v < nv; ++v) {
                    const int dq = v == 0 ? 0 : (v <= nu ? -v : v - nu);
                    const uint32_t aq = sup[(size_t)(hh + dq) * W + ww].arms;
                    const float *row = &tin[(r + R + dq) * IP + c + R];
                    s += row[0];
                    for (int z = 1; z <= arm_left(aq); ++z) s += row[-z];
                    for (int z = 1; z <= arm_right(aq); ++z) s += row[z];
                }
                dst[(size_t)hh * W + ww] = s / (float)sp.count;
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
// Streaming separable aggregation with O(1) work per output, independent of the arm lengths.
//
// The per-pixel loops above cost (wave-maximum arm length) dependent LDS round trips per output.  Here one
// wavefront owns a strip of OUTW output columns of one disparity plane and walks down the rows:
//   row y arrives (2 floats per lane, 128 columns = 100 outputs + 14/13-column halos)
//     -> float64 inclusive prefix sum P along the row (DPP scan across the 64 lanes)
//     -> horizontal-arm sum of pixel (y,c) = P[c+right] - P[c-left-1]              (2 LDS reads)
//     -> running float64 column prefix Q[y][c] += that, kept in a 32-row LDS ring
//   row y-13 leaves: vertical-arm sum = Q[y'+down] - Q[y'-up-1], times 1/|U|, rounded once to float32.
// float64 differences of prefix sums of float32 data carry ~1e-14 absolute error, so the result is the correctly
// rounded region mean; it differs from the reference's sequential float32 sum only by that sum's own rounding
// (same tolerance as any separable order).  Lanes never diverge and no barrier is needed between waves: every
// workgroup is a single wavefront with a private LDS image (LDS operations of one wave execute in order).
//
// Memory side: the vector-memory (TA) pipe costs ~16 cycles per wave instruction whatever its width, so each row
// moves with 4 instructions: one dwordx2 (2 floats), two dwordx4 (support records of the staged and of the emitted
// row, 2 pixels each), one dwordx2 store - all raw buffer ops: per-lane byte offset + wave-uniform row offset, no
// address arithmetic, hardware range check instead of clamping (out-of-image columns read finite neighbours or 0;
// no arm can reach them, so they cancel in the prefix differences).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// 1/n for the region sizes n <= (2*14)^2, rounded to float64 at compile time (the emit stage multiplies the float64
// region sum by it and rounds once to float32: the correctly rounded quotient up to ~1e-8 of near-ties)
struct InvTable {
    double v[800];
    constexpr InvTable() : v()
    {
        v[0] = 0.0;
        for (int i = 1; i < 800; ++i) v[i] = 1.0 / (double)i;
    }
};
__device__ const InvTable kInv = InvTable();

constexpr int CS_IN = 128;  // staged columns per wave (2 per lane)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int R, int RING>
__global__ __launch_bounds__(64) void cbca_stream_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         const Support *__restrict__ sup, int H, int W, int rows,
                                                         int nstrips, int nchunks, int total)
{
    constexpr int RS = (R + 1) & ~1;           // staged halo on the left, even so that lanes own aligned column pairs
    constexpr int OUTW = (CS_IN - RS - R) & ~1; // output columns per wave
    constexpr int RP = (OUTW + 3) & ~1;        // ring pitch in doubles (even, > OUTW)
    constexpr int B = 4;                       // rows advanced together (independent chains -> ILP, 1 sync per stage)
    constexpr int NB = 3;                      // batches of row registers: loads run 2 batches (8 rows) ahead
    constexpr int PRP = CS_IN + 2;             // prow pitch: prow[k+1] = sum of staged elements 0..k, prow[0] = 0
    static_assert(RING >= 2 * R + 2 + B && (RING & (RING - 1)) == 0, "ring must cover up+1+down rows of a batch");
    static_assert((2 * R + 2) * (2 * R + 2) <= 800, "reciprocal table too small");
    static_assert((OUTW & 1) == 0, "lanes own column pairs");
    __shared__ double prow[B * PRP];
    __shared__ double ring[RING * RP];
    const int lane = threadIdx.x;
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
    const size_t plane = (size_t)H * W;

    const int x0 = w0 - RS + 2 * lane;         // image column of this lane's first staged element (even)
    const int c0 = w0 + 2 * lane;              // this lane's first output column (even)
    const bool oc0 = 2 * lane < OUTW && c0 < W, oc1 = 2 * lane + 1 < OUTW && c0 + 1 < W;
    const int i0 = oc0 ? RS + 2 * lane : RS;   // staged index of output column c0 (idle lanes stay in bounds)
    const bool rl = 2 * lane < RP;             // lane owns two ring columns
    // Every access stays inside its plane by construction (the buffer range check does not cover the scalar row
    // offset): pairs entirely outside the image are clamped onto valid columns - no arm can reach those elements, so
    // any finite value cancels in the prefix differences - and the one pair that can straddle the right edge (odd W)
    // is fetched one column early and swizzled.
    const bool vstr = x0 == W - 1, sstr = c0 == W - 1;
    const int x0c = vstr ? W - 2 : min(max(x0, 0), W - 2);
    const int c0c = sstr ? W - 2 : min(c0, W - 2);

    // buffer descriptors (wave-uniform); voffset = per-lane byte offset, soffset = wave-uniform row offset
    const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(in + (size_t)d * plane), 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dst =
        __builtin_amdgcn_make_buffer_rsrc(out + (size_t)d * plane, 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sup =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<Support *>(sup), 0, (int)(plane * 8), 0x00020000);
    const int vb = 4 * x0c, sb = 8 * c0c, ob = 4 * c0;
    const int rowv = 4 * W, rows8 = 8 * W;

    if (lane < B) prow[lane * PRP] = 0.0;
    if (rl) {                                  // Q of the row above the first staged row is zero
        double *z = &ring[((ys - 1) & (RING - 1)) * RP + 2 * lane];
        z[0] = 0.0;
        z[1] = 0.0;
    }
    double q0 = 0.0, q1 = 0.0;

    u32x2 vv[NB * B];                          // two staged floats
    u32x4 sy[NB * B];                          // support records of the staged row (left/right used)
    u32x4 so[NB * B];                          // support records of the emitted row (up/down/count used)
    auto issue = [&](int slot, int y) {
        const int yr = min(y, ye);                          // wave-uniform
        const int yo = min(max(y - R, h0), h1 - 1);         // wave-uniform
        vv[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs_src, vb, yr * rowv, 0);
        sy[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_sup, sb, yr * rows8, 0);
        so[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_sup, sb, yo * rows8, 0);
    };
    const int ylast = h1 - 1 + R;
#pragma unroll
    for (int k = 0; k < NB * B; ++k) issue(k, ys + k);

    for (int yb = ys; yb <= ylast; yb += NB * B) {
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const int y0 = yb + g * B;
            if (y0 > ylast) break;
            // reciprocal region sizes of the rows emitted by this batch (consumed two LDS stages from now)
            double rn0[B], rn1[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                rn0[b] = kInv.v[sstr ? so[g * B + b].w : so[g * B + b].y];
                rn1[b] = kInv.v[so[g * B + b].w];
            }
            // stage 1: B independent float64 row scans -> prow (steps outermost so the B chains interleave)
            {
                double a0[B], t[B], ex[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    a0[b] = (double)__uint_as_float(vstr ? vv[g * B + b].y : vv[g * B + b].x);
                    t[b] = a0[b] + (double)__uint_as_float(vv[g * B + b].y);
                }
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x111>(t[b]);        // row_shr:1
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x112>(t[b]);        // row_shr:2
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x114>(t[b]);        // row_shr:4
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x118>(t[b]);        // row_shr:8
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x142, 0xA>(t[b]);   // row_bcast:15
#pragma unroll
                for (int b = 0; b < B; ++b) t[b] += dpp_f64<0x143, 0xC>(t[b]);   // row_bcast:31
#pragma unroll
                for (int b = 0; b < B; ++b) ex[b] = dpp_f64<0x138>(t[b]);        // wave_shr:1 -> exclusive
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    double2 pp;
                    pp.x = ex[b] + a0[b];
                    pp.y = t[b];
                    *reinterpret_cast<double2 *>(&prow[b * PRP + 1 + 2 * lane]) = pp;
                }
            }
            __syncthreads();  // single-wave workgroup: orders the LDS writes above before the reads below
            // stage 2: horizontal-arm sums, running column prefix, ring rows
            double hs0[B], hs1[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint32_t a = sstr ? sy[g * B + b].z : sy[g * B + b].x, c = sy[g * B + b].z;
                const double *pr = &prow[b * PRP];
                // sum over staged elements [i-left, i+right] = prow[i+right+1] - prow[i-left]
                hs0[b] = pr[i0 + arm_right(a) + 1] - pr[i0 - arm_left(a)];
                hs1[b] = pr[i0 + 1 + arm_right(c) + 1] - pr[i0 + 1 - arm_left(c)];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int y = y0 + b;
                if (y <= ye) {                 // rows past the image bottom are never referenced
                    q0 += hs0[b];
                    q1 += hs1[b];
                    if (rl) {
                        double2 qq;
                        qq.x = q0;
                        qq.y = q1;
                        *reinterpret_cast<double2 *>(&ring[(y & (RING - 1)) * RP + 2 * lane]) = qq;
                    }
                }
            }
            __syncthreads();
            // stage 3: emit rows y-R: vertical-arm sums from the ring, times 1/region size, one rounding to float32
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int yo = y0 + b - R;
                if (yo >= h0 && yo < h1) {
                    const uint32_t a = sstr ? so[g * B + b].z : so[g * B + b].x, c = so[g * B + b].z;
                    const int col = 2 * lane;
                    const double s0 = ring[((yo + arm_down(a)) & (RING - 1)) * RP + col] -
                                      ring[((yo - arm_up(a) - 1) & (RING - 1)) * RP + col];
                    const double s1 = ring[((yo + arm_down(c)) & (RING - 1)) * RP + col + 1] -
                                      ring[((yo - arm_up(c) - 1) & (RING - 1)) * RP + col + 1];
                    u32x2 o;
                    o.x = __float_as_uint((float)(s0 * rn0[b]));
                    o.y = __float_as_uint((float)(s1 * rn1[b]));
                    if (oc1)
                        __builtin_amdgcn_raw_buffer_store_b64(o, rs_dst, ob, yo * rowv, 0);
                    else if (oc0)
                        __builtin_amdgcn_raw_buffer_store_b32(o.x, rs_dst, ob, yo * rowv, 0);
                }
            }
            __syncthreads();  // the next batch overwrites prow and advances the ring
#pragma unroll
            for (int b = 0; b < B; ++b) issue(g * B + b, y0 + b + NB * B);
        }
    }
}

template <int R, int RING>
static int launch_cbca_stream(const float *in, float *out, const Support *sup, int D, int H, int W, hipStream_t s)
{
    constexpr int OUTW = (CS_IN - ((R + 1) & ~1) - R) & ~1;
    MCCNN_REQUIRE(W >= 2, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: W=%d < 2", W);
    const int nstrips = cdiv(W, OUTW);
    // row chunks of ~128 rows: each chunk re-reads 2R halo rows, so taller is cheaper; more chunks = more waves
    const int nchunks = H > 192 ? cdiv(H, 128) : 1;
    const int rows = cdiv(H, nchunks);
    const long total = (long)nstrips * nchunks * D;
    MCCNN_REQUIRE(total <= 0x7fffffffL && (long)H * W * 8 <= 0x7fffffffL, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter: image %dx%d / volume too large for 32-bit buffer offsets", W, H);
    hipLaunchKernelGGL((cbca_stream_kernel<R, RING>), dim3((unsigned)total), dim3(64), 0, s, in, out, sup, H, W, rows,
                       nstrips, nchunks, (int)total);
    return check_launch("mccnn_cbca_iter(stream)");
}

}  // namespace mccnn

extern "C" int mccnn_cross_arms(const float *image, int H, int W, float tau, int L, mccnn_support_t *support,
                                mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image && support, MCCNN_E_INVALID, "mccnn_cross_arms: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cross_arms: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 128, MCCNN_E_UNSUPPORTED, "mccnn_cross_arms: L=%d outside [1,128] (uint8 arms)", L);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(W, 256), H), block(256);
    Support *sup = reinterpret_cast<Support *>(support);
    hipLaunchKernelGGL(cross_arms_kernel, grid, block, 0, s, image, H, W, tau, L, sup);
    int rc = check_launch("mccnn_cross_arms");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_count_kernel, grid, block, 0, s, sup, H, W);
    return check_launch("mccnn_cross_arms(count)");
}

extern "C" int mccnn_cross_region_list(const mccnn_support_t *support, int H, int W, int L, int32_t *region,
                                       mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(support && region, MCCNN_E_INVALID, "mccnn_cross_region_list: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0 && L >= 1, MCCNN_E_INVALID, "mccnn_cross_region_list: bad size");
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipLaunchKernelGGL(cross_region_list_kernel, grid, block, 0, (hipStream_t)stream,
                       reinterpret_cast<const Support *>(support), H, W, (2 * L) * (2 * L), region);
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
    const Support *sup = reinterpret_cast<const Support *>(support);
    if (order == MCCNN_CBCA_SEPARABLE && L <= 14) return launch_cbca_stream<13, 32>(in, out, sup, D, H, W, s);
    if (L <= 14) return launch_cbca<13, 32>(in, out, sup, D, H, W, order, s);
    if (L <= 32) return launch_cbca<31, 16>(in, out, sup, D, H, W, order, s);
    MCCNN_REQUIRE(false, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: L=%d > 32 not built", L);
}
