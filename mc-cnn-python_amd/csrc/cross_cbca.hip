// a3 compute_cross_region (/root/reference/src/process_functional.py:571-657) and
// a4 cost_volume_aggregation (pf:117-183) on gfx950.
//
// The reference materialises, per pixel, the coordinate list of its cross-shaped support region (int32
// [H,W,784,2], 2.35 GB per image at 750x500) and then gathers through it.  Here a pixel carries four uint8 arm
// lengths and the region size; the region is regenerated from them inside the aggregation kernel.
//
// cbca_iter: every disparity plane of the DHW volume is an independent H x W image, so a workgroup stages a
// (TH+2R) x (TW+2R) tile of one plane in LDS (R = L-1 = 13 halo), reduces horizontally into a second LDS tile and
// vertically from there: one HBM read + one HBM write per voxel and iteration (8 B), halo re-reads come from L2.
#include "common.h"

namespace mccnn {

// np.linalg.norm of the 1-vector (cur - other): sqrt(x*x), float32 (pf:588,596,615,623)
__device__ __forceinline__ float norm1(float x)
{
    const float sq = x * x;
    return sqrtf(sq);
}

__global__ __launch_bounds__(256) void cross_arms_kernel(const float *__restrict__ img, int H, int W, float tau, int L,
                                                         uint8_t *__restrict__ arms)
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
    uchar4 a;
    a.x = (unsigned char)up; a.y = (unsigned char)down; a.z = (unsigned char)left; a.w = (unsigned char)right;
    reinterpret_cast<uchar4 *>(arms)[(size_t)h * W + w] = a;
}

// pf:640-653: region size = sum over the vertical arm of the horizontal arm sizes
__global__ __launch_bounds__(256) void cross_count_kernel(const uint8_t *__restrict__ arms, int H, int W,
                                                          int32_t *__restrict__ count)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const uchar4 *A = reinterpret_cast<const uchar4 *>(arms);
    const uchar4 a = A[(size_t)h * W + w];
    int n = 0;
    for (int q = h - a.x; q <= h + a.y; ++q) {
        const uchar4 aq = A[(size_t)q * W + w];
        n += aq.z + aq.w + 1;
    }
    count[(size_t)h * W + w] = n;
}

// pf:637-655: explicit list, order (self, up.., down..) x (self, left.., right..), padded with (-1,-1)
__global__ __launch_bounds__(256) void cross_region_list_kernel(const uint8_t *__restrict__ arms, int H, int W, int maxn,
                                                                int32_t *__restrict__ region)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const uchar4 *A = reinterpret_cast<const uchar4 *>(arms);
    const uchar4 a = A[(size_t)h * W + w];
    int2 *r = reinterpret_cast<int2 *>(region) + ((size_t)h * W + w) * maxn;
    int n = 0;
    const int nv = 1 + a.x + a.y;
    for (int v = 0; v < nv; ++v) {
        const int q = v == 0 ? h : (v <= a.x ? h - v : h + (v - a.x));
        const uchar4 aq = A[(size_t)q * W + w];
        const int nh = 1 + aq.z + aq.w;
        for (int z = 0; z < nh; ++z) {
            const int x = z == 0 ? w : (z <= aq.z ? w - z : w + (z - aq.z));
            r[n++] = make_int2(q, x);
        }
    }
    for (int i = n; i < maxn; ++i) r[i] = make_int2(-1, -1);
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int CB_TW = 64;  // output tile width  (one wave = one full output row)

template <int R, int CB_TH, bool REF_ORDER>
__global__ __launch_bounds__(256) void cbca_iter_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const uint8_t *__restrict__ arms,
                                                        const int32_t *__restrict__ count, int H, int W)
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
    const uchar4 *A = reinterpret_cast<const uchar4 *>(arms);

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
                const uchar4 a = A[(size_t)hh * W + ww];
                const float *row = &tin[r * IP + c + R];
                for (int j = -(int)a.z; j <= (int)a.w; ++j) s += row[j];
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
                const uchar4 a = A[(size_t)hh * W + ww];
                const float *col = &ths[(r + R) * CB_TW + c];
                float s = 0.f;
                for (int i = -(int)a.x; i <= (int)a.y; ++i) s += col[i * CB_TW];
                dst[(size_t)hh * W + ww] = s / (float)count[(size_t)hh * W + ww];  // pf:161
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
                const uchar4 a = A[(size_t)hh * W + ww];
                float s = 0.f;
                const int nv = 1 + a.x + a.y;
                for (int v = 0; v < nv; ++v) {
                    const int dq = v == 0 ? 0 : (v <= a.x ? -v : v - a.x);
                    const uchar4 aq = A[(size_t)(hh + dq) * W + ww];
                    const float *row = &tin[(r + R + dq) * IP + c + R];
                    s += row[0];
                    for (int z = 1; z <= (int)aq.z; ++z) s += row[-z];
                    for (int z = 1; z <= (int)aq.w; ++z) s += row[z];
                }
                dst[(size_t)hh * W + ww] = s / (float)count[(size_t)hh * W + ww];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Streaming separable aggregation with O(1) work per output, independent of the arm lengths.
//
// The divergent per-pixel loops above cost (wave-maximum arm length) LDS round trips per output.  Here one
// wavefront owns a strip of CS_OUT output columns of one disparity plane and walks down the rows:
//   row y arrives (2 floats per lane, 128 columns incl. the 13-column halos)
//     -> float64 inclusive prefix sum P along the row (DPP scan across the 64 lanes)
//     -> horizontal-arm sum of pixel (y,c) = P[c+right] - P[c-left-1]              (2 LDS reads)
//     -> running float64 column prefix Q[y][c] += that, kept in a 32-row LDS ring
//   row y-13 leaves: vertical-arm sum = Q[y'+down] - Q[y'-up-1], divided by the region size (float32 divide).
// float64 differences of prefix sums of float32 data carry ~1e-14 absolute error, so the result is the correctly
// rounded region sum; it differs from the reference's sequential float32 sum only by that sum's own rounding
// (same tolerance as any separable order).  Lanes never diverge and no barrier is needed between waves: every
// workgroup is a single wavefront with a private LDS image (LDS operations of one wave execute in order).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes; *excl receives the sum of all lower lanes
__device__ __forceinline__ double wave_scan_f64(double t, double *excl)
{
    t += dpp_f64<0x111>(t);        // row_shr:1
    t += dpp_f64<0x112>(t);        // row_shr:2
    t += dpp_f64<0x114>(t);        // row_shr:4
    t += dpp_f64<0x118>(t);        // row_shr:8  -> inclusive scan inside each row of 16 lanes
    t += dpp_f64<0x142, 0xA>(t);   // row_bcast:15 -> rows 1 and 3 add the total of the row below
    t += dpp_f64<0x143, 0xC>(t);   // row_bcast:31 -> rows 2 and 3 add the total of lanes 0..31
    *excl = dpp_f64<0x138>(t);     // wave_shr:1 (lane 0 receives 0)
    return t;
}

constexpr int CS_IN = 128;  // staged columns per wave (2 per lane)

template <int R, int RING>
__global__ __launch_bounds__(64) void cbca_stream_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         const uint8_t *__restrict__ arms,
                                                         const int32_t *__restrict__ count, int H, int W, int rows,
                                                         int nstrips, int nchunks, int total)
{
    constexpr int OUTW = CS_IN - 2 * R;        // output columns per wave
    constexpr int RP = (OUTW + 3) & ~1;        // ring pitch in doubles (even, > OUTW)
    constexpr int B = 4;                       // rows advanced together (independent chains -> ILP, 1 sync per stage)
    constexpr int NB = 3;                      // batches of row registers: loads run 2 batches (8 rows) ahead
    constexpr int PRP = CS_IN + 2;             // prow pitch: prow[k+1] = sum of staged elements 0..k, prow[0] = 0
    static_assert(RING >= 2 * R + 2 + B && (RING & (RING - 1)) == 0, "ring must cover up+1+down rows of a batch");
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
    const float *src = in + (size_t)d * plane;
    float *dst = out + (size_t)d * plane;
    const uint32_t *A = reinterpret_cast<const uint32_t *>(arms);  // one packed uchar4 (up,down,left,right) per pixel

    const int x0 = w0 - R + 2 * lane;          // image column of this lane's first staged element
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const int c0 = w0 + 2 * lane;              // this lane's first output column
    const bool oc0 = 2 * lane < OUTW && c0 < W, oc1 = 2 * lane + 1 < OUTW && c0 + 1 < W;
    const int i0 = oc0 ? R + 2 * lane : R;     // staged index of output column c0 (idle lanes stay in bounds)
    const bool rl = 2 * lane < RP;             // lane owns two ring columns

    if (lane < B) prow[lane * PRP] = 0.0;
    if (rl) {                                  // Q of the row above the first staged row is zero
        double *z = &ring[((ys - 1) & (RING - 1)) * RP + 2 * lane];
        z[0] = 0.0;
        z[1] = 0.0;
    }
    double q0 = 0.0, q1 = 0.0;

    float v0[NB * B], v1[NB * B];
    uint32_t ay0[NB * B], ay1[NB * B];         // arms of the row being staged (left/right used)
    uint32_t ao0[NB * B], ao1[NB * B];         // arms of the row being emitted (up/down used)
    int n0[NB * B], n1[NB * B];
    // Branch-free loads: addresses are clamped into the image (values masked where they are consumed), so the
    // prefetch below is straight-line code and the compiler can count vmcnt instead of draining it.
    const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x0 + 1, 0), W - 1);
    const int c0c = min(c0, W - 1), c1c = min(c0 + 1, W - 1);
    auto issue = [&](int slot, int y) {
        const size_t rb = (size_t)min(y, ye) * W;
        v0[slot] = src[rb + x0c];
        v1[slot] = src[rb + x1c];
        ay0[slot] = A[rb + c0c];
        ay1[slot] = A[rb + c1c];
        const size_t ro = (size_t)min(max(y - R, h0), h1 - 1) * W;
        ao0[slot] = A[ro + c0c];
        ao1[slot] = A[ro + c1c];
        n0[slot] = count[ro + c0c];
        n1[slot] = count[ro + c1c];
    };
    const int ylast = h1 - 1 + R;
#pragma unroll
    for (int k = 0; k < NB * B; ++k) issue(k, ys + k);

    for (int yb = ys; yb <= ylast; yb += NB * B) {
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const int y0 = yb + g * B;
            if (y0 > ylast) break;
            // stage 1: B independent float64 row scans -> prow
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int k = g * B + b;
                if (y0 + b <= ye) {
                    const double a0 = xin0 ? (double)v0[k] : 0.0;
                    const double a1 = a0 + (xin1 ? (double)v1[k] : 0.0);
                    double ex;
                    wave_scan_f64(a1, &ex);
                    double2 pp;
                    pp.x = ex + a0;
                    pp.y = ex + a1;
                    *reinterpret_cast<double2 *>(&prow[b * PRP + 1 + 2 * lane]) = pp;
                }
            }
            __syncthreads();  // single-wave workgroup: orders the LDS writes above before the reads below
            // stage 2: horizontal-arm sums, running column prefix, ring rows
            double hs0[B], hs1[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int k = g * B + b;
                const uint32_t a = ay0[k], c = ay1[k];
                const double *pr = &prow[b * PRP];
                // sum over staged elements [i-left, i+right] = prow[i+right+1] - prow[i-left]
                hs0[b] = pr[i0 + (int)(a >> 24) + 1] - pr[i0 - (int)((a >> 16) & 0xff)];
                hs1[b] = pr[i0 + 1 + (int)(c >> 24) + 1] - pr[i0 + 1 - (int)((c >> 16) & 0xff)];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int y = y0 + b;
                if (y <= ye) {
                    q0 += oc0 ? hs0[b] : 0.0;
                    q1 += oc1 ? hs1[b] : 0.0;
                    if (rl) {
                        double2 qq;
                        qq.x = q0;
                        qq.y = q1;
                        *reinterpret_cast<double2 *>(&ring[(y & (RING - 1)) * RP + 2 * lane]) = qq;
                    }
                }
            }
            __syncthreads();
            // stage 3: emit rows y-R: vertical-arm sums from the ring, divide by the region size
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int k = g * B + b;
                const int yo = y0 + b - R;
                if (yo >= h0 && yo < h1) {
                    const uint32_t a = ao0[k], c = ao1[k];
                    const int col = 2 * lane;
                    if (oc0) {
                        const double s = ring[((yo + (int)((a >> 8) & 0xff)) & (RING - 1)) * RP + col] -
                                         ring[((yo - (int)(a & 0xff) - 1) & (RING - 1)) * RP + col];
                        dst[(size_t)yo * W + c0] = (float)s / (float)n0[k];
                    }
                    if (oc1) {
                        const double s = ring[((yo + (int)((c >> 8) & 0xff)) & (RING - 1)) * RP + col + 1] -
                                         ring[((yo - (int)(c & 0xff) - 1) & (RING - 1)) * RP + col + 1];
                        dst[(size_t)yo * W + c0 + 1] = (float)s / (float)n1[k];
                    }
                }
            }
            __syncthreads();  // the next batch overwrites prow and advances the ring
#pragma unroll
            for (int b = 0; b < B; ++b) issue(g * B + b, y0 + b + NB * B);
        }
    }
}

template <int R, int RING>
static int launch_cbca_stream(const float *in, float *out, const uint8_t *arms, const int32_t *count, int D, int H,
                              int W, hipStream_t s)
{
    constexpr int OUTW = CS_IN - 2 * R;
    const int nstrips = cdiv(W, OUTW);
    // row chunks of ~128 rows: each chunk re-reads 2R halo rows, so taller is cheaper; more chunks = more waves
    const int nchunks = H > 192 ? cdiv(H, 128) : 1;
    const int rows = cdiv(H, nchunks);
    const long total = (long)nstrips * nchunks * D;
    if (total > 0x7fffffffL) return -2;
    hipLaunchKernelGGL((cbca_stream_kernel<R, RING>), dim3((unsigned)total), dim3(64), 0, s, in, out, arms, count, H, W,
                       rows, nstrips, nchunks, (int)total);
    return check_launch("mccnn_cbca_iter(stream)");
}

template <int R, int CB_TH>
static int launch_cbca(const float *in, float *out, const uint8_t *arms, const int32_t *count, int D, int H, int W,
                       int order, hipStream_t s)
{
    const dim3 grid(cdiv(W, CB_TW), cdiv(H, CB_TH), D), block(256);
    if (order == MCCNN_CBCA_REFERENCE_ORDER)
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, true>), grid, block, 0, s, in, out, arms, count, H, W);
    else
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, false>), grid, block, 0, s, in, out, arms, count, H, W);
    return check_launch("mccnn_cbca_iter");
}

}  // namespace mccnn

extern "C" int mccnn_cross_arms(const float *image, int H, int W, float tau, int L, uint8_t *arms, int32_t *count,
                                mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image && arms && count, MCCNN_E_INVALID, "mccnn_cross_arms: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cross_arms: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 128, MCCNN_E_UNSUPPORTED, "mccnn_cross_arms: L=%d outside [1,128] (uint8 arms)", L);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipLaunchKernelGGL(cross_arms_kernel, grid, block, 0, s, image, H, W, tau, L, arms);
    int rc = check_launch("mccnn_cross_arms");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_count_kernel, grid, block, 0, s, arms, H, W, count);
    return check_launch("mccnn_cross_arms(count)");
}

extern "C" int mccnn_cross_region_list(const uint8_t *arms, int H, int W, int L, int32_t *region,
                                       mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(arms && region, MCCNN_E_INVALID, "mccnn_cross_region_list: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0 && L >= 1, MCCNN_E_INVALID, "mccnn_cross_region_list: bad size");
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipLaunchKernelGGL(cross_region_list_kernel, grid, block, 0, (hipStream_t)stream, arms, H, W, (2 * L) * (2 * L),
                       region);
    return check_launch("mccnn_cross_region_list");
}

extern "C" int mccnn_cbca_iter(const float *in, float *out, const uint8_t *arms, const int32_t *count, int D, int H,
                               int W, int L, int order, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in && out && arms && count, MCCNN_E_INVALID, "mccnn_cbca_iter: null pointer");
    MCCNN_REQUIRE(in != out, MCCNN_E_INVALID, "mccnn_cbca_iter: in-place aggregation is not defined (ping-pong)");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter: non-positive size");
    MCCNN_REQUIRE(D <= 65535, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: D=%d exceeds grid.z", D);
    MCCNN_REQUIRE(order == MCCNN_CBCA_SEPARABLE || order == MCCNN_CBCA_REFERENCE_ORDER, MCCNN_E_INVALID,
                  "mccnn_cbca_iter: unknown order %d", order);
    hipStream_t s = (hipStream_t)stream;
    if (order == MCCNN_CBCA_SEPARABLE) {
        // the streaming kernel is built for the default distance (arms <= 13); longer arms use the tile kernel
        if (L <= 14) return launch_cbca_stream<13, 32>(in, out, arms, count, D, H, W, s);
    }
    if (L <= 14) return launch_cbca<13, 32>(in, out, arms, count, D, H, W, order, s);
    if (L <= 32) return launch_cbca<31, 16>(in, out, arms, count, D, H, W, order, s);
    MCCNN_REQUIRE(false, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: L=%d > 32 not built", L);
}
