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
    if (L <= 14) return launch_cbca<13, 32>(in, out, arms, count, D, H, W, order, s);
    if (L <= 32) return launch_cbca<31, 16>(in, out, arms, count, D, H, W, order, s);
    MCCNN_REQUIRE(false, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: L=%d > 32 not built", L);
}
