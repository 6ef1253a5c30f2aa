// a7-a11 of the hot path (/root/reference/src/process_functional.py:239-470) and the feature-head epilogue
// (model.py:64) on gfx950.  All of these touch O(H*W) data except WTA / sub-pixel, which stream the DHW volume.
#include "common.h"

namespace mccnn {

// ---- a7 disparity_prediction (pf:245-254): first strict minimum -------------------------------------------------
// Adjacent lanes own adjacent columns, so every plane row is read in 256-B runs; 8 B/voxel for both volumes.
__global__ __launch_bounds__(256) void wta_kernel(const float *__restrict__ vol, int D, long N,
                                                  float *__restrict__ disp)
{
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float best = __builtin_huge_valf();
    int bd = -1;
    const float *p = vol + n;
    int d = 0;
    for (; d + 4 <= D; d += 4) {
        const float v0 = p[(size_t)(d + 0) * N], v1 = p[(size_t)(d + 1) * N], v2 = p[(size_t)(d + 2) * N],
                    v3 = p[(size_t)(d + 3) * N];
        if (v0 < best) { best = v0; bd = d; }
        if (v1 < best) { best = v1; bd = d + 1; }
        if (v2 < best) { best = v2; bd = d + 2; }
        if (v3 < best) { best = v3; bd = d + 3; }
    }
    for (; d < D; ++d) {
        const float v = p[(size_t)d * N];
        if (v < best) { best = v; bd = d; }
    }
    disp[n] = (float)bd;  // pf:254 stores the index into a float32 map (-1 only if every cost is NaN/+inf)
}

// ---- a8 interpolation (pf:279-378) ------------------------------------------------------------------------------
// One workgroup per image row.  pf:299-303 asks, for a pixel w whose own disparity failed the check, whether ANY d in
// [0, min(w+1, D)) has |d - dr[w-d]| <= 1 - a walk over up to D right-map pixels per pixel.  Turned around: a right-map
// pixel x with value r can satisfy that only for d within a few integers of r, i.e. for w = x + d in a handful of
// places; every thread evaluates the reference's own float32 expression for those candidates of its x and sets the
// bits of the w it hits in a row mask in LDS.  Same predicate, same arithmetic, O(1) per pixel.
constexpr int LR_MAX_WORDS = 512;   // rows up to 16384 pixels; wider rows take lr_status_walk_kernel

__global__ __launch_bounds__(256) void lr_status_kernel(const float *__restrict__ dl, const float *__restrict__ dr,
                                                        int H, int W, int D, int32_t *__restrict__ status)
{
    __shared__ uint32_t hit[LR_MAX_WORDS];
    const int h = blockIdx.x, tid = threadIdx.x;
    const float *drow = dr + (size_t)h * W;
    const int nwords = (W + 31) >> 5;
    for (int i = tid; i < nwords; i += 256) hit[i] = 0u;
    __syncthreads();
    for (int x = tid; x < W; x += 256) {
        const float r = drow[x];
        if (!(fabsf(r) < 1.0e6f)) continue;          // NaN / inf / absurd values match no d < D
        const int f = (int)floorf(r);
#pragma unroll
        for (int k = -2; k <= 3; ++k) {              // |d - r| <= 1 in float32 implies d within [floor(r) - 2, floor(r) + 3]
            const int d = f + k, w = x + d;
            if (d >= 0 && d < D && w < W && fabsf((float)d - r) <= 1.f) atomicOr(&hit[w >> 5], 1u << (w & 31));
        }
    }
    __syncthreads();
    for (int w = tid; w < W; w += 256) {
        const float lf = dl[(size_t)h * W + w];
        const int ld = (int)lf;  // pf:287 int() truncation
        int st;
        if (!(lf >= 0.f) || w < ld) {
            // pf:289-291.  A negative or NaN disparity cannot come out of the reference (its WTA asserts a finite
            // minimum, pf:253); mccnn_wta writes -1 for a pixel whose costs are all NaN/+inf, and such a pixel is treated
            // as occluded here instead of indexing the right map out of bounds.
            st = 2;
        } else if (fabsf((float)ld - drow[w - ld]) <= 1.f) {
            st = 0;  // pf:294
        } else {
            st = (hit[w >> 5] >> (w & 31)) & 1u ? 1 : 2;   // pf:299-303
        }
        status[(size_t)h * W + w] = st;
    }
}

__global__ __launch_bounds__(256) void lr_status_walk_kernel(const float *__restrict__ dl, const float *__restrict__ dr,
                                                             int H, int W, int D, int32_t *__restrict__ status)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const float *drow = dr + (size_t)h * W;
    const float lf = dl[(size_t)h * W + w];
    const int ld = (int)lf;
    int st;
    if (!(lf >= 0.f) || w < ld) {
        st = 2;
    } else if (fabsf((float)ld - drow[w - ld]) <= 1.f) {
        st = 0;
    } else {
        st = 2;
        const int lim = min(w + 1, D);
        for (int d = 0; d < lim; ++d)
            if (fabsf((float)d - drow[w - d]) <= 1.f) { st = 1; break; }
    }
    status[(size_t)h * W + w] = st;
}

__device__ __forceinline__ float median_upto4(float *v, int n)
{
    // np.median of 1..n float32 values (n <= 4 in the reference's rule, <= 16 in the paper's): sort, middle element or
    // mean of the two middle ones
    for (int i = 1; i < n; ++i) {
        const float x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
        v[j + 1] = x;
    }
    if (n & 1) return v[n >> 1];
    return (v[(n >> 1) - 1] + v[n >> 1]) / 2.f;
}

// The two rules the reference names but leaves out (pf:318 "in origin paper, they use 16 directions", pf:361 "they use
// left"), as in the MC-CNN paper, sec. 4.4: a mismatch takes the median of the nearest matches along 16 rays (steps of
// (dx, dy) in {0, +-0.5, +-1}^2 on the unit square's boundary, positions rounded half up), an occlusion takes the
// nearest match to its LEFT.  Opt-in: they change the output.
__constant__ float kRay16[16][2] = {{1, 0}, {1, 0.5f}, {1, 1}, {0.5f, 1}, {0, 1}, {-0.5f, 1}, {-1, 1}, {-1, 0.5f},
                                    {-1, 0}, {-1, -0.5f}, {-1, -1}, {-0.5f, -1}, {0, -1}, {0.5f, -1}, {1, -1}, {1, -0.5f}};

template <bool RAYS16, bool OCC_LEFT>
__global__ __launch_bounds__(256) void interpolate_paper_kernel(const float *__restrict__ dl,
                                                                const int32_t *__restrict__ st, int H, int W,
                                                                float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const size_t p = (size_t)h * W + w;
    const int s = st[p];
    float res = dl[p];
    if (s == 1) {
        float nb[16];
        int c = 0;
        if (RAYS16) {
            for (int r = 0; r < 16; ++r) {
                const float dx = kRay16[r][0], dy = kRay16[r][1];
                float xx = (float)w, yy = (float)h;
                for (;;) {
                    xx += dx;
                    yy += dy;
                    const int xi = (int)floorf(xx + 0.5f), yi = (int)floorf(yy + 0.5f);
                    if (xi < 0 || xi >= W || yi < 0 || yi >= H) break;
                    if (st[(size_t)yi * W + xi] == 0) { nb[c++] = dl[(size_t)yi * W + xi]; break; }
                }
            }
        } else {   // the reference's four axis directions, in its order (pf:321-347)
            for (int x = w + 1; x < W; ++x) if (st[(size_t)h * W + x] == 0) { nb[c++] = dl[(size_t)h * W + x]; break; }
            for (int x = w - 1; x >= 0; --x) if (st[(size_t)h * W + x] == 0) { nb[c++] = dl[(size_t)h * W + x]; break; }
            for (int y = h + 1; y < H; ++y) if (st[(size_t)y * W + w] == 0) { nb[c++] = dl[(size_t)y * W + w]; break; }
            for (int y = h - 1; y >= 0; --y) if (st[(size_t)y * W + w] == 0) { nb[c++] = dl[(size_t)y * W + w]; break; }
        }
        if (c > 0) res = median_upto4(nb, c);
    } else if (s == 2) {
        if (OCC_LEFT) {
            for (int x = w - 1; x >= 0; --x) if (st[(size_t)h * W + x] == 0) { res = dl[(size_t)h * W + x]; break; }
        } else {
            for (int x = w + 1; x < W; ++x) if (st[(size_t)h * W + x] == 0) { res = dl[(size_t)h * W + x]; break; }
        }
    }
    out[p] = res;
}

__global__ __launch_bounds__(256) void interpolate_kernel(const float *__restrict__ dl,
                                                          const int32_t *__restrict__ st, int H, int W,
                                                          float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const size_t p = (size_t)h * W + w;
    const int s = st[p];
    float res = dl[p];
    if (s == 1) {  // pf:316-356: nearest match right, left, below, above (reads the raw map only)
        float nb[4];
        int c = 0;
        for (int x = w + 1; x < W; ++x) if (st[(size_t)h * W + x] == 0) { nb[c++] = dl[(size_t)h * W + x]; break; }
        for (int x = w - 1; x >= 0; --x) if (st[(size_t)h * W + x] == 0) { nb[c++] = dl[(size_t)h * W + x]; break; }
        for (int y = h + 1; y < H; ++y) if (st[(size_t)y * W + w] == 0) { nb[c++] = dl[(size_t)y * W + w]; break; }
        for (int y = h - 1; y >= 0; --y) if (st[(size_t)y * W + w] == 0) { nb[c++] = dl[(size_t)y * W + w]; break; }
        if (c > 0) res = median_upto4(nb, c);
    } else if (s == 2) {  // pf:358-373: nearest match to the right
        for (int x = w + 1; x < W; ++x) if (st[(size_t)h * W + x] == 0) { res = dl[(size_t)h * W + x]; break; }
    }
    out[p] = res;
}

// The same rule without per-pixel walks (the walks cost the longest run of unmatched pixels in the wave: 0.11 ms at
// 750x500 when most of the map is unmatched).  Pass A: one lane per image column runs down and up its column carrying
// the row of the last match (the loads do not depend on the carry, so they pipeline) and leaves "nearest match above /
// below" as two 16-bit row indices in the pixel's own slot of the output map.  Pass B: one workgroup per image row
// ballots the row's matches into a bit mask in LDS; "nearest match right / left" is then a masked find-first-set over
// at most W/64 words.  Same neighbours, same order, same median: bit-identical to interpolate_kernel.
__global__ __launch_bounds__(64) void interpolate_vertical_kernel(const int32_t *__restrict__ st, int H, int W,
                                                                  uint16_t *__restrict__ slots)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= W) return;
    uint32_t last = 0xffffu;
    if (blockIdx.y == 0) {   // half-word 0: nearest match above
#pragma unroll 32
        for (int y = 0; y < H; ++y) {
            const size_t p = (size_t)y * W + w;
            const int s = st[p];
            slots[2 * p] = (uint16_t)last;
            if (s == 0) last = (uint32_t)y;
        }
    } else {                 // half-word 1: nearest match below
#pragma unroll 32
        for (int y = H - 1; y >= 0; --y) {
            const size_t p = (size_t)y * W + w;
            const int s = st[p];
            slots[2 * p + 1] = (uint16_t)last;
            if (s == 0) last = (uint32_t)y;
        }
    }
}

constexpr int INTERP_MAX_WORDS = 256;   // rows up to 16384 pixels

__global__ __launch_bounds__(256) void interpolate_row_kernel(const float *__restrict__ dl,
                                                              const int32_t *__restrict__ st, int H, int W,
                                                              float *__restrict__ out)
{
    __shared__ unsigned long long mask[INTERP_MAX_WORDS];
    const int h = blockIdx.x, tid = threadIdx.x;
    const size_t row = (size_t)h * W;
    const int nwords = (W + 63) >> 6;
    for (int base = 0; base < W; base += 256) {
        const int w = base + tid;
        const bool matched = w < W && st[row + w] == 0;
        const unsigned long long m = __ballot(matched);
        if ((tid & 63) == 0 && ((base + tid) >> 6) < nwords) mask[(base + tid) >> 6] = m;
    }
    __syncthreads();
    const uint32_t *bits = reinterpret_cast<const uint32_t *>(out);
    for (int base = 0; base < W; base += 256) {
        const int w = base + tid;
        if (w >= W) continue;
        const size_t p = row + w;
        const int s = st[p];
        float res = dl[p];
        const uint32_t vert = bits[p];
        if (s != 0) {
            int xr = -1;   // nearest match to the right
            if (w + 1 < W) {
                int i = (w + 1) >> 6;
                unsigned long long m = mask[i] & (~0ull << ((w + 1) & 63));
                while (m == 0 && ++i < nwords) m = mask[i];
                if (m) xr = i * 64 + __ffsll((long long)m) - 1;
            }
            if (s == 1) {  // pf:316-356: nearest match right, left, below, above (reads the raw map only)
                float nb[4];
                int c = 0;
                if (xr >= 0) nb[c++] = dl[row + xr];
                if (w >= 1) {
                    int i = (w - 1) >> 6;
                    unsigned long long m = mask[i] & (~0ull >> (63 - ((w - 1) & 63)));
                    while (m == 0 && --i >= 0) m = mask[i];
                    if (m) nb[c++] = dl[row + i * 64 + 63 - __clzll((long long)m)];
                }
                const uint32_t below = vert >> 16, above = vert & 0xffffu;
                if (below != 0xffffu) nb[c++] = dl[(size_t)below * W + w];
                if (above != 0xffffu) nb[c++] = dl[(size_t)above * W + w];
                if (c > 0) res = median_upto4(nb, c);
            } else if (xr >= 0) {  // pf:358-373: nearest match to the right
                res = dl[row + xr];
            }
        }
        out[p] = res;
    }
}

// ---- a9 subpixel_enhance (pf:387-396) ---------------------------------------------------------------------------
// NUMPY1: the scalar promotion of NumPy < 2, which the reference's own Python 2.7 + NumPy 1.14 environment applies to
// pf:396: `C_p - C_m` stays float32 (two float32 scalars), but `2. * C` pairs a float32 scalar with a Python float and
// becomes float64, and so does everything downstream - the quotient and the subtraction run in float64 and the result
// is rounded to float32 once, on the store.  The default is NumPy >= 2's all-float32 chain, which is what the golden
// vectors (generated by running the reference under NumPy 2.2) pin; the two differ by <= 2.5e-5 px.
template <bool NUMPY1>
__global__ __launch_bounds__(256) void subpixel_kernel(const float *__restrict__ dl, const float *__restrict__ vol,
                                                       int D, long N, float *__restrict__ out)
{
    const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float d = dl[n];
    const int im = (int)(d - 1.f), ip = (int)(d + 1.f), ic = (int)d;
    float res = d;
    if (!(im < 0 || ip >= D)) {
        const float cm = vol[(size_t)im * N + n], cp = vol[(size_t)ip * N + n], c = vol[(size_t)ic * N + n];
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

// ---- a10 median_filter (pf:409-417) -----------------------------------------------------------------------------
constexpr int MAXWIN = 49;

__global__ __launch_bounds__(256) void median_kernel(const float *__restrict__ dl, int H, int W, int fh, int fw,
                                                     float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const int rh = (fh - 1) / 2, rw = (fw - 1) / 2;
    const int hs = max(0, h - rh), he = min(H, h + rh + 1), ws = max(0, w - rw), we = min(W, w + rw + 1);
    float v[MAXWIN];
    int n = 0;
    bool has_nan = false;
    for (int y = hs; y < he; ++y)
        for (int x = ws; x < we; ++x) {
            const float t = dl[(size_t)y * W + x];
            has_nan |= (t != t);
            // insertion keeps v[0..n) sorted ascending
            int j = n - 1;
            while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
            v[j + 1] = t;
            ++n;
        }
    float res;
    if (has_nan) res = __builtin_nanf("");  // np.median propagates NaN
    else if (n & 1) res = v[n >> 1];
    else res = (v[(n >> 1) - 1] + v[n >> 1]) / 2.f;  // np.mean of the two middle float32 values
    out[(size_t)h * W + w] = res;
}

// 5x5 window on images of at least 5x5 pixels (the only size match.py uses, match.py:172): the 25 taps live in
// registers (out-of-window taps = +inf, which sorts behind every real value), an odd-even transposition network sorts
// them with compile-time indices (no scratch memory), and the clipped window size n in {9,12,15,16,20,25} picks the rank.
__global__ __launch_bounds__(256) void median5x5_kernel(const float *__restrict__ dl, int H, int W,
                                                        float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    float v[25];
    int n = 0;
    bool has_nan = false;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int y = h + dy, x = w + dx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            const float t = in ? dl[(size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)] : __builtin_huge_valf();
            has_nan |= (t != t);
            n += in ? 1 : 0;
            v[(dy + 2) * 5 + dx + 2] = t;
        }
#pragma unroll
    for (int pass = 0; pass < 25; ++pass)
#pragma unroll
        for (int i = pass & 1; i + 1 < 25; i += 2) {
            const float lo = fminf(v[i], v[i + 1]), hi = fmaxf(v[i], v[i + 1]);
            v[i] = lo;
            v[i + 1] = hi;
        }
    float res;
    switch (n) {
        case 25: res = v[12]; break;
        case 15: res = v[7]; break;
        case 9: res = v[4]; break;
        case 20: res = (v[9] + v[10]) / 2.f; break;    // np.mean of the two middle float32 values
        case 16: res = (v[7] + v[8]) / 2.f; break;
        default: res = (v[5] + v[6]) / 2.f; break;     // 12
    }
    if (has_nan) res = __builtin_nanf("");  // np.median propagates NaN
    out[(size_t)h * W + w] = res;
}

// ---- a11 bilateral_filter (pf:440-466) --------------------------------------------------------------------------
// NumPy float32 add.reduce over n contiguous values (pairwise_sum, n <= 128), then + identity.
__device__ __forceinline__ float np_sum_small(const float *a, int n)
{
    float res;
    if (n < 8) {
        res = 0.f;
        for (int i = 0; i < n; ++i) res += a[i];
    } else {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
    }
    return 0.f + res;
}

__global__ __launch_bounds__(256) void bilateral_kernel(const float *__restrict__ img, const float *__restrict__ dl,
                                                        int H, int W, int fh, int fw, const float *__restrict__ table,
                                                        float thr, float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const int ch = (fh - 1) / 2, cw = (fw - 1) / 2;
    const int hs = max(0, h - ch), he = min(H, h + ch + 1), ws = max(0, w - cw), we = min(W, w + cw + 1);
    const float cur = img[(size_t)h * W + w];
    float wgt[MAXWIN], val[MAXWIN];
    int n = 0;
    for (int y = hs; y < he; ++y)
        for (int x = ws; x < we; ++x) {
            const float t = img[(size_t)y * W + x] - cur;
            const float sq = t * t;
            const float diff = sqrtf(sq);                                   // pf:458-459
            const float gate = diff < thr ? 1.f : 0.f;                      // pf:460
            const float f = gate * table[(ch + (y - h)) * fw + (cw + (x - w))];  // pf:462
            wgt[n] = f;
            val[n] = f * dl[(size_t)y * W + x];                             // pf:465
            ++n;
        }
    const float wsum = np_sum_small(wgt, n);  // pf:463
    const float vsum = np_sum_small(val, n);  // pf:466
    out[(size_t)h * W + w] = vsum / wsum;
}

// The reference's 5x5 window (match.py:170): interior pixels have all 25 taps, so the window walk and NumPy's summation
// order (8 running sums over taps 0..23, the fixed combine tree, then tap 24, then + identity) unroll completely and
// everything stays in registers; the general kernel keeps its taps in indexed arrays, i.e. in scratch memory.  Border
// pixels (clipped windows, other tap counts and orders) take the general code.
__global__ __launch_bounds__(256) void bilateral5x5_kernel(const float *__restrict__ img, const float *__restrict__ dl,
                                                           int H, int W, const float *__restrict__ table, float thr,
                                                           float *__restrict__ out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const float cur = img[(size_t)h * W + w];
    if (h >= 2 && h + 2 < H && w >= 2 && w + 2 < W) {
        float wgt[25], val[25];
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const size_t q = (size_t)(h + dy - 2) * W + (w + dx - 2);
                const float t = img[q] - cur;
                const float sq = t * t;
                const float diff = sqrtf(sq);                                   // pf:458-459
                const float gate = diff < thr ? 1.f : 0.f;                      // pf:460
                const float f = gate * table[dy * 5 + dx];                      // pf:462
                wgt[dy * 5 + dx] = f;
                val[dy * 5 + dx] = f * dl[q];                                   // pf:465
            }
        float sums[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float *a = k ? val : wgt;
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = a[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += a[8 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += a[16 + j];
            float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            res += a[24];
            sums[k] = 0.f + res;
        }
        out[(size_t)h * W + w] = sums[1] / sums[0];
        return;
    }
    const int hs = max(0, h - 2), he = min(H, h + 3), ws = max(0, w - 2), we = min(W, w + 3);
    float wgt[25], val[25];
    int n = 0;
    for (int y = hs; y < he; ++y)
        for (int x = ws; x < we; ++x) {
            const float t = img[(size_t)y * W + x] - cur;
            const float sq = t * t;
            const float diff = sqrtf(sq);
            const float gate = diff < thr ? 1.f : 0.f;
            const float f = gate * table[(2 + (y - h)) * 5 + (2 + (x - w))];
            wgt[n] = f;
            val[n] = f * dl[(size_t)y * W + x];
            ++n;
        }
    const float wsum = np_sum_small(wgt, n);
    const float vsum = np_sum_small(val, n);
    out[(size_t)h * W + w] = vsum / wsum;
}

// ---- a1 epilogues -------------------------------------------------------------------------------------------------
// bias (+ ReLU) in place over an NCHW tensor: blockIdx.y = n*C + c, 4 consecutive elements per thread (planes start at
// arbitrary 4-byte offsets, so the 16-byte accesses are declared 4-byte aligned)
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ x, const float *__restrict__ bias, int C,
                                                       long plane, int relu)
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const float b = bias[blockIdx.y % C];
    float *p = x + (size_t)blockIdx.y * plane;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < plane) {
        f4u v = *reinterpret_cast<f4u *>(p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = v[j] + b;
            v[j] = relu ? fmaxf(t, 0.f) : t;
        }
        *reinterpret_cast<f4u *>(p + i) = v;
    } else {
        for (long j = i; j < plane; ++j) {
            const float t = p[j] + b;
            p[j] = relu ? fmaxf(t, 0.f) : t;
        }
    }
}

// First layer of the stack (model.py:51-53: 1 -> C maps, 3x3 VALID, bias, ReLU) fused with the zero padding of the
// input (process_functional.py:20-25): 9 multiply-adds per output do not deserve a library convolution plus three
// elementwise launches.  Thread = one output pixel, looping over the C maps (weights / bias indexed uniformly: scalar
// loads); every map's store is a coalesced row segment.
__global__ __launch_bounds__(256) void conv1_pad_bias_relu_kernel(const float *__restrict__ img,
                                                                  const float *__restrict__ w,
                                                                  const float *__restrict__ bias,
                                                                  float *__restrict__ out, int H, int W, int pad, int C,
                                                                  int Ho, int Wo)
{
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int yo = blockIdx.y, n = blockIdx.z;
    if (xo >= Wo) return;
    float v[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int y = yo + i - pad, x = xo + j - pad;
            v[i * 3 + j] = (y >= 0 && y < H && x >= 0 && x < W) ? img[((size_t)n * H + y) * W + x] : 0.f;
        }
    float *o = out + ((size_t)n * C * Ho + yo) * Wo + xo;
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += w[c * 9 + k] * v[k];
        acc += bias[c];
        o[(size_t)c * Ho * Wo] = fmaxf(acc, 0.f);
    }
}

// tf.nn.l2_normalize(dim=-1) (model.py:64) of (conv output + last bias), NCHW in -> NHWC out: 64 pixels x C channels
// per workgroup through a padded LDS tile: plane rows are read in 256-B runs and each pixel's C-vector is written as
// one contiguous run.
template <int C>
__global__ __launch_bounds__(256) void l2norm_chw_to_hwc_kernel(const float *__restrict__ chw,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ hwc, long N)
{
    __shared__ float tile[C][65];
    __shared__ float scale[64];
    const long n0 = (long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < C; c += 4)
        tile[c][tx] = (n0 + tx < N) ? chw[(size_t)c * N + n0 + tx] + (bias ? bias[c] : 0.f) : 0.f;
    __syncthreads();
    if (ty == 0) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(tile[c][tx], tile[c][tx], s);
        s = fmaxf(s, 1e-12f);
        scale[tx] = 1.f / sqrtf(s);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
        const int px = i / C, c = i - px * C;
        if (n0 + px < N) hwc[(size_t)(n0 + px) * C + c] = tile[c][px] * scale[px];
    }
}

}  // namespace mccnn

extern "C" int mccnn_wta(const float *vol_dhw, int D, int H, int W, float *disparity, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(vol_dhw && disparity, MCCNN_E_INVALID, "mccnn_wta: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_wta: non-positive size");
    const long N = (long)H * W;
    hipLaunchKernelGGL(wta_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, vol_dhw, D, N, disparity);
    return check_launch("mccnn_wta");
}

extern "C" int mccnn_lr_status(const float *disp_left, const float *disp_right, int H, int W, int D, int32_t *status,
                               mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp_left && disp_right && status, MCCNN_E_INVALID, "mccnn_lr_status: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_lr_status: non-positive size");
    if (W <= 32 * LR_MAX_WORDS)
        hipLaunchKernelGGL(lr_status_kernel, dim3(H), dim3(256), 0, (hipStream_t)stream, disp_left, disp_right, H, W, D,
                           status);
    else
        hipLaunchKernelGGL(lr_status_walk_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, disp_left,
                           disp_right, H, W, D, status);
    return check_launch("mccnn_lr_status");
}

extern "C" int mccnn_interpolate(const float *disp_left, const int32_t *status, int H, int W, float *out,
                                 mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp_left && status && out, MCCNN_E_INVALID, "mccnn_interpolate: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_interpolate: non-positive size");
    MCCNN_REQUIRE(disp_left != out, MCCNN_E_INVALID, "mccnn_interpolate: out must not alias the input map");
    if (H < 65535 && W <= 64 * INTERP_MAX_WORDS && (const void *)status != (const void *)out) {
        hipLaunchKernelGGL(interpolate_vertical_kernel, dim3(cdiv(W, 64), 2), dim3(64), 0, (hipStream_t)stream, status, H,
                           W, reinterpret_cast<uint16_t *>(out));
        int rc = check_launch("mccnn_interpolate(vertical)");
        if (rc) return rc;
        hipLaunchKernelGGL(interpolate_row_kernel, dim3(H), dim3(256), 0, (hipStream_t)stream, disp_left, status, H, W,
                           out);
        return check_launch("mccnn_interpolate(rows)");
    }
    hipLaunchKernelGGL(interpolate_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, disp_left, status,
                       H, W, out);
    return check_launch("mccnn_interpolate");
}

extern "C" int mccnn_interpolate_ex(const float *disp_left, const int32_t *status, int H, int W, int directions,
                                    int occlusion_from_left, float *out, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp_left && status && out, MCCNN_E_INVALID, "mccnn_interpolate_ex: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_interpolate_ex: non-positive size");
    MCCNN_REQUIRE(disp_left != out, MCCNN_E_INVALID, "mccnn_interpolate_ex: out must not alias the input map");
    MCCNN_REQUIRE(directions == 4 || directions == 16, MCCNN_E_INVALID, "mccnn_interpolate_ex: directions=%d (4 or 16)",
                  directions);
    const dim3 grid(cdiv(W, 256), H), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (directions == 16 && occlusion_from_left)
        hipLaunchKernelGGL((interpolate_paper_kernel<true, true>), grid, block, 0, s, disp_left, status, H, W, out);
    else if (directions == 16)
        hipLaunchKernelGGL((interpolate_paper_kernel<true, false>), grid, block, 0, s, disp_left, status, H, W, out);
    else if (occlusion_from_left)
        hipLaunchKernelGGL((interpolate_paper_kernel<false, true>), grid, block, 0, s, disp_left, status, H, W, out);
    else
        hipLaunchKernelGGL(interpolate_kernel, grid, block, 0, s, disp_left, status, H, W, out);
    return check_launch("mccnn_interpolate_ex");
}

extern "C" int mccnn_subpixel_ex(const float *disp, const float *vol_dhw, int D, int H, int W, int numpy1_promotion,
                                 float *out, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp && vol_dhw && out, MCCNN_E_INVALID, "mccnn_subpixel_ex: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_subpixel_ex: non-positive size");
    const long N = (long)H * W;
    if (numpy1_promotion)
        hipLaunchKernelGGL(subpixel_kernel<true>, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, disp, vol_dhw, D,
                           N, out);
    else
        hipLaunchKernelGGL(subpixel_kernel<false>, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, disp, vol_dhw,
                           D, N, out);
    return check_launch("mccnn_subpixel_ex");
}

extern "C" int mccnn_subpixel(const float *disp, const float *vol_dhw, int D, int H, int W, float *out,
                              mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp && vol_dhw && out, MCCNN_E_INVALID, "mccnn_subpixel: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_subpixel: non-positive size");
    const long N = (long)H * W;
    hipLaunchKernelGGL(subpixel_kernel<false>, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, disp, vol_dhw, D, N,
                       out);
    return check_launch("mccnn_subpixel");
}

extern "C" int mccnn_median(const float *disp, int H, int W, int fh, int fw, float *out, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(disp && out, MCCNN_E_INVALID, "mccnn_median: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_median: non-positive size");
    MCCNN_REQUIRE(disp != out, MCCNN_E_INVALID, "mccnn_median: out must not alias the input map");
    MCCNN_REQUIRE(fh >= 1 && fw >= 1 && (fh & 1) && (fw & 1) && fh * fw <= MAXWIN, MCCNN_E_UNSUPPORTED,
                  "mccnn_median: window %dx%d must be odd x odd with at most %d taps", fh, fw, MAXWIN);
    if (fh == 5 && fw == 5 && H >= 5 && W >= 5)
        hipLaunchKernelGGL(median5x5_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, disp, H, W, out);
    else
        hipLaunchKernelGGL(median_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, disp, H, W, fh, fw,
                           out);
    return check_launch("mccnn_median");
}

extern "C" int mccnn_bilateral(const float *image, const float *disp, int H, int W, int fh, int fw, const float *table,
                               float thr, float *out, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image && disp && table && out, MCCNN_E_INVALID, "mccnn_bilateral: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_bilateral: non-positive size");
    MCCNN_REQUIRE(disp != out, MCCNN_E_INVALID, "mccnn_bilateral: out must not alias the input map");
    MCCNN_REQUIRE(fh >= 1 && fw >= 1 && (fh & 1) && (fw & 1) && fh * fw <= MAXWIN, MCCNN_E_UNSUPPORTED,
                  "mccnn_bilateral: window %dx%d must be odd x odd with at most %d taps", fh, fw, MAXWIN);
    if (fh == 5 && fw == 5)
        hipLaunchKernelGGL(bilateral5x5_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, image, disp, H,
                           W, table, thr, out);
    else
        hipLaunchKernelGGL(bilateral_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, image, disp, H, W,
                           fh, fw, table, thr, out);
    return check_launch("mccnn_bilateral");
}

extern "C" int mccnn_bias_act(float *x, const float *bias, int N, int C, long plane, int relu, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(x && bias, MCCNN_E_INVALID, "mccnn_bias_act: null pointer");
    MCCNN_REQUIRE(N > 0 && C > 0 && plane > 0, MCCNN_E_INVALID, "mccnn_bias_act: non-positive size");
    MCCNN_REQUIRE((long)N * C <= 65535, MCCNN_E_UNSUPPORTED, "mccnn_bias_act: N*C=%ld exceeds grid.y", (long)N * C);
    hipLaunchKernelGGL(bias_act_kernel, dim3(cdiv(plane, 1024), N * C), dim3(256), 0, (hipStream_t)stream, x, bias, C,
                       plane, relu);
    return check_launch("mccnn_bias_act");
}

extern "C" int mccnn_conv1_pad_bias_relu(const float *images, const float *weights, const float *bias, float *out,
                                         int N, int H, int W, int pad, int C, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(images && weights && bias && out, MCCNN_E_INVALID, "mccnn_conv1_pad_bias_relu: null pointer");
    MCCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && pad >= 0, MCCNN_E_INVALID,
                  "mccnn_conv1_pad_bias_relu: bad size");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    MCCNN_REQUIRE(Ho > 0 && Wo > 0 && Ho <= 65535 && N <= 65535, MCCNN_E_UNSUPPORTED,
                  "mccnn_conv1_pad_bias_relu: output %dx%d outside the grid", Wo, Ho);
    hipLaunchKernelGGL(conv1_pad_bias_relu_kernel, dim3(cdiv(Wo, 256), Ho, N), dim3(256), 0, (hipStream_t)stream,
                       images, weights, bias, out, H, W, pad, C, Ho, Wo);
    return check_launch("mccnn_conv1_pad_bias_relu");
}

extern "C" int mccnn_l2norm_chw_to_hwc(const float *chw, const float *bias, float *hwc, int C, int H, int W,
                                       mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(chw && hwc, MCCNN_E_INVALID, "mccnn_l2norm_chw_to_hwc: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_l2norm_chw_to_hwc: non-positive size");
    MCCNN_REQUIRE(C == 64, MCCNN_E_UNSUPPORTED, "mccnn_l2norm_chw_to_hwc: C=%d, built for 64 feature maps", C);
    const long N = (long)H * W;
    hipLaunchKernelGGL((l2norm_chw_to_hwc_kernel<64>), dim3(cdiv(N, 64)), dim3(256), 0, (hipStream_t)stream, chw, bias,
                       hwc, N);
    return check_launch("mccnn_l2norm_chw_to_hwc");
}
