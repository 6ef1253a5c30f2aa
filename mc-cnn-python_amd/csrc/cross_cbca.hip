// a3 compute_cross_region (/root/reference/src/process_functional.py:571-657) and
// a4 cost_volume_aggregation (pf:117-183) on gfx950.
//
// The reference materialises, per pixel, the coordinate list of its cross-shaped support region (int32
// [H,W,784,2], 2.35 GB per image at 750x500) and then gathers through it.  Here a pixel carries
// one packed 32-bit word (four 5-bit arm lengths + the 12-bit region size, mccnn_support_t) plus two derived words
// for the streaming kernel (ready-made LDS addresses of its horizontal lookups; reciprocal of the size + vertical arms);
// the region is regenerated from the arms.
//
// Two aggregation kernels, same result set:
//   cbca_stream_kernel  (MCCNN_CBCA_SEPARABLE, default distance) - O(1) work per output via float64 prefix sums,
//                       wave-specialised stages stream 256-column strips of two disparity planes; 8 B/voxel/iteration
//                       of HBM traffic.
//   cbca_iter_kernel    LDS-tiled; REFERENCE_ORDER variant walks the region in the reference's list order and is
//                       bit-exact; its separable variant serves distances > 14.
#include <math.h>

#include <algorithm>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "support.h"

namespace mccnn {

// np.linalg.norm of the 1-vector (cur - other): sqrt(x*x), float32 (pf:588,596,615,623)
__device__ __forceinline__ float norm1(float x)
{
    const float sq = x * x;
    return sqrtf(sq);
}

// blockIdx.z selects the image (mccnn_cross_arms_pair: both views in one launch)
__global__ __launch_bounds__(256) void cross_arms_kernel(const float *__restrict__ img0, const float *__restrict__ img1,
                                                         int H, int W, float tau, int L, Support *__restrict__ sup0,
                                                         Support *__restrict__ sup1)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    const float *__restrict__ img = blockIdx.z ? img1 : img0;
    Support *__restrict__ sup = blockIdx.z ? sup1 : sup0;
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

// ---------------------------------------------------------------------------------------------------------------
// Geometry of the streaming aggregation kernel (cbca_stream_kernel below).  It is shared with cross_count_kernel
// because the support planes carry, per pixel, ready-made LDS byte addresses for that kernel's layout: the pixel's
// position inside its strip (and hence the lane and the LDS slot of every prefix-sum entry it needs) is a function of
// its column alone, so the address arithmetic is done once per image instead of once per pixel per plane per iteration.
namespace s4 {
constexpr int R = 13;             // longest arm the kernel serves (distance threshold L <= 14)
constexpr int CPL = 4;            // OUTPUT columns per lane (hsum / emit stages, 16-byte stores)
constexpr int CPS = 5;            // STAGED columns per lane of the scan stage (one 16-byte + one 4-byte load per row)
constexpr int CS = 64 * CPS;      // staged columns per strip
constexpr int HL = 15;            // staged columns left of the first output column (>= R + 1, multiple of CPS so that
                                  // a scan lane lies entirely inside or entirely outside the image at its left edge)
constexpr int OUTW = 64 * CPL;    // 256 output columns per strip: every lane of the hsum / emit waves is used, and
                                  // a 750-wide image is 3 strips (with 4 columns per scan lane and 224 outputs it was 4)
#ifndef CBCA_B
#define CBCA_B 3
#endif
constexpr int B = CBCA_B;         // rows per pipeline batch (the kernel runs at its LDS / vector-memory throughput: 2, 3
                                  // or 4 rows per barrier measure the same; 3 is what the double-buffered prow leaves
                                  // room for beside the rings of two planes in 160 KiB)
constexpr int RING = 32;          // rows of the column-prefix ring (power of two: the wrap is a bit mask)
constexpr int SUBB = 64 * 8;      // one column-phase sub-array: 64 lanes x 8 bytes
constexpr int ROWB = CPL * SUBB;  // bytes of one ring row (float64 column prefixes of the 256 outputs)
constexpr int PROWB = CPS * SUBB; // bytes of one prow row (float64 row prefixes of the 320 staged columns)
static_assert(HL >= R + 1 && HL % CPS == 0 && HL + OUTW + R <= CS, "strip geometry");
static_assert(ROWB == 2048 && RING * ROWB == 65536, "the emit stage's address math assumes 2 KiB rows, 32 of them");
static_assert(RING >= B + 2 * R + 1, "ring must hold the emit window plus the batch being written");
// Column-phase layout of a prow row: staged entry i lives in sub-array i % 5 at slot i / 5 (the scan lane that owns
// it), so the five values a lane owns go out as conflict-free 8-byte stores (lane stride 8 B) and gathers of
// neighbouring lanes that use equal arms hit distinct banks.  Ring rows use the same scheme with 4 phases.
__host__ __device__ constexpr uint32_t elem(int i) { return (uint32_t)((i % CPS) * SUBB + (i / CPS) * 8); }
}  // namespace s4

// Support buffer: plane 0 [H][W] uint32 (arms + size, documented in mccnn.h), then - each 16-byte aligned -
//   plane 1 [H][W] uint32  "hsum words": LDS byte offsets (within one prow row) of the two prefix entries whose
//                          difference is the pixel's horizontal-arm sum: bits 0-15 entry i+right, 16-31 entry i-left-1,
//                          i = column % OUTW + HL;
//   plane 2 [H][W] uint64  "emit words": the float64 reciprocal of the region size, whose 12 low mantissa bits carry
//                          down (bits 7-11) and 31 - up (bits 2-6); the upper 40 mantissa bits are chosen so that the
//                          word AS IT IS (fields included) is the float64 nearest to 1/n among the words with those low
//                          bits - the kernel multiplies by it without masking (|rel. error| <= 2^-41).

// pf:640-653: region size = sum over the vertical arm of the horizontal arm sizes.  The size goes into the upper 12
// bits of the same word whose lower 20 bits (the arms, written by the previous kernel and never changed here) the
// neighbours are reading: relaxed atomics make that formally race-free, and any mix of old/new words is correct.
// The same kernel fills the derived planes of the support buffer (above).
__global__ __launch_bounds__(256) void cross_count_kernel(Support *__restrict__ sup0, Support *__restrict__ sup1, int H,
                                                          int W)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (w >= W) return;
    Support *__restrict__ sup = blockIdx.z ? sup1 : sup0;
    auto ld = [&](size_t i) { return __hip_atomic_load(&sup[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const size_t p = (size_t)h * W + w;
    const uint32_t a = ld(p);
    uint32_t n = 0;
    for (int q = h - arm_up(a); q <= h + arm_down(a); ++q) {
        const uint32_t aq = ld((size_t)q * W + w);
        n += arm_left(aq) + arm_right(aq) + 1;
    }
    __hip_atomic_fetch_or(&sup[p], n << 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    char *base = reinterpret_cast<char *>(sup);
    // arms beyond what the streaming kernel serves (L > 14 never reaches it) are clamped so the words stay in range
    const int up = min(arm_up(a), s4::R), down = min(arm_down(a), s4::R);
    const int left = min(arm_left(a), s4::R), right = min(arm_right(a), s4::R);
    const int i = w % s4::OUTW + s4::HL;
    reinterpret_cast<uint32_t *>(base + hsum_plane_offset(H, W))[p] = s4::elem(i + right) | (s4::elem(i - left - 1) << 16);
    const unsigned long long rbits = (unsigned long long)__double_as_longlong(1.0 / (double)n);
    const unsigned long long field = ((unsigned long long)down << 7) | ((unsigned long long)(31 - up) << 2);
    reinterpret_cast<unsigned long long *>(base + emit_plane_offset(H, W))[p] =
        (((rbits - field + 0x800ull) >> 12) << 12) | field;
    // window mask of the pixel-major reference-order kernel (support.h)
    reinterpret_cast<hwd_mask_t *>(base + wmask_plane_offset(H, W))[p] =
        (hwd_mask_t)(((2ull << (left + right)) - 1ull) << (w % HWD_G + HWD_R - left));
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

// Counting sort of a 16 x 64 tile's pixels by falling walk cost (ties in arbitrary order: the order only decides
// which lane walks which pixel).  Pixels outside the image sort last.  blockIdx.z selects the view.
__global__ __launch_bounds__(256) void cross_perm_kernel(const Support *__restrict__ sup0, const Support *__restrict__ sup1,
                                                         int H, int W)
{
    constexpr int NB = 1024;                    // bins: keys are below 28 * 32
    __shared__ int bins[NB];
    __shared__ int wsum[4];
    const Support *sup = blockIdx.z ? sup1 : sup0;
    uint16_t *perm = reinterpret_cast<uint16_t *>(const_cast<char *>(reinterpret_cast<const char *>(sup)) +
                                                  perm_plane_offset(H, W)) +
                     ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (PERM_TH * CB_TW_C);
    const int tid = threadIdx.x, w0 = blockIdx.x * CB_TW_C, h0 = blockIdx.y * PERM_TH;
    for (int i = tid; i < NB; i += 256) bins[i] = 0;
    __syncthreads();
    int key[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = tid + 256 * k, hh = h0 + (idx >> 6), ww = w0 + (idx & 63);
        // key: number of rows of the region (1 + up + down) first, its mean width second - the walk costs a wave the
        // most rows among its lanes times the widest arm on each of them, so rows matter most (keyed by region size
        // alone: 2.10 instead of 1.87 ms per iteration)
        int n = 0;
        if (hh < H && ww < W) {
            const Support a = sup[(size_t)hh * W + ww];
            const int rows = 1 + min((int)arm_up(a), 13) + min((int)arm_down(a), 13);      // 1 .. 27
            n = (rows << 5) | min((int)sup_count(a) / rows, 31);
        }
        key[k] = NB - 1 - n;                    // ascending key = falling cost
        atomicAdd(&bins[key[k]], 1);
    }
    __syncthreads();
    // exclusive prefix over the bins: 4 bins per thread, wave scan, wave totals through LDS
    const int b0 = bins[4 * tid], b1 = bins[4 * tid + 1], b2 = bins[4 * tid + 2], b3 = bins[4 * tid + 3];
    int incl = b0 + b1 + b2 + b3;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - (b0 + b1 + b2 + b3);
    for (int wv = 0; wv < (tid >> 6); ++wv) base += wsum[wv];
    __syncthreads();
    bins[4 * tid] = base;
    bins[4 * tid + 1] = base + b0;
    bins[4 * tid + 2] = base + b0 + b1;
    bins[4 * tid + 3] = base + b0 + b1 + b2;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) perm[atomicAdd(&bins[key[k]], 1)] = (uint16_t)(tid + 256 * k);
}

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
    // arms are clamped to the staged halo: a support plane built with a larger distance than the caller states
    // (mccnn_cbca_iter rejects that when it can tell) must not walk outside the tile
    auto aL = [](uint32_t a) { return min(arm_left(a), R); };
    auto aR = [](uint32_t a) { return min(arm_right(a), R); };
    auto aU = [](uint32_t a) { return min(arm_up(a), R); };
    auto aD = [](uint32_t a) { return min(arm_down(a), R); };
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
                for (int j = -aL(a); j <= aR(a); ++j) s += row[j];
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
                for (int i = -aU(sp); i <= aD(sp); ++i) s += col[i * CB_TW];
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
                const int nu = aU(sp), nv = 1 + nu + aD(sp);
                for (int v = 0; v < nv; ++v) {
                    const int dq = v == 0 ? 0 : (v <= nu ? -v : v - nu);
                    const uint32_t aq = sup[(size_t)(hh + dq) * W + ww];
                    const float *row = &tin[(r + R + dq) * IP + c + R];
                    s += row[0];
                    for (int z = 1; z <= aL(aq); ++z) s += row[-z];
                    for (int z = 1; z <= aR(aq); ++z) s += row[z];
                }
                dst[(size_t)hh * W + ww] = s / (float)sup_count(sp);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Support regions from BOTH views (MC-CNN paper, sec. 4.1; the reference leaves it out as "impractical to run" and its
// attempt, compute_disparity_union_region pf:661-729, is dead code with a NameError).  Opt-in, changes the output.
// For the volume of side S at disparity d, a pixel q = (y, x) has the partner q' = (y, x - d) (S = left) or (y, x + d)
// (S = right) in the other view; every arm used at q is min(arm of S at q, arm of the other view at q'), for the
// vertical arm of the anchor and for the horizontal arm of every pixel on it.  A pixel whose partner falls outside
// the image keeps its own arms.  Same tile staging and the same flat float32 running sum (vertical: self, up..,
// down..; horizontal: self, left.., right..) as the reference-order kernel; the region size is counted on the way.
template <int R, int CB_TH>
__global__ __launch_bounds__(256) void cbca_both_views_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                              const Support *__restrict__ sup,
                                                              const Support *__restrict__ sup_other, int H, int W,
                                                              int dsign)
{
    constexpr int IW = CB_TW + 2 * R, IH = CB_TH + 2 * R, IP = IW + 1;
    __shared__ float tin[IH * IP];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * CB_TW, h0 = blockIdx.y * CB_TH;
    const int shift = dsign * (int)blockIdx.z;       // partner column = x + shift
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
    const int c = tid & 63;
    const int ww = w0 + c;
    const int xo = ww + shift;
    const bool partner = xo >= 0 && xo < W;
    // the four arms at (qy, ww), intersected with the partner's: {up, down, left, right}
    auto arms = [&](int qy, int &u, int &d, int &l, int &r) {
        const uint32_t a = sup[(size_t)qy * W + ww];
        u = min(arm_up(a), R), d = min(arm_down(a), R), l = min(arm_left(a), R), r = min(arm_right(a), R);
        if (partner) {
            const uint32_t b = sup_other[(size_t)qy * W + xo];
            u = min(u, arm_up(b)), d = min(d, arm_down(b)), l = min(l, arm_left(b)), r = min(r, arm_right(b));
        }
    };
    for (int k = 0; k < CB_TH / 4; ++k) {
        const int rr = (tid >> 6) + 4 * k;
        const int hh = h0 + rr;
        if (hh < H && ww < W) {
            int nu, nd, l0, r0;
            arms(hh, nu, nd, l0, r0);
            float s = 0.f;
            int n = 0;
            const int nv = 1 + nu + nd;
            for (int v = 0; v < nv; ++v) {
                const int dq = v == 0 ? 0 : (v <= nu ? -v : v - nu);
                int u, d, l, r;
                arms(hh + dq, u, d, l, r);
                const float *row = &tin[(rr + R + dq) * IP + c + R];
                s += row[0];
                for (int z = 1; z <= l; ++z) s += row[-z];
                for (int z = 1; z <= r; ++z) s += row[z];
                n += l + r + 1;
            }
            dst[(size_t)hh * W + ww] = s / (float)n;
        }
    }
}

// The reference's flat running sum (pf:157-161) for FOUR planes at once.  The walk over a pixel's region is one
// dependent float32 chain per output by definition, and the kernel above spends its time on what surrounds each
// link of it (an LDS read, the wait, the loop bookkeeping of a divergent walk: ~8 instructions per element, with the
// wave waiting for its longest walk - on the synthetic pair the mean region has 30 pixels, the mean over waves of the
// largest region 127): 3.5 of its 4.1 ms per iteration at 750x500x256.  The support region does not
// depend on the disparity, so a lane can walk once for its pixel in four neighbouring planes: the tile is staged as
// float4 per pixel, one ds_read_b128 fetches the four values and two v_pk_add_f32 advance the four chains - the
// same additions in the same order per plane (packed float32 adds are IEEE adds), a quarter of the instructions:
// 3.3 ms.  Measured and dropped: the arms in LDS instead of a global load per row (3.5 ms: the lower occupancy costs
// more), chunked walks with the reads of a chunk issued together and lanes past their arm adding -0.0f, the exact
// identity (3.6 ms: more work for the typical short arm).  What is left is divergence - lanes waiting for the longest
// walk in their wave.
typedef float cb_f2 __attribute__((ext_vector_type(2)));

template <int R, int TH>
__global__ __launch_bounds__(256) void cbca_ref4_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const Support *__restrict__ sup, int D, int H, int W)
{
    constexpr int IW = CB_TW + 2 * R, IH = TH + 2 * R;
    __shared__ float4 tin[IH * IW];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * CB_TW, h0 = blockIdx.y * TH, d0 = blockIdx.z * 4;
    const size_t plane = (size_t)H * W;
    auto aL = [](uint32_t a) { return min(arm_left(a), R); };
    auto aR = [](uint32_t a) { return min(arm_right(a), R); };
    auto aU = [](uint32_t a) { return min(arm_up(a), R); };
    auto aD = [](uint32_t a) { return min(arm_down(a), R); };
    const float *src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) src[p] = in + (size_t)min(d0 + p, D - 1) * plane;   // a short last group repeats a plane
    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int hh = h0 - R + r, ww = w0 - R + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
            const size_t q = (size_t)hh * W + ww;
            v = make_float4(src[0][q], src[1][q], src[2][q], src[3][q]);
        }
        tin[i] = v;
    }
    __syncthreads();
    // Pixels are dealt to lanes in the order of the tile's permutation plane (falling region size): the 64 lanes of a
    // wave then walk regions of nearly equal size instead of waiting for the largest of 64 neighbours.  The 16 groups
    // of 64 go to the four waves in snake order (w, 7-w, 8+w, 15-w), so the waves finish together too.
    static_assert(TH == PERM_TH, "the permutation plane is built for 16-row tiles");
    const uint16_t *perm = reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(sup) + perm_plane_offset(H, W)) +
                           ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (TH * CB_TW);
    const int wv = tid >> 6, ln = tid & 63;
    for (int k = 0; k < TH / 4; ++k) {
        const int grp = (k & 1) ? 4 * k + 3 - wv : 4 * k + wv;
        const int idx = perm[grp * 64 + ln];
        const int r = idx >> 6, c = idx & 63;
        const int hh = h0 + r, ww = w0 + c;
        if (hh < H && ww < W) {
            const Support sp = sup[(size_t)hh * W + ww];
            cb_f2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
            const int nu = aU(sp), nv = 1 + nu + aD(sp);
            auto add = [&](const float4 x) {
                const cb_f2 a = {x.x, x.y}, b = {x.z, x.w};
                s01 += a;
                s23 += b;
            };
            // the arms of the next row are fetched while this row is summed, and the first PRE entries of both arms
            // are read together with the anchor (unconditionally: the staged halo is R >= PRE columns wide)
            constexpr int PRE = R < 2 ? R : 2;
            auto offset = [&](int v) { return v == 0 ? 0 : (v <= nu ? -v : v - nu); };
            uint32_t aq_next = sp;
            for (int v = 0; v < nv; ++v) {
                const int dq = offset(v);
                const uint32_t aq = aq_next;
                if (v + 1 < nv) aq_next = sup[(size_t)(hh + offset(v + 1)) * W + ww];
                const float4 *row = &tin[(r + R + dq) * IW + c + R];
                const int nl = aL(aq), nr = aR(aq);
                const float4 x0 = row[0];
                float4 xl[PRE], xr[PRE];
#pragma unroll
                for (int z = 0; z < PRE; ++z) {
                    xl[z] = row[-1 - z];
                    xr[z] = row[1 + z];
                }
                add(x0);
#pragma unroll
                for (int z = 0; z < PRE; ++z)
                    if (z < nl) add(xl[z]);
                for (int z = PRE + 1; z <= nl; ++z) add(row[-z]);
#pragma unroll
                for (int z = 0; z < PRE; ++z)
                    if (z < nr) add(xr[z]);
                for (int z = PRE + 1; z <= nr; ++z) add(row[z]);
            }
            const float n = (float)sup_count(sp);
            const float res[4] = {s01.x / n, s01.y / n, s23.x / n, s23.y / n};   // pf:161
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (d0 + p < D) out[(size_t)(d0 + p) * plane + (size_t)hh * W + ww] = res[p];
        }
    }
}

template <int R, int CB_TH>
static int launch_cbca(const float *in, float *out, const Support *sup, int D, int H, int W, int order, hipStream_t s)
{
    const dim3 grid(cdiv(W, CB_TW), cdiv(H, CB_TH), D), block(256);
    if (order == MCCNN_CBCA_REFERENCE_ORDER && R <= 13) {
        constexpr int TH4 = 16;   // 42 x 90 float4 = 59 KiB of LDS: two workgroups per CU
        hipLaunchKernelGGL((cbca_ref4_kernel<(R <= 13 ? R : 13), TH4>), dim3(cdiv(W, CB_TW), cdiv(H, TH4), cdiv(D, 4)),
                           block, 0, s, in, out, sup, D, H, W);
    } else if (order == MCCNN_CBCA_REFERENCE_ORDER)
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, true>), grid, block, 0, s, in, out, sup, H, W);
    else
        hipLaunchKernelGGL((cbca_iter_kernel<R, CB_TH, false>), grid, block, 0, s, in, out, sup, H, W);
    return check_launch("mccnn_cbca_iter");
}

// ---------------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// cbca_stream_kernel: the separable aggregation (MCCNN_CBCA_SEPARABLE, L <= 14), O(1) work per output.
//
// A strip of 256 output columns (284 staged: 15 + 13 halo columns) of one disparity plane is streamed down its rows:
//   row y arrives (5 floats per lane of the scan wave)
//     -> float64 inclusive prefix sum P along the row (lane-local prefix + DPP scan across the 64 lanes)   [scan]
//     -> horizontal-arm sum of pixel (y,c) = P[i+right] - P[i-left-1]; running float64 column prefix
//        Q[y][c] += that, kept in a 32-row LDS ring                                                        [hsum]
//   row y-13 leaves: vertical-arm sum = Q[y'+down] - Q[y'-up-1], times 1/|U|, rounded once to float32     [emit]
// float64 differences of prefix sums of float32 data carry ~1e-14 absolute error, so the result is the correctly
// rounded region mean (up to the 2^-41 of the stored reciprocal); it differs from the reference's sequential float32
// sum only by that sum's own rounding.  Lanes never diverge and the cost does not depend on the arm lengths.
//
// What a row costs, and why it is laid out this way (MI355X, measured; docs/DESIGN_LOG_r1-r4.md 4.2 has the numbers):
//   * LDS addresses come ready-made in the support words (namespace s4): a horizontal lookup is one AND or one shift,
//     a vertical lookup is v_lshl_add + v_and_or (the 32-row ring wraps by mask), the reciprocal is used as stored;
//   * prow and the ring use the column-phase layout (s4::elem): stores and the ring gathers are conflict-free by
//     construction, prow gathers are conflict-free wherever neighbouring lanes use equal arms;
//   * 16-byte stores and 16+4-byte loads per lane, halo columns 10 % of a strip, all 64 lanes of the hsum / emit waves
//     busy, the 64-lane scan paid once per 256 outputs; 750 columns are 3 strips.
// Pipeline: waves are specialised by stage and run in lock step, one s_barrier per batch of B = 3 rows:
//     iteration t:  scan  batch t    -> prow[t & 1]
//                   hsum  batch t-1  -> ring rows B(t-1) .. B(t-1)+B-1
//                   emit  batch t-2  :  "above" lookups Q[y+down] now; the "below" lookups Q[y-up-1] were fetched one
//                                       iteration earlier (rows <= y-1 are long complete) - with that the 27-row
//                                       window plus the rows being written fit the 32-row ring.
// scan x3 (one row each), hsum x1, emit x3 (one row each): seven waves per plane.
// A workgroup runs PPW = 2 such pipelines (two planes of the same strip) behind the same barriers: the second one's
// support words are the cache lines the first one just brought into L1 (TCP->L2 read requests -35 %).
// Edges: nothing is clamped along a row.  Lanes whose columns fall outside the image read whatever the buffer range
// check or the neighbouring row gives them (finite: volumes must be finite); no arm reaches those entries, so they
// cancel in the prefix differences.  Rows past the bottom re-read the last row (never referenced).  Stores outside the
// image / chunk are dropped by an out-of-range offset.
#ifndef CBCA_NSCAN
#define CBCA_NSCAN 3   // scan waves per pipeline (rows of a batch split between them: one row each)
#define CBCA_NEMIT 3   // emit waves per pipeline (rows split: one row each)
#endif
#ifndef CBCA_NHS
#define CBCA_NHS 1     // hsum waves per pipeline (each owns CPL / NHS of a lane's four columns)
#endif
#ifndef CBCA_PPW
#define CBCA_PPW 2     // pipelines (planes) per workgroup
#endif
constexpr int NHS = CBCA_NHS;
constexpr int PPW = CBCA_PPW;

// One pipeline step: workgroup barrier, and nothing may be scheduled across it - without the scheduling barriers the
// compiler hoists the register-only unpacking of all four unrolled batches above the first s_barrier, which turns
// the s_waitcnt vmcnt(N) of the oldest batch into vmcnt(0) and serialises every iteration behind a fresh HBM load.
__device__ __forceinline__ void pipe_sync()
{
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
}

// One launch aggregates up to two volumes of the same shape (the left and the right view's, each with its own
// support planes): the work items of both are dealt from one pool, so 750x500x256 is 768 items = exactly three
// rounds of 256 resident workgroups instead of 2 x (one round + a chunked half round).
struct StreamJobs {
    const float *in[2];
    float *out[2];
    const Support *sup[2];
    int n;
};

template <int NSCAN, int NEMIT>
__global__ __launch_bounds__(64 * PPW * (NSCAN + NHS + NEMIT)) void cbca_stream_kernel(
    const StreamJobs jobs, int D, int H, int W, int rows, int nstrips, int nchunks, int total, int nfull)
{
    using namespace s4;
#ifndef CBCA_NPF
#define CBCA_NPF 4
#endif
    constexpr int NPF = CBCA_NPF;               // batches of loads every wave keeps in flight
    constexpr int SR = B / NSCAN, ER = B / NEMIT;
    constexpr int PROW_BYTES = 2 * B * PROWB;   // double-buffered by batch parity
    constexpr int NW = NSCAN + NHS + NEMIT;     // waves of one pipeline
    __shared__ double lds[PPW * (PROW_BYTES + RING * ROWB) / 8];   // 79 KiB per pipeline
    const int lane = threadIdx.x & 63;
    const int wave_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 of a workgroup share a SIMD: the second pipeline takes its roles in rotated order so that a
    // pipeline's longest instruction stream (scan) is not paired with its twin
    const int pipe = wave_wg / NW, wave = (wave_wg % NW + pipe * (NW / 2)) % NW;
    char *const ldsb = reinterpret_cast<char *>(lds) + pipe * (PROW_BYTES + RING * ROWB);
    // Work items.  The first `nfull` workgroups (a multiple of 8, dispatched first) each take one (strip, plane group) at
    // full height; the remaining ones take the remaining (strip, plane group) pairs cut into `nchunks` row chunks of
    // `rows` rows (each re-stages 2R halo rows).  Inside either group the order is XCD-aware: the dispatcher deals
    // consecutive workgroups to the 8 XCDs round-robin, and every XCD gets a contiguous range of items, so that
    // neighbouring strips of one plane (shared halo columns, shared support rows) meet in one L2.
    auto xcd_order = [](int b, int n) {
        const int q = n >> 3, r = n & 7, x = b & 7;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    };
    int item, chunk = 0, nrows = H;
    if ((int)blockIdx.x < nfull) {
        item = xcd_order(blockIdx.x, nfull);
    } else {
        const int id = xcd_order(blockIdx.x - nfull, total - nfull);
        item = nfull + id / nchunks;
        chunk = id % nchunks;
        nrows = rows;
    }
    const int per_job = nstrips * ((D + PPW - 1) / PPW);
    const int job = item >= per_job;            // at most two jobs
    item -= job * per_job;
    const float *const in = job ? jobs.in[1] : jobs.in[0];
    float *const out = job ? jobs.out[1] : jobs.out[0];
    const Support *const sup = job ? jobs.sup[1] : jobs.sup[0];
    const int strip = item % nstrips;
    // an odd last plane is simply done by both pipelines (identical stores)
    const int d = min((item / nstrips) * PPW + pipe, D - 1);
    const int w0 = strip * OUTW, h0 = chunk * nrows, h1 = min(h0 + nrows, H);
    const int ys = max(h0 - R, 0), ye = min(h1 - 1 + R, H - 1);
    const int nb = (h1 - 1 + R - ys + B) / B;   // batch k holds rows ys + k*B + (0..B-1)
    const size_t plane = (size_t)H * W;
    const int rowv = 4 * W;
    const uint32_t lane8 = 8u * (uint32_t)lane;
    const char *supb = reinterpret_cast<const char *>(sup);

    // Iterations: T = NPF*n4 + 2, n4 = ceil(nb / NPF).  Every wave runs a two-iteration prologue and then n4 groups of
    // NPF straight-line iterations (no exits, no guards: the compiler's s_waitcnt vmcnt counts stay exact, which is
    // what keeps NPF batches of loads in flight).  Batches past nb are padding: their rows are clamped loads, their
    // ring rows are never referenced and their stores are dropped.
    const int n4 = (nb + NPF - 1) / NPF;

    if (wave < NSCAN) {
        // ---------------- scan: rows b0 .. b0+SR-1 of batch t ----------------
        const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(in + (size_t)d * plane), 0, (int)(plane * 4), 0x00020000);
        const int b0 = wave * SR;
        const int x0 = w0 - HL + CPS * lane;     // image column of this lane's first staged entry
        const int vb = 4 * (x0 + CPS <= 0 ? 0 : x0);   // lanes entirely left of the image read (and never publish) column 0..
        u32x4 vv[NPF][SR];
        uint32_t v5[NPF][SR];
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int j = 0; j < SR; ++j) {
                const int so = min(ys + k * B + b0 + j, ye) * rowv;
                vv[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, vb, so, 0);
                v5[slot][j] = __builtin_amdgcn_raw_buffer_load_b32(rs_src, vb + 16, so, 0);
            }
        };
        auto scan_batch = [&](int slot, int par, int t) {
            double p0[SR], p1[SR], p2[SR], p3[SR], tt[SR], ex[SR];
#pragma unroll
            for (int j = 0; j < SR; ++j) {
                p0[j] = (double)__uint_as_float(vv[slot][j].x);
                p1[j] = p0[j] + (double)__uint_as_float(vv[slot][j].y);
                p2[j] = p1[j] + (double)__uint_as_float(vv[slot][j].z);
                p3[j] = p2[j] + (double)__uint_as_float(vv[slot][j].w);
                tt[j] = p3[j] + (double)__uint_as_float(v5[slot][j]);
            }
            // steps outermost so the rows' dependent chains interleave (a lone chain issues one VALU per ~5 cycles)
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x111>(tt[j]);        // row_shr:1
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x112>(tt[j]);        // row_shr:2
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x114>(tt[j]);        // row_shr:4
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x118>(tt[j]);        // row_shr:8
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x142, 0xA>(tt[j]);   // row_bcast:15
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x143, 0xC>(tt[j]);   // row_bcast:31
#pragma unroll
            for (int j = 0; j < SR; ++j) ex[j] = dpp_f64<0x138>(tt[j]);         // wave_shr:1 -> exclusive
#pragma unroll
            for (int j = 0; j < SR; ++j) {
                char *pr = ldsb + (par * B + b0 + j) * PROWB + lane8;
                *reinterpret_cast<double *>(pr + 0 * SUBB) = ex[j] + p0[j];
                *reinterpret_cast<double *>(pr + 1 * SUBB) = ex[j] + p1[j];
                *reinterpret_cast<double *>(pr + 2 * SUBB) = ex[j] + p2[j];
                *reinterpret_cast<double *>(pr + 3 * SUBB) = ex[j] + p3[j];
                *reinterpret_cast<double *>(pr + 4 * SUBB) = tt[j];
            }
            issue(slot, t + NPF);
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        scan_batch(0, 0, 0);
        pipe_sync();
        scan_batch(1 % NPF, 1, 1);
        pipe_sync();
        for (int g = 0; g < n4; ++g) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                scan_batch((u + 2) % NPF, u & 1, 2 + g * NPF + u);
                pipe_sync();
            }
        }
    } else if (wave < NSCAN + NHS) {
        // ---------------- hsum: batch t-1, columns j0 .. j0+CH-1 of every lane ----------------
        const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(supb) + hsum_plane_offset(H, W), 0, (int)(plane * 4), 0x00020000);
        constexpr int CH = CPL / NHS;
        const int j0 = (wave - NSCAN) * CH;
        const int sb = 4 * (w0 + CPL * lane + j0);
        double q[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) q[j] = 0.0;
        uint32_t sy[NPF][B][CH];
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int so = min(ys + k * B + b, ye) * rowv;
                if constexpr (CH == 4) {
                    const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rs_hs, sb, so, 0);
                    sy[slot][b][0] = t4.x, sy[slot][b][1] = t4.y, sy[slot][b][2 % CH] = t4.z, sy[slot][b][3 % CH] = t4.w;
                } else {
                    const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(rs_hs, sb, so, 0);
                    sy[slot][b][0] = t2.x, sy[slot][b][1] = t2.y;
                }
            }
        };
        auto hsum_batch = [&](int slot, int par, int k) {
            double hs[B][CH];
            {
                double pa[B][CH], pb[B][CH];     // all prow reads in flight before the first use
                const char *prb = ldsb + par * (B * PROWB);
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const uint32_t wd = sy[slot][b][j];
                        pa[b][j] = *reinterpret_cast<const double *>(prb + b * PROWB + (wd & 0xffffu));
                        pb[b][j] = *reinterpret_cast<const double *>(prb + b * PROWB + (wd >> 16));
                    }
#pragma unroll
                for (int b = 0; b < B; ++b)
#pragma unroll
                    for (int j = 0; j < CH; ++j) hs[b][j] = pa[b][j] - pb[b][j];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                // rows past the image bottom are staged from clamped loads and never referenced by an arm:
                // they run through unconditionally into ring rows nobody reads
                char *rr = ldsb + PROW_BYTES + ((k * B + b) & (RING - 1)) * ROWB + j0 * SUBB + lane8;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    q[j] += hs[b][j];
                    *reinterpret_cast<double *>(rr + j * SUBB) = q[j];
                }
            }
            issue(slot, k + NPF);
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        {   // Q of the row above the first staged row (relative row -1 = ring slot 31) is zero
            char *z = ldsb + PROW_BYTES + (RING - 1) * ROWB + j0 * SUBB + lane8;
#pragma unroll
            for (int j = 0; j < CH; ++j) *reinterpret_cast<double *>(z + j * SUBB) = 0.0;
        }
        pipe_sync();
        hsum_batch(0, 0, 0);
        pipe_sync();
        for (int g = 0; g < n4; ++g) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                hsum_batch((u + 1) % NPF, (u + 1) & 1, 1 + g * NPF + u);
                pipe_sync();
            }
        }
    } else {
        // ---------------- emit: rows b0 .. b0+ER-1 of batch t-2 ----------------
        const __amdgpu_buffer_rsrc_t rs_ew = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(supb) + emit_plane_offset(H, W), 0, (int)(plane * 8), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dst =
            __builtin_amdgcn_make_buffer_rsrc(out + (size_t)d * plane, 0, (int)(plane * 4), 0x00020000);
        const int b0 = (wave - NSCAN - NHS) * ER;
        const int c0 = w0 + CPL * lane;
        constexpr int kDrop = 0x7ffffff0;        // byte offset past every plane: the range check drops the store
        const int nval = min(max(W - c0, 0), CPL);   // valid output columns of this lane
        const int ob = 4 * c0;
        const int obf = nval == CPL ? ob : kDrop;
        const bool ragged = __builtin_amdgcn_readfirstlane((W & (CPL - 1)) != 0 && w0 + OUTW > W);
        const int eb = 8 * c0;
        const char *ringb = ldsb + PROW_BYTES;
        u32x4 ew[NPF][ER][2];                    // emit words of 4 pixels: {lo0, hi0, lo1, hi1}, {lo2, hi2, lo3, hi3}
        double qb[ER][CPL];                      // "below" lookups of the batch emitted in the next iteration
        auto issue = [&](int slot, int k) {
#pragma unroll
            for (int b = 0; b < ER; ++b) {
                const int so = min(max(ys + k * B + b0 + b - R, h0), h1 - 1) * (2 * rowv);
                ew[slot][b][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_ew, eb, so, 0);
                ew[slot][b][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_ew, eb + 16, so, 0);
            }
        };
        // ring lookups of batch k: sh = 4 selects the "down" field ((y + down) mod 32), sh = 9 the "31 - up" field
        auto lookups = [&](int slot, int k, int sh, double (&dst)[ER][CPL]) {
#pragma unroll
            for (int b = 0; b < ER; ++b) {
                uint32_t yms = (uint32_t)(k * B + b0 + b - R) << 11;   // bits >= 16 fall to the mask
                asm volatile("" : "+s"(yms));    // one SGPR per row (else the constant is added per lane)
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const uint32_t lo = ew[slot][b][j >> 1][(j & 1) * 2];
                    const uint32_t ad = (((lo << sh) + yms) & 0xF800u) | lane8;
                    dst[b][j] = *reinterpret_cast<const double *>(ringb + ad + j * SUBB);
                }
            }
        };
        auto emit_batch = [&](int sa, int sn, int ka) {
            double qa[ER][CPL], qn[ER][CPL];
            lookups(sa, ka, 4, qa);
            lookups(sn, ka + 1, 9, qn);
#pragma unroll
            for (int b = 0; b < ER; ++b) {
                const int yo = ys + ka * B + b0 + b - R;
                const bool valid = yo >= h0 && yo < h1;      // wave-uniform
                const int so = min(max(yo, h0), h1 - 1) * rowv;
                u32x4 o;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const uint32_t lo = ew[sa][b][j >> 1][(j & 1) * 2], hi = ew[sa][b][j >> 1][(j & 1) * 2 + 1];
                    const double rn = __hiloint2double((int)hi, (int)lo);
                    o[j] = __float_as_uint((float)((qa[b][j] - qb[b][j]) * rn));
                }
                buffer_store_b128<0>(o, rs_dst, valid ? obf : kDrop, so);
                if (ragged && valid && nval > 0 && nval < CPL) {   // the one lane that straddles the right edge
                    if (nval >= 2) {
                        u32x2 o2 = {o.x, o.y};
                        __builtin_amdgcn_raw_buffer_store_b64(o2, rs_dst, ob, so, 0);
                        if (nval == 3) __builtin_amdgcn_raw_buffer_store_b32(o.z, rs_dst, ob + 8, so, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(o.x, rs_dst, ob, so, 0);
                    }
                }
            }
            issue(sa, ka + NPF);
#pragma unroll
            for (int b = 0; b < ER; ++b)
#pragma unroll
                for (int j = 0; j < CPL; ++j) qb[b][j] = qn[b][j];
        };
#pragma unroll
        for (int k = 0; k < NPF; ++k) issue(k, k);
        pipe_sync();
        lookups(0, 0, 9, qb);                    // iteration 1: "below" lookups of batch 0 (rows < its first row)
        pipe_sync();
        for (int g = 0; g < n4; ++g) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                emit_batch(u, (u + 1) % NPF, g * NPF + u);
                pipe_sync();
            }
        }
    }
}

template <int NSCAN, int NEMIT>
static int launch_cbca_stream(const StreamJobs &jobs, int D, int H, int W, hipStream_t s)
{
    const int nstrips = cdiv(W, s4::OUTW);
    // One workgroup per CU is resident (two planes, 158 KiB of LDS), and every (strip, plane group) costs the same, so
    // the launch is a number of rounds.  items / slots is rarely whole: the whole rounds run at full height, and the
    // remainder is cut into row chunks (each re-stages 2R halo rows, never shorter than 64 rows) so that it fills the
    // chip too - e.g. 750x500x256: 384 items on 256 CUs = 256 full-height workgroups + 128 items in two halves,
    // 526 + 276 staged rows per CU instead of 2 x 526 (one chunk) or 3 x 276 (everything in halves).
    const int slots = (2 / PPW) * device_cus8();   // groups of 8: the two parts of the launch keep their XCD phase
    const int DG = cdiv(D, PPW);                 // plane groups
    const long items = (long)nstrips * DG * jobs.n;
    const long nfull = slots > 0 ? items / slots * slots : 0;
    const long rem = items - nfull;
    int nchunks = 1;
    if (rem > 0) {
        double best = 1e30;
        for (int n = 1; n <= max(1, min(H / 64, 16)); ++n) {
            const double cost = ceil((double)rem * n / slots) * ((double)H / n + 2 * s4::R);   // staged rows per CU
            if (cost < best - 1e-9) {
                best = cost;
                nchunks = n;
            }
        }
    }
    const int rows = cdiv(H, nchunks);
    const long total = nfull + rem * nchunks;
    MCCNN_REQUIRE(total <= 0x7fffffffL && (long)H * W * 8 <= 0x7fffffffL, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter: image %dx%d / volume too large for 32-bit buffer offsets", W, H);
    hipLaunchKernelGGL((cbca_stream_kernel<NSCAN, NEMIT>), dim3((unsigned)total),
                       dim3(64 * PPW * (NSCAN + NHS + NEMIT)), 0, s, jobs, D, H, W, rows, nstrips, nchunks,
                       (int)total, (int)nfull);
    return check_launch("mccnn_cbca_iter(stream)");
}

}  // namespace mccnn

// Host-side record of what mccnn_cross_arms last wrote where (pointer -> H, W, L), so that mccnn_cbca_iter can refuse a
// support plane built for another image size or with longer arms than the distance it is told (which selects the
// kernel and its halo).  Pointers it has never seen (e.g. a device-side copy) pass: the kernels are memory-safe for
// any arms (clamped above / baked within range), only the sums would be those of the clamped region.
namespace {
// gen: when the arms were written (what was derived from a buffer can tell if it is stale); used: the last time an entry
// point looked the buffer up - eviction goes by `used`, so a long-lived buffer that is in use is never dropped
struct SupportInfo { int H, W, L; unsigned long long gen; unsigned long long used; };
std::mutex g_support_mu;
std::unordered_map<const void *, SupportInfo> g_support;
unsigned long long g_support_gen = 0;     // counts the buffers mccnn_cross_arms has written (two per pair call)
unsigned long long g_support_tick = 0;    // counts registrations and look-ups
}  // namespace

unsigned long long mccnn::support_generation(const mccnn_support_t *support)
{
    std::lock_guard<std::mutex> lock(g_support_mu);
    const auto it = g_support.find(support);
    return it == g_support.end() ? 0ull : it->second.gen;
}

int mccnn::check_support_record(const mccnn_support_t *support, int H, int W, int L, const char *who, bool must_be_known)
{
    std::lock_guard<std::mutex> lock(g_support_mu);
    const auto it = g_support.find(support);
    if (it != g_support.end()) it->second.used = ++g_support_tick;
    MCCNN_REQUIRE(!must_be_known || it != g_support.end(), MCCNN_E_INVALID,
                  "%s: `support` is not a buffer mccnn_cross_arms has written (the kernel also reads the planes behind "
                  "plane 0: pass the whole mccnn_support_bytes(H, W) buffer, not a copy of its first plane)", who);
    if (it != g_support.end()) {
        MCCNN_REQUIRE(it->second.H == H && it->second.W == W, MCCNN_E_INVALID,
                      "%s: support plane was built for a %dx%d image, volume is %dx%d", who, it->second.W, it->second.H,
                      W, H);
        MCCNN_REQUIRE(it->second.L <= L, MCCNN_E_INVALID,
                      "%s: support plane was built with distance %d, called with L=%d", who, it->second.L, L);
    }
    return 0;
}

static int launch_cross_arms(const float *img0, const float *img1, mccnn_support_t *sup0, mccnn_support_t *sup1, int n,
                             int H, int W, float tau, int L, hipStream_t s)
{
    using namespace mccnn;
    const dim3 grid(cdiv(W, 256), H, n), block(256);
    hipLaunchKernelGGL(cross_arms_kernel, grid, block, 0, s, img0, img1, H, W, tau, L, sup0, sup1);
    int rc = check_launch("mccnn_cross_arms");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_count_kernel, grid, block, 0, s, sup0, sup1, H, W);
    rc = check_launch("mccnn_cross_arms(count)");
    if (rc) return rc;
    hipLaunchKernelGGL(cross_perm_kernel, dim3(cdiv(W, CB_TW_C), cdiv(H, PERM_TH), n), block, 0, s, sup0, sup1, H, W);
    rc = check_launch("mccnn_cross_arms(order)");
    if (rc == 0) {
        std::lock_guard<std::mutex> lock(g_support_mu);
        // The registry is authoritative (the *_hwd / *_prog entry points refuse buffers it does not know), so it is never
        // emptied: when it outgrows its bound the LEAST RECENTLY USED half goes (every look-up by an entry point
        // refreshes an entry, so a buffer that is still being aggregated with stays however old its arms are).
        if (g_support.size() > 4096) {
            std::vector<unsigned long long> used;
            used.reserve(g_support.size());
            for (const auto &kv : g_support) used.push_back(kv.second.used);
            std::nth_element(used.begin(), used.begin() + used.size() / 2, used.end());
            const unsigned long long keep_from = used[used.size() / 2];
            for (auto it = g_support.begin(); it != g_support.end();)
                it = it->second.used < keep_from ? g_support.erase(it) : std::next(it);
        }
        g_support[sup0] = SupportInfo{H, W, L, ++g_support_gen, ++g_support_tick};
        g_support[sup1] = SupportInfo{H, W, L, ++g_support_gen, ++g_support_tick};
    }
    return rc;
}

extern "C" size_t mccnn_support_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    return mccnn::wmask_plane_offset(H, W) + mccnn::wmask_plane_bytes(H, W);
}

extern "C" int mccnn_cross_arms(const float *image, int H, int W, float tau, int L, mccnn_support_t *support,
                                mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image && support, MCCNN_E_INVALID, "mccnn_cross_arms: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cross_arms: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 32, MCCNN_E_UNSUPPORTED,
                  "mccnn_cross_arms: L=%d outside [1,32] (5-bit arms, 12-bit region size)", L);
    return launch_cross_arms(image, image, support, support, 1, H, W, tau, L, (hipStream_t)stream);
}

extern "C" int mccnn_cross_arms_pair(const float *image_left, const float *image_right, int H, int W, float tau, int L,
                                     mccnn_support_t *support_left, mccnn_support_t *support_right,
                                     mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image_left && image_right && support_left && support_right, MCCNN_E_INVALID,
                  "mccnn_cross_arms_pair: null pointer");
    MCCNN_REQUIRE(support_left != support_right, MCCNN_E_INVALID, "mccnn_cross_arms_pair: the two buffers must differ");
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cross_arms_pair: non-positive size");
    MCCNN_REQUIRE(L >= 1 && L <= 32, MCCNN_E_UNSUPPORTED,
                  "mccnn_cross_arms_pair: L=%d outside [1,32] (5-bit arms, 12-bit region size)", L);
    return launch_cross_arms(image_left, image_right, support_left, support_right, 2, H, W, tau, L, (hipStream_t)stream);
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

extern "C" int mccnn_cbca_iter_both(const float *in, float *out, const mccnn_support_t *support_self,
                                    const mccnn_support_t *support_other, int D, int H, int W, int L, int side,
                                    mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in && out && support_self && support_other, MCCNN_E_INVALID, "mccnn_cbca_iter_both: null pointer");
    MCCNN_REQUIRE(in != out, MCCNN_E_INVALID, "mccnn_cbca_iter_both: in-place aggregation is not defined (ping-pong)");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_cbca_iter_both: non-positive size");
    MCCNN_REQUIRE(D <= 65535, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter_both: D=%d exceeds grid.z", D);
    MCCNN_REQUIRE(side == MCCNN_SIDE_LEFT || side == MCCNN_SIDE_RIGHT, MCCNN_E_INVALID,
                  "mccnn_cbca_iter_both: side %d", side);
    MCCNN_REQUIRE(L >= 1 && L <= 32, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter_both: L=%d outside [1,32]", L);
    const int dsign = side == MCCNN_SIDE_LEFT ? -1 : 1;
    hipStream_t s = (hipStream_t)stream;
    if (L <= 14) {
        const dim3 grid(cdiv(W, CB_TW), cdiv(H, 32), D);
        hipLaunchKernelGGL((cbca_both_views_kernel<13, 32>), grid, dim3(256), 0, s, in, out, support_self, support_other,
                           H, W, dsign);
    } else {
        const dim3 grid(cdiv(W, CB_TW), cdiv(H, 16), D);
        hipLaunchKernelGGL((cbca_both_views_kernel<31, 16>), grid, dim3(256), 0, s, in, out, support_self, support_other,
                           H, W, dsign);
    }
    return check_launch("mccnn_cbca_iter_both");
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
    if (const int rc = check_support_record(support, H, W, L, "mccnn_cbca_iter")) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (order == MCCNN_CBCA_SEPARABLE && L <= 14) {
        const StreamJobs jobs = {{in, nullptr}, {out, nullptr}, {support, nullptr}, 1};
        return launch_cbca_stream<CBCA_NSCAN, CBCA_NEMIT>(jobs, D, H, W, s);
    }
    if (L <= 14) return launch_cbca<13, 32>(in, out, support, D, H, W, order, s);
    if (L <= 32) return launch_cbca<31, 16>(in, out, support, D, H, W, order, s);
    MCCNN_REQUIRE(false, MCCNN_E_UNSUPPORTED, "mccnn_cbca_iter: L=%d > 32 not built", L);
}

extern "C" int mccnn_cbca_iter_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                    const float *in_right, float *out_right, const mccnn_support_t *support_right,
                                    int D, int H, int W, int L, int order, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(in_left && out_left && support_left && in_right && out_right && support_right, MCCNN_E_INVALID,
                  "mccnn_cbca_iter_pair: null pointer");
    MCCNN_REQUIRE(in_left != out_left && in_right != out_right && out_left != out_right && in_left != out_right &&
                      in_right != out_left,
                  MCCNN_E_INVALID, "mccnn_cbca_iter_pair: outputs must not alias an input or each other");
    if (order == MCCNN_CBCA_SEPARABLE && L >= 1 && L <= 14 && D > 0 && H > 0 && W > 0 && D <= 65535) {
        int rc = check_support_record(support_left, H, W, L, "mccnn_cbca_iter_pair");
        if (rc) return rc;
        rc = check_support_record(support_right, H, W, L, "mccnn_cbca_iter_pair");
        if (rc) return rc;
        const StreamJobs jobs = {{in_left, in_right}, {out_left, out_right}, {support_left, support_right}, 2};
        return launch_cbca_stream<CBCA_NSCAN, CBCA_NEMIT>(jobs, D, H, W, (hipStream_t)stream);
    }
    // every other variant: the two single-volume launches, with their argument checks
    const int rc = mccnn_cbca_iter(in_left, out_left, support_left, D, H, W, L, order, stream);
    if (rc) return rc;
    return mccnn_cbca_iter(in_right, out_right, support_right, D, H, W, L, order, stream);
}
