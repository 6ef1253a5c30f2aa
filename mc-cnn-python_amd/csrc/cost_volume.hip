// a2  compute_cost_volume (/root/reference/src/process_functional.py:78-113) on gfx950.
//
//   lcv[d,h,w] = -<fl[h,w,:], fr[h,w-d,:]>            w >= d          (pf:87-91, 111)
//   lcv[d,h,w] = mean(lcv[d,h,w+1..w+3])               w <  d, right-to-left recurrence   (pf:94-95)
//   rcv[d,h,w] = lcv[d,h,w+d]                          w <  W-d        (pf:103-104)
//   rcv[d,h,w] = mean(rcv[d,h,w-3..w-1])               w >= W-d, left-to-right recurrence (pf:105-106)
//
// The score matrix of one image row, S[w,w'] = <fl[w], fr[w']>, is a banded product (0 <= w-w' < D); every
// entry feeds lcv[d,h,w] and rcv[d,h,w'] (d = w-w'), both contiguous in w for fixed d, so a workgroup that owns
// 64 columns x 64 disparities of one row writes 256-B runs into both volumes.  Bound: HBM writes (8 B/voxel for
// 128 flop/voxel).  The recurrences run on the already negated values: rounding is sign-symmetric, so
// -((0 - x1 - x2 - x3) / 3) on them has the bits of the reference's fill-then-negate, the sign of an exactly-zero sum
// included (round 6: a plain 0 + x1 + x2 + x3 gave +0.0 there where the reference has -0.0).
//
// MCCNN_CV_EXACT  : VALU kernel that reproduces NumPy's float32 pairwise summation of the 64 products
//                   (8 running sums, fixed combine tree - numpy loops_utils pairwise_sum) bit for bit.
// MCCNN_CV_MFMA   : the 64-channel contraction on the matrix cores, v_mfma_f32_32x32x2_f32 (exact-f32 fma chain in
//                   channel order); differs from NumPy's order by <= 2e-6 on unit-norm features.
#include "common.h"

namespace mccnn {

constexpr int CV_TW = 64;   // output columns per workgroup
constexpr int CV_DT = 64;   // disparities per workgroup
constexpr int CV_C = 64;    // feature channels (NET num_conv_feature_maps)
#ifndef CVM_WAVES_PER_SIMD
#define CVM_WAVES_PER_SIMD 6
#endif
constexpr int CV_LD = 68;   // padded LDS row (floats): 16-B slot index = (row + c4) mod 16 -> conflict-free b128

// PIXEL_MAJOR: the volumes are written "HWD" [H][W][Dp] (the layout the whole bit-exact variant works on) instead of
// [D][H][W]: the 64 x 64 scores of the tile go through LDS (over the left-feature tile, which lives in registers by
// then) - a left-volume pixel receives its 64 disparities as one 256-byte run, a right-volume pixel x = w - d receives
// the anti-diagonal of the tile that belongs to it (up to 64 disparities, one dword per lane).  Entries with w < d
// (left) and x >= W - d (right) are left to cost_volume_fill_hwd_kernel.
template <bool PIXEL_MAJOR>
__global__ __launch_bounds__(256) void cost_volume_exact_kernel(const float *__restrict__ fl,
                                                                const float *__restrict__ fr, int H, int W, int D,
                                                                float *__restrict__ lcv, float *__restrict__ rcv, int Dp)
{
    __shared__ __attribute__((aligned(16))) float sR[(CV_TW + CV_DT - 1) * CV_LD];
    __shared__ __attribute__((aligned(16))) float sL[CV_TW * CV_LD];
    const int w0 = blockIdx.x * CV_TW, h = blockIdx.y, d0 = blockIdx.z * CV_DT;
    if (w0 + CV_TW - 1 < d0) return;  // every (w, d) of this tile has w < d: border fill handles it
    const int tid = threadIdx.x;
    const size_t rowbase = (size_t)h * W;
    for (int i = tid; i < CV_TW * 16; i += 256) {
        const int px = i >> 4, c4 = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w0 + px < W) v = *reinterpret_cast<const float4 *>(fl + (rowbase + w0 + px) * CV_C + c4 * 4);
        *reinterpret_cast<float4 *>(&sL[px * CV_LD + c4 * 4]) = v;
    }
    const int xr0 = w0 - d0 - (CV_DT - 1);  // right-image column held in sR row 0
    for (int i = tid; i < (CV_TW + CV_DT - 1) * 16; i += 256) {
        const int r = i >> 4, c4 = i & 15;
        const int x = xr0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 0 && x < W) v = *reinterpret_cast<const float4 *>(fr + (rowbase + x) * CV_C + c4 * 4);
        *reinterpret_cast<float4 *>(&sR[r * CV_LD + c4 * 4]) = v;
    }
    __syncthreads();

    const int wl = tid & 63, dq = tid >> 6;  // dq is wave-uniform: each wave owns 16 disparities
    const int w = w0 + wl;
    float a[CV_C];
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        const float4 v = *reinterpret_cast<const float4 *>(&sL[wl * CV_LD + c4 * 4]);
        a[c4 * 4 + 0] = v.x; a[c4 * 4 + 1] = v.y; a[c4 * 4 + 2] = v.z; a[c4 * 4 + 3] = v.w;
    }
    const size_t plane = (size_t)H * W;
    constexpr int TP = 65;              // pitch of the score tile (PIXEL_MAJOR)
    float *const sT = sL;               // every thread holds its left features in registers from here on
    if (PIXEL_MAJOR) __syncthreads();
    for (int k = 0; k < 16; ++k) {
        const int d = d0 + dq * 16 + k;
        if (d >= D) break;
        const int r = wl + (CV_DT - 1) - (dq * 16 + k);  // sR row of right column w - d
        if (w < W && w >= d) {
            const float *b = &sR[r * CV_LD];
            float acc[8];
            // numpy pairwise_sum, n = 64: r[j] = p[j]; r[j] += p[8i+j]; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 b0 = *reinterpret_cast<const float4 *>(b + i * 8);
                const float4 b1 = *reinterpret_cast<const float4 *>(b + i * 8 + 4);
                const float p0 = a[i * 8 + 0] * b0.x, p1 = a[i * 8 + 1] * b0.y, p2 = a[i * 8 + 2] * b0.z,
                            p3 = a[i * 8 + 3] * b0.w, p4 = a[i * 8 + 4] * b1.x, p5 = a[i * 8 + 5] * b1.y,
                            p6 = a[i * 8 + 6] * b1.z, p7 = a[i * 8 + 7] * b1.w;
                if (i == 0) {
                    acc[0] = p0; acc[1] = p1; acc[2] = p2; acc[3] = p3;
                    acc[4] = p4; acc[5] = p5; acc[6] = p6; acc[7] = p7;
                } else {
                    acc[0] += p0; acc[1] += p1; acc[2] += p2; acc[3] += p3;
                    acc[4] += p4; acc[5] += p5; acc[6] += p6; acc[7] += p7;
                }
            }
            float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            s = 0.f + s;  // np.sum adds the pairwise result to the identity
            s = -1.f * s;
            if (PIXEL_MAJOR) {
                sT[wl * TP + dq * 16 + k] = s;
            } else {
                lcv[(size_t)d * plane + rowbase + w] = s;
                rcv[(size_t)d * plane + rowbase + (w - d)] = s;
            }
        }
    }
    if (PIXEL_MAJOR) {
        __syncthreads();
        const int nd = min(CV_DT, D - d0);           // disparities of this tile that exist
        // left volume: pixel w0 + px, disparities d0 .. d0 + nd - 1 (those with d <= w), 64 lanes = 64 disparities
        const int dl = tid & 63;
        for (int px = tid >> 6; px < CV_TW; px += 4) {
            const int ww = w0 + px, d = d0 + dl;
            if (ww < W && dl < nd && ww >= d) lcv[(rowbase + ww) * (size_t)Dp + d] = sT[px * TP + dl];
        }
        // right volume: pixel x = w - d; its entries of this tile are (w = x + d, d), d0 <= d < d0 + nd
        for (int xi = tid >> 6; xi < CV_TW + CV_DT - 1; xi += 4) {
            const int x = xr0 + xi, d = d0 + dl, px = x + d - w0;      // tile row of w = x + d
            if (x >= 0 && dl < nd && px >= 0 && px < CV_TW && w0 + px < W)
                rcv[(rowbase + x) * (size_t)Dp + d] = sT[px * TP + dl];
        }
    }
}

// ---- bit-exact products for the pixel-major volumes, two scores per LDS row (round 4) ---------------------------------
// cost_volume_exact_kernel<true> is bound by its LDS reads: every score fetches the 64 right-image features of its own
// (w, d) - 256 bytes - out of LDS.  But S(w, d + 1) needs the row of right pixel w - d - 1, which is exactly the row
// lane w - 1 has just fetched for S(w - 1, d): here every lane fetches ONE row per pair of disparities and multiplies
// it twice - with its own left features for d, and, read across the lane boundary by the DPP operand modifier
// (wave_shr:1: lane l reads lane l - 1's register, no extra instruction), for d + 1.  Half the LDS bytes per score; the
// arithmetic and its order per score are unchanged (NumPy's float32 pairwise sum of the 64 rounded products, pf:87-91):
// bit-identical.  Lane 0 has no left neighbour: it is a ghost lane that carries the pixel in front of the tile (the
// last pixel of the tile to the left, which computes that pixel's scores itself), so a tile is 63 pixels wide.  The
// compute is unconditional - a lane's rows must exist for its neighbour - and only the hand-over to the score tile is
// predicated.
constexpr int CVP_TW = 63;   // new pixels per tile (lanes 1 .. 63)
#ifndef CVP_ABL
#define CVP_ABL 0            // timing-only ablations (wrong results): 1 no products, 2 no volume stores
#endif
// (an ablated part stays in the code, behind a condition no real call meets, so that registers and LDS do not change)
__device__ __forceinline__ bool cvp_never(int D) { return D == 12345; }
__global__ __launch_bounds__(256, 3) void cost_volume_exact_pairs_kernel(const float *__restrict__ fl,
                                                                         const float *__restrict__ fr, int H, int W, int D,
                                                                         float *__restrict__ lcv, float *__restrict__ rcv,
                                                                         int Dp)
{
    // A workgroup owns 63 new pixels of one image row and walks ALL their disparity tiles: the left features stay in
    // registers, and the right-image rows live in a ring of 128 LDS rows indexed by (column & 127) - a tile of 64
    // disparities needs the 127 columns w0 - d0 - 63 .. w0 - d0 + 63, of which the next tile reuses 63: only 64 new
    // rows are fetched per tile (into registers before the tile's products, into the ring behind them).
    constexpr int RING = 128;
    __shared__ __attribute__((aligned(16))) float sR[RING * CV_LD];
    __shared__ __attribute__((aligned(16))) float sL[CV_TW * CV_LD];
    const int w0 = (int)blockIdx.x * CVP_TW - 1, h = blockIdx.y;   // lane 0: pixel w0 (ghost)
    const int tid = threadIdx.x;
    const size_t rowbase = (size_t)h * W;
    for (int i = tid; i < CV_TW * 16; i += 256) {
        const int px = i >> 4, c4 = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w0 + px >= 0 && w0 + px < W) v = *reinterpret_cast<const float4 *>(fl + (rowbase + w0 + px) * CV_C + c4 * 4);
        *reinterpret_cast<float4 *>(&sL[px * CV_LD + c4 * 4]) = v;
    }
    // the first tile's columns w0 - 63 .. w0 + 63
    for (int i = tid; i < (CV_TW + CV_DT - 1) * 16; i += 256) {
        const int x = w0 - (CV_DT - 1) + (i >> 4), c4 = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x >= 0 && x < W) v = *reinterpret_cast<const float4 *>(fr + (rowbase + x) * CV_C + c4 * 4);
        *reinterpret_cast<float4 *>(&sR[(x & (RING - 1)) * CV_LD + c4 * 4]) = v;
    }
    __syncthreads();

    const int wl = tid & 63, dq = tid >> 6;  // dq is wave-uniform: each wave owns 16 disparities = 8 pairs of a tile
    const int w = w0 + wl;
    float a[CV_C];
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        const float4 v = *reinterpret_cast<const float4 *>(&sL[wl * CV_LD + c4 * 4]);
        a[c4 * 4 + 0] = v.x; a[c4 * 4 + 1] = v.y; a[c4 * 4 + 2] = v.z; a[c4 * 4 + 3] = v.w;
    }
    constexpr int TP = CV_LD;           // pitch of the score tile: a multiple of four, so a pixel's disparities leave as float4
    float *const sT = sL;               // every thread holds its left features in registers from here on
    __syncthreads();
#pragma unroll 1
    for (int d0 = 0; d0 < D && w0 + CV_TW - 1 >= d0; d0 += CV_DT) {   // beyond: every (w, d) has w < d (border fill)
        // the next tile's 64 new columns w0 - d0 - 127 .. w0 - d0 - 64 on their way (4 x 16 bytes per thread)
        const bool more = d0 + CV_DT < D && w0 + CV_TW - 1 >= d0 + CV_DT;
        float4 nx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, x = w0 - d0 - (2 * CV_DT - 1) + (i >> 4), c4 = i & 15;
            nx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (more && x >= 0 && x < W) nx[q] = *reinterpret_cast<const float4 *>(fr + (rowbase + x) * CV_C + c4 * 4);
        }
#pragma unroll 1
        for (int kk = 0; kk < (((CVP_ABL & 1) && !cvp_never(D)) ? 0 : 8); ++kk) {
            const int dloc = dq * 16 + 2 * kk;                   // the even disparity of the pair (tile-relative)
            const int r = (w - d0 - dloc) & (RING - 1);          // ring row of right column w - d_even
            const float *b = &sR[r * CV_LD];
            float acc[8], acn[8];                                // this lane's score for d_even; for d_even + 1 (neighbour's row)
            // numpy pairwise_sum, n = 64: r[j] = p[j]; r[j] += p[8i+j]; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
            // eight channels at a time, the next eight fetched under them
            float4 n0 = *reinterpret_cast<const float4 *>(b), n1 = *reinterpret_cast<const float4 *>(b + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float bv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
                if (i < 7) {
                    n0 = *reinterpret_cast<const float4 *>(b + (i + 1) * 8);
                    n1 = *reinterpret_cast<const float4 *>(b + (i + 1) * 8 + 4);
                }
                // the float32 operations of four channels as one block of instructions, independent ones next to each
                // other (the compiler otherwise defers one half of them to the end of the row and keeps - or spills -
                // everything they need until then)
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    if (i == 0) {
                        asm volatile("v_mul_f32 %0, %8, %12\n\tv_mul_f32 %1, %9, %13\n\tv_mul_f32 %2, %10, %14\n\tv_mul_f32 %3, %11, %15\n\t"
                                     "v_mul_f32_dpp %4, %12, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %5, %13, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %6, %14, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %7, %15, %11 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                     : "=&v"(acc[j]), "=&v"(acc[j + 1]), "=&v"(acc[j + 2]), "=&v"(acc[j + 3]),
                                       "=&v"(acn[j]), "=&v"(acn[j + 1]), "=&v"(acn[j + 2]), "=&v"(acn[j + 3])
                                     : "v"(a[j]), "v"(a[j + 1]), "v"(a[j + 2]), "v"(a[j + 3]),
                                       "v"(bv[j]), "v"(bv[j + 1]), "v"(bv[j + 2]), "v"(bv[j + 3]));
                    } else {
                        float t0, t1, t2, t3;
                        asm volatile("v_mul_f32 %8, %12, %16\n\tv_mul_f32 %9, %13, %17\n\tv_mul_f32 %10, %14, %18\n\tv_mul_f32 %11, %15, %19\n\t"
                                     "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"
                                     "v_mul_f32_dpp %8, %16, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %9, %17, %13 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %10, %18, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mul_f32_dpp %11, %19, %15 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %9\n\tv_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11"
                                     : "+v"(acc[j]), "+v"(acc[j + 1]), "+v"(acc[j + 2]), "+v"(acc[j + 3]),
                                       "+v"(acn[j]), "+v"(acn[j + 1]), "+v"(acn[j + 2]), "+v"(acn[j + 3]),
                                       "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                     : "v"(a[i * 8 + j]), "v"(a[i * 8 + j + 1]), "v"(a[i * 8 + j + 2]), "v"(a[i * 8 + j + 3]),
                                       "v"(bv[j]), "v"(bv[j + 1]), "v"(bv[j + 2]), "v"(bv[j + 3]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            float sn = ((acn[0] + acn[1]) + (acn[2] + acn[3])) + ((acn[4] + acn[5]) + (acn[6] + acn[7]));
            s = 0.f + s;  // np.sum adds the pairwise result to the identity
            sn = 0.f + sn;
            s = -1.f * s;
            sn = -1.f * sn;
            const int d = d0 + dloc;
            if (wl >= 1 && w < W) {
                if (d < D && w >= d) sT[wl * TP + dloc] = s;
                if (d + 1 < D && w >= d + 1) sT[wl * TP + dloc + 1] = sn;
            }
        }
        __syncthreads();                             // products done: the score tile is complete, the oldest ring rows are free
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, x = w0 - d0 - (2 * CV_DT - 1) + (i >> 4), c4 = i & 15;
            if (more) *reinterpret_cast<float4 *>(&sR[(x & (RING - 1)) * CV_LD + c4 * 4]) = nx[q];
        }
        const int nd = ((CVP_ABL & 2) && !cvp_never(D)) ? 0 : min(CV_DT, D - d0);   // disparities of this tile that exist
        const int xr0 = w0 - d0 - (CV_DT - 1);
        // Both volumes leave as 16-byte stores: 16 lanes x four disparities = one pixel's 64 disparities of this tile,
        // four pixels per wave instruction (a quarter of the store instructions of the one-float-per-lane form: the
        // store phase is bound by their count).  A group of four that straddles an edge (w = d, the last disparity, the
        // image border) goes out float by float.
        const int q4 = (tid & 15) * 4, sub = tid >> 4;
        // left volume: pixel w0 + px (px >= 1), disparities d0 .. d0 + nd - 1 (those with d <= w)
        for (int px = 1 + sub; px < CV_TW; px += 16) {
            const int ww = w0 + px, d = d0 + q4;
            if (ww >= W || q4 >= nd || ww < d) continue;
            const float4 v = *reinterpret_cast<const float4 *>(&sT[px * TP + q4]);
            float *dst = lcv + (rowbase + ww) * (size_t)Dp + d;
            if (q4 + 3 < nd && ww >= d + 3) {
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q4 + j < nd && ww >= d + j) dst[j] = e[j];
            }
        }
        // right volume: pixel x = w - d; its entries of this tile are (w = x + d, d), d0 <= d < d0 + nd: the score tile's
        // anti-diagonals (four LDS reads per 16-byte store)
        for (int xi = sub; xi < CV_TW + CV_DT - 1; xi += 16) {
            const int x = xr0 + xi, d = d0 + q4, px = x + d - w0;      // tile row of w = x + d (first of the four)
            if (x < 0 || q4 >= nd || px + 3 < 1 || px >= CV_TW) continue;
            float *dst = rcv + (rowbase + x) * (size_t)Dp + d;
            if (q4 + 3 < nd && px >= 1 && px + 3 < CV_TW && w0 + px + 3 < W) {
                float4 v;
                v.x = sT[px * TP + q4];
                v.y = sT[(px + 1) * TP + q4 + 1];
                v.z = sT[(px + 2) * TP + q4 + 2];
                v.w = sT[(px + 3) * TP + q4 + 3];
                *reinterpret_cast<float4 *>(dst) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q4 + j < nd && px + j >= 1 && px + j < CV_TW && w0 + px + j < W) dst[j] = sT[(px + j) * TP + q4 + j];
            }
        }
        __syncthreads();                             // the score tile has been read, the ring holds the next tile's rows
    }
}

// ---- MFMA variant ---------------------------------------------------------------------------------------------
// One wave computes a 32(w) x 32(w') block of S = <fl[w], fr[w']>; a workgroup of 4 waves covers 64 w x 64 w' and
// keeps only the band 0 <= w-w' < D.  A/B operands: lane l holds pixel l&31, channels 8 (l>>5) .. +7 of a 16-channel
// K step; C/D: col = lane&31 (w'), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (w).
//
// What bounds this kernel is neither the matrix cores nor the store pattern but how many workgroups a CU can hold:
// a workgroup is a chain of latencies (operand fetch, LDS staging, products, the transposition through LDS, and
// above all the drain of its stores before its LDS can be handed on), and the first version - float32-input MFMA,
// 34.5 KiB of LDS, 4 workgroups per CU - took the same 0.32 ms whether its 2048 matrix-core cycles per block were cut
// to 384, or its stores were rearranged into whole 512-byte runs (a 128 x 64 output-stationary tiling, 1.5 x the
// products, 121 KiB of LDS: 0.40 ms).  So this version is built for a small footprint instead:
//   * operands as two f16 numbers each (x * 2^10 = hi + lo; unit vectors, 22 significand bits), three
//     v_mfma_f32_32x32x16_f16 products per K step (hi*hi + hi*lo + lo*hi), float32 accumulation: <= 3e-7 from the exact
//     dot product, 384 matrix-core cycles per block;
//   * the 64 channels staged in two halves (18 KiB for both operand tiles; the second half waits in registers);
//   * the product tile T diagonal-major but folded: diagonal dd >= 0 (columns dd..63) and diagonal dd - 64 (columns
//     0..dd-1) share row dd of T[64][68] - 17 KiB instead of 34 - and the quads that straddle the fold are the ragged
//     ends of both diagonals, written element by element.
// 18 KiB of LDS: 8 workgroups per CU.  The accumulator is staged through LDS so that stores run along w for fixed d.
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 cv_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h4 __attribute__((ext_vector_type(4)));
constexpr int CVM_SP = 144;  // LDS bytes per operand pixel and channel half: [K step][hi 32 B | lo 32 B] + 16 pad
constexpr int CVM_TP = 68;   // LDS pitch of the folded product tile T[(w-x) & 63][w] (floats, 16-byte rows)

__device__ __forceinline__ void cv_split4(const float4 v, cv_h4 &hi, cv_h4 &lo)
{
    // unit vectors never get near the clamp (|x| <= 1 -> 1024); it keeps features that are not normalised finite
    const float x[4] = {__builtin_amdgcn_fmed3f(v.x * 1024.f, -65504.f, 65504.f),
                        __builtin_amdgcn_fmed3f(v.y * 1024.f, -65504.f, 65504.f),
                        __builtin_amdgcn_fmed3f(v.z * 1024.f, -65504.f, 65504.f),
                        __builtin_amdgcn_fmed3f(v.w * 1024.f, -65504.f, 65504.f)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)x[j];
        hi[j] = h;
        lo[j] = (_Float16)(x[j] - (float)h);
    }
}

// PIXEL_MAJOR (round 4): the product tile goes to LDS as it is, T[w][x], and leaves as whole disparity runs of the
// pixel-major volumes "HWD" [H][W][Dp] - a left pixel w takes its row of T (d = w - x: 64 consecutive disparities, 256
// bytes), a right pixel x its column - so that the matrix-core cost volume can feed the pixel-major pipeline without a
// layout change (mccnn_cost_volume_hwd with MCCNN_CV_MFMA).  Same products, same bits as the plane-major form.
template <bool PIXEL_MAJOR>
__global__ __launch_bounds__(256, CVM_WAVES_PER_SIMD) void cost_volume_mfma_kernel(
    const float *__restrict__ fl, const float *__restrict__ fr, int H, int W, int D, float *__restrict__ lcv,
    float *__restrict__ rcv, int nwt, int nbands, int total, int Dp)
{
    // operand tiles (one channel half at a time) and, after the products are done, the folded product tile share one
    // allocation
    constexpr int LDS_BYTES = (2 * 64 * CVM_SP > 64 * CVM_TP * 4) ? 2 * 64 * CVM_SP : 64 * CVM_TP * 4;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char *oL = lds, *oR = lds + 64 * CVM_SP;
    float *sT = reinterpret_cast<float *>(lds);
    // tile (bw, bx): left columns w0..w0+63, right columns x0..x0+63; d = w - x in (w0-x0-63 .. w0-x0+63)
    // Work order.  The pieces of one 128-byte line of a volume row come from neighbouring tiles (the next w tile, the
    // next band), and the dispatcher deals consecutive workgroups to the 8 XCDs round-robin - with the plain grid
    // order every L2 ended up holding partial lines and wrote them back partially (the store phase cost 0.28 ms of
    // 0.46).  So the launch is 1-D and each XCD gets a contiguous range of (row, band, w tile) work items: all
    // tiles of an image row meet in one L2 and lines leave it whole.  (Placement only affects speed.)
    int id;
    {
        const int b = blockIdx.x, qq = total >> 3, r = total & 7, xc = b & 7;
        id = (xc < r ? xc * (qq + 1) : r * (qq + 1) + (xc - r) * qq) + (b >> 3);
    }
    const int wt = id % nwt, band = (id / nwt) % nbands;
    const int h = id / (nwt * nbands);
    const int w0 = wt * 64;
    const int x0 = w0 - band * 64;  // band 0 -> x0 = w0, 1 -> x0 = w0-64, ...
    if (x0 + 63 < 0) return;
    if (w0 - x0 - 63 >= D) return;
    const int tid = threadIdx.x;
    const size_t rowbase = (size_t)h * W;
    // piece p = j*256 + tid of a channel half: pixel p >> 3, channels 4 (p & 7) .. + 3 of the half
    float4 vl[2][2], vr[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = j * 256 + tid, px = p >> 3, c4 = hf * 8 + (p & 7);
            vl[hf][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            vr[hf][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w0 + px < W) vl[hf][j] = *reinterpret_cast<const float4 *>(fl + (rowbase + w0 + px) * CV_C + c4 * 4);
            const int x = x0 + px;
            if (x >= 0 && x < W) vr[hf][j] = *reinterpret_cast<const float4 *>(fr + (rowbase + x) * CV_C + c4 * 4);
        }
    const int wave = tid >> 6, lane = tid & 63;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;  // this wave's 32x32 sub-block (w rows, x cols)
    f32x16 acc, acc2;                                       // hi*hi, and the two cross terms
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f, acc2[i] = 0.f;
    const char *pa = oL + (wr + (lane & 31)) * CVM_SP + 16 * (lane >> 5);
    const char *pb = oR + (wc + (lane & 31)) * CVM_SP + 16 * (lane >> 5);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        if (hf) __syncthreads();   // everyone is done with the first half's tiles
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = j * 256 + tid, px = p >> 3, c = p & 7;
            const int off = px * CVM_SP + (c >> 2) * 64 + (c & 3) * 8;
            cv_h4 hh, ll;
            cv_split4(vl[hf][j], hh, ll);
            *reinterpret_cast<cv_h4 *>(oL + off) = hh;
            *reinterpret_cast<cv_h4 *>(oL + off + 32) = ll;
            cv_split4(vr[hf][j], hh, ll);
            *reinterpret_cast<cv_h4 *>(oR + off) = hh;
            *reinterpret_cast<cv_h4 *>(oR + off + 32) = ll;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const cv_h8 a_hi = *reinterpret_cast<const cv_h8 *>(pa + q * 64);
            const cv_h8 a_lo = *reinterpret_cast<const cv_h8 *>(pa + q * 64 + 32);
            const cv_h8 b_hi = *reinterpret_cast<const cv_h8 *>(pb + q * 64);
            const cv_h8 b_lo = *reinterpret_cast<const cv_h8 *>(pb + q * 64 + 32);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc2, 0, 0, 0);
        }
    }
    __syncthreads();   // every wave is done with the operand tiles: their LDS becomes the product tile
    if (PIXEL_MAJOR) {
        constexpr int TP = 65;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int row = wr + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
            const int col = wc + (lane & 31);
            sT[row * TP + col] = (acc[rg] + acc2[rg]) * (-1.f / 1048576.f);
        }
        __syncthreads();
        const int dbase = w0 - x0;
        // left volume: pixel w0 + r takes row r of T: lane = x, d = dbase + r - lane
        for (int r = wave; r < 64; r += 4) {
            const int w = w0 + r, d = dbase + r - lane;
            if (w < W && d >= 0 && d < D && w >= d) lcv[(rowbase + w) * (size_t)Dp + d] = sT[r * TP + lane];
        }
        // right volume: pixel x0 + c takes column c of T: lane = w, d = dbase + lane - c
        for (int c = wave; c < 64; c += 4) {
            const int x = x0 + c, d = dbase + lane - c;
            if (x >= 0 && w0 + lane < W && d >= 0 && d < D) rcv[(rowbase + x) * (size_t)Dp + d] = sT[lane * TP + c];
        }
        return;
    }
    // Products go to LDS diagonal-major and folded, already negated: T[(w - x) & 63][w] (w, x local).  A row of T
    // holds diagonal dd = row in its columns row..63 and diagonal row - 64 in its columns 0..row-1; both run along
    // w, which is how both volumes are written (lcv[d][h][w] and rcv[d][h][w - d]).
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        const int row = wr + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
        const int col = wc + (lane & 31);
        sT[((row - col) & 63) * CVM_TP + row] = (acc[rg] + acc2[rg]) * (-1.f / 1048576.f);
    }
    __syncthreads();
    const size_t plane = (size_t)H * W;
    const int dbase = w0 - x0;  // d of the main diagonal
    // A lane stores 4 consecutive w of one diagonal (16 B, dword aligned) to both volumes - the vector-memory pipe
    // costs about the same per wave instruction whatever its width; one instruction covers 4 rows of T x 16 quads.
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const int q = lane & 15, sub = lane >> 4;
    for (int rgp = wave; rgp < 16; rgp += 4) {
        const int row = rgp * 4 + sub;
        const int wl = 4 * q;          // first w_local of this lane's quad
        const float4 t = *reinterpret_cast<const float4 *>(&sT[row * CVM_TP + wl]);
        const float v[4] = {t.x, t.y, t.z, t.w};
        const int w = w0 + wl;
        if (wl >= row || wl + 3 < row) {
            const int dd = wl >= row ? row : row - 64;
            const int d = dbase + dd;
            if (d < 0 || d >= D || w >= W) continue;
            float *pl = lcv + (size_t)d * plane + rowbase + w;
            float *pr = rcv + (size_t)d * plane + rowbase + (w - d);
            if (w + 3 < W) {
                f4u o;
                o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
                *reinterpret_cast<f4u *>(pl) = o;
                *reinterpret_cast<f4u *>(pr) = o;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (w + j < W) {
                        pl[j] = v[j];
                        pr[j] = v[j];
                    }
            }
        }
    }
    // The quad of a row that straddles the fold (rows not divisible by 4) holds the last elements of one diagonal and
    // the first of the other: one element per thread, all lanes busy - 2 store instructions per wave instead of 8
    // nearly empty ones inside every iteration of the loop above.
    {
        const int row = tid >> 2, j = tid & 3;
        const int c = (row & ~3) + j;                     // column of the element in its row of T
        const int d = dbase + (c >= row ? row : row - 64);
        const int w = w0 + c;
        if ((row & 3) != 0 && d >= 0 && d < D && w < W) {
            const float v = sT[row * CVM_TP + c];
            lcv[(size_t)d * plane + rowbase + w] = v;
            rcv[(size_t)d * plane + rowbase + (w - d)] = v;
        }
    }
}

// Border recurrences on the negated volumes.  One wavefront per (plane d, 64 image rows, side): lane = row, each lane
// runs its own 3-tap recurrence (<= d sequential steps); 64 steps are staged in LDS and written back transposed, so the
// stores are 256-byte runs along w instead of 64 scattered dwords per instruction.
__global__ __launch_bounds__(64) void cost_volume_fill_kernel(float *__restrict__ lcv, float *__restrict__ rcv, int D,
                                                              int H, int W)
{
    __shared__ float tile[64 * 65];
    const int lane = threadIdx.x;
    const int d = D - 1 - blockIdx.y;          // plane 0 has no border; the longest recurrences start first
    const int h0 = blockIdx.x * 64;
    const int h = min(h0 + lane, H - 1);       // surplus lanes repeat the last row and store nothing
    const int nrows = min(64, H - h0);
    if (blockIdx.z == 0) {
        // pf:94-95: column w = d-1 .. 0 from columns w+1, w+2, w+3 (NumPy sums them in ascending order)
        float *L = lcv + ((size_t)d * H + h) * W;
        float x1 = L[d], x2 = L[d + 1], x3 = L[d + 2];
        for (int wtop = d - 1; wtop >= 0; wtop -= 64) {
            const int cnt = min(64, wtop + 1);
            for (int j = 0; j < cnt; ++j) {
                // (the reference sums BEFORE it negates, and NumPy's reduction starts from the identity +0: the sum of the
                // reference's values is 0 - x1 - x2 - x3 on the negated ones - the same bits, a zero sum's sign included)
                float s = 0.f - x1;
                s = s - x2;
                s = s - x3;
                const float v = -(s / 3.f);
                tile[lane * 65 + j] = v;
                x3 = x2; x2 = x1; x1 = v;
            }
            __syncthreads();
            if (lane < cnt)
                for (int r = 0; r < nrows; ++r)
                    lcv[((size_t)d * H + h0 + r) * W + (wtop - lane)] = tile[r * 65 + lane];
            __syncthreads();
        }
    } else {
        // pf:105-106: column w = W-d .. W-1 from columns w-3, w-2, w-1
        float *Rr = rcv + ((size_t)d * H + h) * W;
        float x1 = Rr[W - d - 3], x2 = Rr[W - d - 2], x3 = Rr[W - d - 1];
        for (int wbot = W - d; wbot < W; wbot += 64) {
            const int cnt = min(64, W - wbot);
            for (int j = 0; j < cnt; ++j) {
                // (the reference sums BEFORE it negates, and NumPy's reduction starts from the identity +0: the sum of the
                // reference's values is 0 - x1 - x2 - x3 on the negated ones - the same bits, a zero sum's sign included)
                float s = 0.f - x1;
                s = s - x2;
                s = s - x3;
                const float v = -(s / 3.f);
                tile[lane * 65 + j] = v;
                x1 = x2; x2 = x3; x3 = v;
            }
            __syncthreads();
            if (lane < cnt)
                for (int r = 0; r < nrows; ++r)
                    rcv[((size_t)d * H + h0 + r) * W + (wbot + lane)] = tile[r * 65 + lane];
            __syncthreads();
        }
    }
}

// Border recurrences (pf:94-95, 105-106) on pixel-major volumes: disparities on lanes (4 per lane and 256-group), one
// wave per image row and side, sweeping the columns the borders touch.  Left volume: columns D+1 down to 0; a lane's
// component d takes the stored score while column >= d (those are its three seeds when the sweep reaches d+2, d+1, d)
// and the 3-tap mean of its last three values below that - the same float32 operations in the same order as the
// plane-major fill kernel (NumPy sums the slice in ascending column order: nearest column first here, oldest first on
// the right-volume side).  Right volume: columns W-D-2 up to W-1, scores while column < W - d.
typedef uint32_t cv_u32x4 __attribute__((ext_vector_type(4)));
template <int NG>
__global__ __launch_bounds__(64) void cost_volume_fill_hwd_kernel(float *__restrict__ lcv, float *__restrict__ rcv, int D,
                                                                  int Dp, int H, int W)
{
    constexpr int PF = 8;
    constexpr int kDrop = 0x7ffffff0;
    const int lane = threadIdx.x, h = blockIdx.x;
    const bool left = blockIdx.y == 0;
    float *row = (left ? lcv : rcv) + (size_t)h * W * Dp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int)((size_t)W * Dp * 4), 0x00020000);
    const unsigned pix = (unsigned)Dp * 4u;
    const int c_first = left ? D + 1 : max(W - D - 2, 0), nsteps = left ? D + 2 : W - c_first;
    auto col = [&](int t) { return left ? c_first - t : c_first + t; };
    int dbase[NG], voff[NG];
    float x1[NG][4], x2[NG][4], x3[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        dbase[g] = g * 256 + lane * 4;
        voff[g] = dbase[g] < Dp ? dbase[g] * 4 : kDrop;
#pragma unroll
        for (int i = 0; i < 4; ++i) x1[g][i] = x2[g][i] = x3[g][i] = 0.f;
    }
    cv_u32x4 buf[PF][NG];
    auto issue = [&](int slot, int t) {
        const unsigned so = (unsigned)col(min(t, nsteps - 1)) * pix;
#pragma unroll
        for (int g = 0; g < NG; ++g) buf[slot][g] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[g], so, 0);
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) issue(k, k);
    for (int t0 = 0; t0 < nsteps; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            if (t >= nsteps) continue;
            const int c = col(t);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const cv_u32x4 u = buf[k][g];
                const float v[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
                cv_u32x4 o;
                bool any = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = dbase[g] + i;
                    const bool stored = left ? c >= d : c < W - d;     // the score itself (d >= D: pad, never used)
                    float s = 0.f - x1[g][i];     // the reference's (0 + x1 + x2 + x3) / 3 on its not yet negated values
                    s = s - x2[g][i];
                    s = s - x3[g][i];
                    const float val = stored ? v[i] : -(s / 3.f);
                    any |= !stored && d < D;
                    if (left) { x3[g][i] = x2[g][i]; x2[g][i] = x1[g][i]; x1[g][i] = val; }     // x1 = column c + 1 next
                    else      { x1[g][i] = x2[g][i]; x2[g][i] = x3[g][i]; x3[g][i] = val; }     // x3 = column c - 1 next
                    o[i] = __float_as_uint(val);
                }
                // components that still hold their score are written back unchanged
                buffer_store_b128<0>(o, rs, any ? voff[g] : kDrop, (unsigned)c * pix);
            }
            issue(k, t + PF);
        }
    }
}


// The same recurrences with ONE disparity per lane: a wave owns 64 consecutive disparities of one image row and side,
// so a row has Dp / 64 waves instead of one (cfg2: 4000 waves instead of 1000 - the four-per-lane form above is one
// wave per SIMD walking D + 2 dependent 1 KiB loads, 0.13 ms of latency for 34 M entries), each sweeping only the
// columns its own disparities touch (chunk q: 64 q + 66 of them), 16 loads in flight.  Same float32 operations in the
// same order per entry as cost_volume_fill_hwd_kernel (and as the plane-major fill kernel): bit-identical.
__global__ __launch_bounds__(64) void cost_volume_fill_hwd_lanes_kernel(float *__restrict__ lcv, float *__restrict__ rcv,
                                                                        int D, int Dp, int H, int W)
{
    constexpr int PF = 16;
    constexpr int kDrop = 0x7ffffff0;
    const int lane = threadIdx.x, h = blockIdx.x;
    const bool left = blockIdx.y == 0;
    const int d = (int)blockIdx.z * 64 + lane;
    const int dmax = min((int)blockIdx.z * 64 + 63, D - 1);       // the largest disparity of this wave that exists
    if ((int)blockIdx.z * 64 >= D) return;
    float *row = (left ? lcv : rcv) + (size_t)h * W * Dp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int)((size_t)W * Dp * 4), 0x00020000);
    const unsigned pix = (unsigned)Dp * 4u;
    const int c_first = left ? dmax + 2 : max(W - dmax - 3, 0), nsteps = left ? c_first + 1 : W - c_first;
    auto col = [&](int t) { return left ? c_first - t : c_first + t; };
    const int voff = d < Dp ? d * 4 : kDrop;
    const int sto = d < D ? voff : kDrop;
    float x1 = 0.f, x2 = 0.f, x3 = 0.f;
    float buf[PF];
    // Phase A: the columns from which some lane of the wave still takes its score (columns >= the wave's smallest
    // disparity on the left side, < W - that disparity on the right): at most 66 of them.  Their loads are
    // unconditional (a step past the last such column re-reads it, unused), so the compiler's vmcnt bookkeeping stays
    // exact - a load inside a condition made every step wait for everything in flight, its own store included.
    const int dmin = (int)blockIdx.z * 64;
    const int nA = min(nsteps, left ? c_first - dmin + 1 : (W - dmin) - c_first);
    auto issue = [&](int slot, int t) {
        const int c = col(min(t, nA - 1));
        buf[slot] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)c * pix, 0));
    };
    // s / 3.f, correctly rounded, in three operations instead of the compiler's ten: q = RN(s * RN(1/3)), one residual,
    // one correction (division by a constant, Brisebarre / Muller).  tools/probe/div3_exhaustive.c compares it with the
    // division for ALL 2^32 float32 inputs: identical bits (denormals, NaN payloads included) except for -0.0 and
    // +-inf, which are handed through as they are (s / 3 = s for them).  The recurrence is a chain of 250 dependent
    // steps per wave with four waves per SIMD: the division was most of its time.
    auto third = [](float s) {
        const float zh = 0.3333333432674407958984375f;            // RN(1/3) = 0x3eaaaaab
        const float q = s * zh;
        const float r = __builtin_fmaf(-3.0f, q, s);
        const float q2 = __builtin_fmaf(r, zh, q);
        return (__builtin_fabsf(s) < __builtin_inff() && s != 0.f) ? q2 : s;
    };
    auto step = [&](int c, bool stored, float have) {
        // the reference's (0 + x1 + x2 + x3) / 3 on its not yet negated values (pf:94-95 run before pf:111-112, and NumPy's
        // reduction starts from the identity +0): 0 - x1 - x2 - x3 on the negated ones, negated again - the same bits as
        // summing the stored values except that a sum that is exactly zero comes out as -0.0, like the reference's
        float s = 0.f - x1;
        s = s - x2;
        s = s - x3;
        const float val = stored ? have : -third(s);
        if (left) { x3 = x2; x2 = x1; x1 = val; }               // x1 = column c + 1 next
        else      { x1 = x2; x2 = x3; x3 = val; }               // x3 = column c - 1 next
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, stored ? kDrop : sto, (unsigned)c * pix, 0);
    };
    if (nA > 0) {
#pragma unroll
        for (int k = 0; k < PF; ++k) issue(k, k);
        for (int t0 = 0; t0 < nA; t0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int t = t0 + k;
                if (t < nA) {
                    const int c = col(t);
                    step(c, left ? c >= d : c < W - d, buf[k]);     // stored: the score itself (d >= D: pad, never used)
                }
                issue(k, t + PF);
            }
        }
    }
    // Phase B: behind them the sweep is pure recurrence for every lane - no load, hence no wait: the stores just leave
    for (int t = max(nA, 0); t < nsteps; ++t) step(col(t), false, 0.f);
}

}  // namespace mccnn

extern "C" int mccnn_cost_volume(const float *fl, const float *fr, int H, int W, int C, int D, float *lcv, float *rcv,
                                 int mode, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(fl && fr && lcv && rcv, MCCNN_E_INVALID, "mccnn_cost_volume: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0 && D > 0, MCCNN_E_INVALID, "mccnn_cost_volume: non-positive size");
    MCCNN_REQUIRE(C == CV_C, MCCNN_E_UNSUPPORTED, "mccnn_cost_volume: C=%d, kernels are built for 64 channels", C);
    MCCNN_REQUIRE(D <= W - 2, MCCNN_E_UNSUPPORTED,
                  "mccnn_cost_volume: D=%d needs W >= D+2 (the reference's border recurrence pf:106 is degenerate "
                  "beyond that), W=%d", D, W);
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256);
    if (mode == MCCNN_CV_EXACT) {
        const dim3 grid(cdiv(W, CV_TW), H, cdiv(D, CV_DT));
        hipLaunchKernelGGL(cost_volume_exact_kernel<false>, grid, block, 0, s, fl, fr, H, W, D, lcv, rcv, 0);
    } else if (mode == MCCNN_CV_MFMA) {
        const int nwt = cdiv(W, 64), nbands = cdiv(D + 63, 64) + 1;
        const long total = (long)nwt * nbands * H;
        MCCNN_REQUIRE(total <= 0x7fffffffL, MCCNN_E_UNSUPPORTED, "mccnn_cost_volume: %ld tiles exceed the grid", total);
        hipLaunchKernelGGL(cost_volume_mfma_kernel<false>, dim3((unsigned)total), block, 0, s, fl, fr, H, W, D, lcv, rcv, nwt,
                           nbands, (int)total, 0);
    } else {
        MCCNN_REQUIRE(false, MCCNN_E_INVALID, "mccnn_cost_volume: unknown mode %d", mode);
    }
    int rc = check_launch("mccnn_cost_volume");
    if (rc) return rc;
    if (D > 1)
        hipLaunchKernelGGL(cost_volume_fill_kernel, dim3(cdiv(H, 64), D - 1, 2), dim3(64), 0, s, lcv, rcv, D, H, W);
    return check_launch("mccnn_cost_volume(fill)");
}

extern "C" int mccnn_cost_volume_hwd(const float *fl, const float *fr, int H, int W, int C, int D, float *lcv_hwd,
                                     float *rcv_hwd, int mode, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(fl && fr && lcv_hwd && rcv_hwd, MCCNN_E_INVALID, "mccnn_cost_volume_hwd: null pointer");
    MCCNN_REQUIRE(H > 0 && W > 0 && D > 0, MCCNN_E_INVALID, "mccnn_cost_volume_hwd: non-positive size");
    MCCNN_REQUIRE(C == CV_C, MCCNN_E_UNSUPPORTED, "mccnn_cost_volume_hwd: C=%d, kernels are built for 64 channels", C);
    MCCNN_REQUIRE(D <= W - 2, MCCNN_E_UNSUPPORTED,
                  "mccnn_cost_volume_hwd: D=%d needs W >= D+2 (the reference's border recurrence pf:106 is degenerate "
                  "beyond that), W=%d", D, W);
    MCCNN_REQUIRE(mode == MCCNN_CV_EXACT || mode == MCCNN_CV_MFMA, MCCNN_E_INVALID, "mccnn_cost_volume_hwd: unknown mode %d",
                  mode);
    MCCNN_REQUIRE(D <= 512, MCCNN_E_UNSUPPORTED, "mccnn_cost_volume_hwd: D=%d > 512", D);
    const int Dp = mccnn_hwd_pitch(D);
    MCCNN_REQUIRE((size_t)W * Dp * 4 < ((size_t)1 << 31), MCCNN_E_UNSUPPORTED,
                  "mccnn_cost_volume_hwd: a %d x %d row exceeds a buffer descriptor's reach", W, D);
    hipStream_t s = (hipStream_t)stream;
    if (mode == MCCNN_CV_MFMA) {
        const int nwt = cdiv(W, 64), nbands = cdiv(D + 63, 64) + 1;
        const long total = (long)nwt * nbands * H;
        MCCNN_REQUIRE(total <= 0x7fffffffL, MCCNN_E_UNSUPPORTED, "mccnn_cost_volume_hwd: %ld tiles exceed the grid", total);
        hipLaunchKernelGGL(cost_volume_mfma_kernel<true>, dim3((unsigned)total), dim3(256), 0, s, fl, fr, H, W, D, lcv_hwd,
                           rcv_hwd, nwt, nbands, (int)total, Dp);
    } else {
#ifdef CV_HWD_TILE_KERNEL      // A/B builds: one LDS row per score
        const dim3 grid(cdiv(W, CV_TW), H, cdiv(D, CV_DT));
        hipLaunchKernelGGL(cost_volume_exact_kernel<true>, grid, dim3(256), 0, s, fl, fr, H, W, D, lcv_hwd, rcv_hwd, Dp);
#else
        const dim3 grid(cdiv(W, CVP_TW), H, 1);
        hipLaunchKernelGGL(cost_volume_exact_pairs_kernel, grid, dim3(256), 0, s, fl, fr, H, W, D, lcv_hwd, rcv_hwd, Dp);
#endif
    }
    int rc = check_launch("mccnn_cost_volume_hwd");
    if (rc) return rc;
    if (D > 1) {
#ifdef CV_FILL_FOUR_PER_LANE     // A/B builds: the four-disparities-per-lane sweep (one wave per row and side)
        if (Dp <= 256)
            hipLaunchKernelGGL(cost_volume_fill_hwd_kernel<1>, dim3(H, 2), dim3(64), 0, s, lcv_hwd, rcv_hwd, D, Dp, H, W);
        else
            hipLaunchKernelGGL(cost_volume_fill_hwd_kernel<2>, dim3(H, 2), dim3(64), 0, s, lcv_hwd, rcv_hwd, D, Dp, H, W);
#else
        hipLaunchKernelGGL(cost_volume_fill_hwd_lanes_kernel, dim3(H, 2, cdiv(D, 64)), dim3(64), 0, s, lcv_hwd, rcv_hwd, D, Dp,
                           H, W);
#endif
    }
    return check_launch("mccnn_cost_volume_hwd(fill)");
}
