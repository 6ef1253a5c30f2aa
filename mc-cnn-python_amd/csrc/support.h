// The packed per-pixel support word (mccnn_support_t, include/mccnn.h) and its decoders, shared by the kernels that
// read the support planes mccnn_cross_arms writes (cross_cbca.hip, cbca_hwd.hip).
#pragma once
#include "common.h"

namespace mccnn {

typedef uint32_t Support;  // == mccnn_support_t: bits 0-4 up, 5-9 down, 10-14 left, 15-19 right, 20-31 region size
static_assert(sizeof(mccnn_support_t) == 4, "support record must be 4 bytes");

__host__ __device__ __forceinline__ int arm_up(uint32_t a) { return (int)(a & 31u); }
__host__ __device__ __forceinline__ int arm_down(uint32_t a) { return (int)((a >> 5) & 31u); }
__host__ __device__ __forceinline__ int arm_left(uint32_t a) { return (int)((a >> 10) & 31u); }
__host__ __device__ __forceinline__ int arm_right(uint32_t a) { return (int)((a >> 15) & 31u); }
__host__ __device__ __forceinline__ int sup_count(uint32_t a) { return (int)(a >> 20); }

// ---- support buffer layout (mccnn_support_bytes): plane 0 [H][W] words, then - each 16-byte aligned - the derived
// planes private to the aggregation kernels.  Written by cross_count_kernel / cross_perm_kernel (cross_cbca.hip).
//   hsum  [H][W] uint32, emit [H][W] uint64 : cbca_stream_kernel's ready-made LDS addresses / reciprocals
//   perm  per 16 x 64 tile uint16[1024]      : cbca_ref4_kernel's lane order
//   wmask [H][W] uint32 / uint64             : cbca_hwd_kernel's window masks (below)
__host__ __device__ __forceinline__ size_t hsum_plane_offset(int H, int W) { return ((size_t)H * W * 4 + 15) & ~(size_t)15; }
__host__ __device__ __forceinline__ size_t emit_plane_offset(int H, int W)
{
    return (hsum_plane_offset(H, W) + (size_t)H * W * 4 + 15) & ~(size_t)15;
}
constexpr int PERM_TH = 16, CB_TW_C = 64;
__host__ __device__ __forceinline__ size_t perm_plane_offset(int H, int W)
{
    return (hsum_plane_offset(H, W) + (size_t)H * W * 4 + (size_t)H * W * 8 + 31) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t perm_plane_bytes(int H, int W)
{
    return (size_t)((W + CB_TW_C - 1) / CB_TW_C) * ((H + PERM_TH - 1) / PERM_TH) * (PERM_TH * CB_TW_C) * 2;
}

// The pixel-major reference-order kernel (cbca_hwd.hip) gives a wave HWD_G horizontally adjacent pixels (columns
// x0 .. x0 + HWD_G - 1, x0 a multiple of HWD_G) and keeps, per region row, the columns x0 - 13 .. x0 + HWD_G + 12 in a
// register window of HWD_NW slots.  A pixel's window mask has one bit per slot its horizontal arm covers:
//     wmask(x) = ones(left + right + 1) << (x % HWD_G + 13 - left)          (arms clamped to 13)
// so the union of a row's loads is an OR, "pixel sits this row out" is a zero mask, and every step of a chain is one
// scalar bit test - the walk costs the scalar unit ~3 instructions per pixel and row instead of ~25.
#ifndef CBCA_HWD_G
#define CBCA_HWD_G 5
#endif
constexpr int HWD_G = CBCA_HWD_G;
constexpr int HWD_R = 13;
constexpr int HWD_NW = HWD_G + 2 * HWD_R;
static_assert(HWD_NW <= 64, "window masks are at most 64-bit words");
template <bool WIDE> struct HwdMask { typedef uint32_t T; };
template <> struct HwdMask<true> { typedef uint64_t T; };
typedef HwdMask<(HWD_NW > 32)>::T hwd_mask_t;       // one bit per window slot
__host__ __device__ __forceinline__ size_t wmask_plane_offset(int H, int W)
{
    return (perm_plane_offset(H, W) + perm_plane_bytes(H, W) + 15) & ~(size_t)15;
}
// + 128: a wave reads the HWD_G words of its group in one scalar load, also where the group straddles the right edge
__host__ __device__ __forceinline__ size_t wmask_plane_bytes(int H, int W)
{
    return (size_t)H * W * sizeof(hwd_mask_t) + 128;
}

// Host-side record of what mccnn_cross_arms last wrote where (cross_cbca.hip): refuses a support plane built for
// another image size or with longer arms than the caller states; unknown pointers pass unless must_be_known (the
// pixel-major kernels read planes behind plane 0, which only a buffer written by mccnn_cross_arms has).
int check_support_record(const mccnn_support_t *support, int H, int W, int L, const char *who, bool must_be_known = false);
// Generation of the buffer's contents: every mccnn_cross_arms write gets a new one (0: never written).
unsigned long long support_generation(const mccnn_support_t *support);

}  // namespace mccnn
