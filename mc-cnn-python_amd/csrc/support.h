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

// Host-side record of what mccnn_cross_arms last wrote where (cross_cbca.hip): refuses a support plane built for
// another image size or with longer arms than the caller states; unknown pointers pass.
int check_support_record(const mccnn_support_t *support, int H, int W, int L, const char *who);

}  // namespace mccnn
