// Shared helpers for the gfx950 kernels of libmccnn_hip.so.
// Built with -ffp-contract=off: the reference never fuses a multiply with an add, and several stages are
// bit-exact against it, so every fused multiply-add in this library is written explicitly (fmaf / MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mccnn.h"

namespace mccnn {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char *fmt, ...);

// Records the failed launch (hipGetLastError after a <<<>>>) and returns the ABI error code.
int check_launch(const char *what);

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Compute units of the CURRENT device, rounded down to whole groups of 8 (one CU of every XCD), never below 8;
// looked up per device (a process may drive several) and remembered.
int device_cus8();

// 16-byte buffer store + one wait state.  gfx950 hazard (tools/probe/storehazard.hip): when a VALU instruction writes
// the store's first data register in the issue slot right behind a buffer_store_dwordx4 whose soffset is an SGPR, the
// store picks up the NEW value in lanes 12-15 of every 16 (about 1 % of the stores under load).  The compiler pads
// only the immediate-soffset form (2 wait states) and lets the register allocator reuse the data registers at once
// otherwise; one wait state is enough.  With data that came out of LDS (ds_read + wait in front of the store) the same
// pattern still fails, 4e-7 of the time instead of 1e-2; 12-byte stores behave like 16-byte ones, 8- and 4-byte stores
// are not affected.  The asm reads the data registers, so no later writer can move above it.
#ifdef __HIPCC__
template <int AUX, typename V4>
__device__ __forceinline__ void buffer_store_b128(V4 v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, AUX);
    asm volatile("s_nop 0" ::"v"(v));
}
template <int AUX, typename V3>   // the 12-byte form is a "more than 8 bytes" store as well
__device__ __forceinline__ void buffer_store_b96(V3 v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b96(v, rs, voff, soff, AUX);
    asm volatile("s_nop 0" ::"v"(v));
}
#endif

#define MCCNN_REQUIRE(cond, code, ...)        \
    do {                                      \
        if (!(cond)) {                        \
            ::mccnn::set_error(__VA_ARGS__);  \
            return (code);                    \
        }                                     \
    } while (0)

}  // namespace mccnn
