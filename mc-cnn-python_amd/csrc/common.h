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

#define MCCNN_REQUIRE(cond, code, ...)        \
    do {                                      \
        if (!(cond)) {                        \
            ::mccnn::set_error(__VA_ARGS__);  \
            return (code);                    \
        }                                     \
    } while (0)

}  // namespace mccnn
