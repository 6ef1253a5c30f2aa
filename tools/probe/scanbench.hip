// Probe: cycles of the float64 64-lane DPP prefix scan used by the streaming CBCA kernel (4 rows interleaved),
// alone on a SIMD and with 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 scanbench.hip -o scanbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double *out, long long *cyc, int iters)
{
    constexpr int SR = 4;
    double tt[SR], acc[SR];
    for (int j = 0; j < SR; ++j) tt[j] = threadIdx.x * 0.5 + j, acc[j] = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x111>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x112>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x114>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x118>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x142, 0xA>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x143, 0xC>(tt[j]);
#pragma unroll
            for (int j = 0; j < SR; ++j) acc[j] += dpp_f64<0x138>(tt[j]);
        } else if (MODE == 1) {   // plain dependent f64 adds, same count (7 per row)
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < SR; ++j) tt[j] += acc[j] + 1.0;
        } else {                  // row_shr steps only (no bcast)
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < SR; ++j) tt[j] += dpp_f64<0x111>(tt[j]);
        }
#pragma unroll
        for (int j = 0; j < SR; ++j) tt[j] = tt[j] * 1e-3 + 1.0;
    }
    const long long t1 = clock64();
    double s = 0;
    for (int j = 0; j < SR; ++j) s += tt[j] + acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char *name, int waves_per_cu)
{
    double *out; long long *cyc;
    const int blocks = 256, threads = 64 * waves_per_cu, iters = 2000;
    hipMalloc(&out, blocks * threads * 8); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    printf("%-34s waves/CU %2d: %.1f clock64 ticks per 4-row batch\n", name, waves_per_cu, m / 256 / iters);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : {1, 4, 8, 16}) {
        run<0>("full scan (7 steps x 4 rows)", w);
        run<1>("28 x 2 plain f64 adds", w);
        run<2>("7 row_shr steps x 4 rows", w);
    }
    return 0;
}
