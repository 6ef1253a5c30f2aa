// Probe: what does gfx950's direct-to-LDS buffer load cost next to the ordinary path through registers?
// Every wave streams `iters` x 8 KiB of an L2-resident buffer into its own LDS region and reads it back:
//   mode 0: buffer_load_dwordx4 -> VGPR -> ds_write_b128, ds_read_b128
//   mode 1: buffer_load_dwordx4 ... lds (16 bytes per lane), ds_read_b128
//   mode 2: buffer_load_dword ... lds (4 bytes per lane, four instructions for the same bytes), ds_read_b128
//   mode 3: buffer_load_dwordx4 -> VGPR only (no LDS at all; the reference point)
//   hipcc --offload-arch=gfx950 -O3 ldsdma.hip -o ldsdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int NB = 8;   // 1 KiB blocks per round

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters, int nbytes)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * NB * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    lds_char *const base = (lds_char *)lds + wv * (NB * 1024);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, nbytes, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned wave = blockIdx.x * 4 + wv;
    for (int it = 0; it < iters; ++it) {
        const unsigned soff = ((wave * 131u + (unsigned)it * 17u) % (unsigned)(nbytes / (NB * 1024))) * (NB * 1024);
        if (MODE == 0 || MODE == 3) {
            u32x4 v[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) v[b] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff + b * 1024, 0);
            if (MODE == 0) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(base + b * 1024 + lane * 16) = v[b];
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc.x += __uint_as_float(v[b].x); acc.y += __uint_as_float(v[b].y);
                    acc.z += __uint_as_float(v[b].z); acc.w += __uint_as_float(v[b].w);
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(base + b * 1024), 16, lane * 16, soff + b * 1024, 0, 0);
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(base + b * 1024 + q * 256), 4, lane * 4,
                                                             soff + b * 1024 + q * 256, 0, 0);
        }
        if (MODE != 3) {
            __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const f32x4 r = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(base + b * 1024 + lane * 16);
                acc += r;
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE> static void run(const char *name, const float *in, float *out, int nbytes)
{
    const int blocks = 256 * 2, iters = 400;    // 8 waves per CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 4, nbytes);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters, nbytes);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 4 * iters * NB * 1024;
    printf("%-58s %7.3f ms  %7.1f GB/s  = %5.1f B/clk/CU @2.4 GHz\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
}
int main()
{
    const int nbytes = 1 << 20;    // stays in every L2
    float *in, *out;
    hipMalloc(&in, nbytes); hipMemset(in, 0, nbytes);
    hipMalloc(&out, 256 * 2 * 256 * 4);
    run<3>("buffer_load_dwordx4 -> registers (no LDS)", in, out, nbytes);
    run<0>("buffer_load_dwordx4 -> registers -> ds_write_b128 -> ds_read", in, out, nbytes);
    run<1>("buffer_load_dwordx4 ... lds -> ds_read", in, out, nbytes);
    run<2>("4 x buffer_load_dword ... lds -> ds_read", in, out, nbytes);
    return 0;
}
