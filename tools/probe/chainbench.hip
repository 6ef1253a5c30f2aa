// Probe: what a wave-uniform, scalar-controlled float32 add chain costs on gfx950 - the inner loop of cbca_hwd_kernel
// (one s_bitcmp + s_cbranch per region element, four v_add_f32 into a 4-float accumulator, operands in registers) -
// as a function of the waves resident per SIMD.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize chainbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
constexpr int NS = 12;   // slots per arm
struct V4 { float x, y, z, w; };
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void add4(V4 &a, const V4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void addpk(V4 &a, const V4 &b)
{
    f2 lo = {a.x, a.y}, hi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
    lo += blo; hi += bhi;
    a.x = lo.x; a.y = lo.y; a.z = hi.x; a.w = hi.y;
}
template <int MODE, int Z> __device__ __forceinline__ void walk(V4 &a, const V4 (&win)[NS], uint32_t m)
{
    if constexpr (Z < NS) {
        if (MODE == 1 || (m & (1u << Z))) {
            if (MODE == 3) addpk(a, win[Z]); else add4(a, win[Z]);
            walk<MODE, Z + 1>(a, win, m);
        }
    }
}
// MODE 0: test + branch + 4 adds per element; 1: adds only; 3: test + branch + 2 packed adds; 4: like 0, one add
template <int MODE>
__global__ __launch_bounds__(64) void k(const float *in, const uint32_t *masks, float *out, int iters, int nmask)
{
    extern __shared__ float lds[];
    V4 win[NS];
    for (int z = 0; z < NS; ++z) {
        const float *p = in + (z * 64 + threadIdx.x) * 4;
        win[z].x = p[0]; win[z].y = p[1]; win[z].z = p[2]; win[z].w = p[3];
    }
    V4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j].x = acc[j].y = acc[j].z = acc[j].w = 0.f;
    for (int it = 0; it < iters; ++it) {
        const uint32_t *mp = masks + (it % nmask) * 4;     // scalar loads: four anchors' masks
        const uint32_t m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3];
        walk<MODE, 0>(acc[0], win, m0);
        walk<MODE, 0>(acc[1], win, m1);
        walk<MODE, 0>(acc[2], win, m2);
        walk<MODE, 0>(acc[3], win, m3);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j].x + acc[j].y + acc[j].z + acc[j].w;
    if (s == 12345.f) lds[threadIdx.x] = s;
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> static void run(const char *name, int wps, int arm, float *in, uint32_t *masks, float *out)
{
    const int iters = 4000, nmask = 64;
    std::vector<uint32_t> h(nmask * 4, arm >= 32 ? 0xffffffffu : ((1u << arm) - 1u));
    hipMemcpy(masks, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * 4 * wps;                 // one round of resident waves
    const size_t lds = (160 * 1024 / (4 * wps)) & ~255; // LDS caps the waves per CU at 4 * wps
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, in, masks, out, 10, nmask);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, in, masks, out, iters, nmask);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double elems = (double)iters * 4 * (arm > NS ? NS : arm);
    const double ns_wave = ms * 1e6 / elems;           // per element, as one wave sees it
    printf("%-44s arm %2d waves/SIMD %d: %6.2f ns per element per wave = %5.1f clk @2.4GHz; per SIMD %5.1f clk/element\n", name,
           arm, wps, ns_wave, ns_wave * 2.4, ns_wave * 2.4 / wps);
}
int main()
{
    float *in, *out; uint32_t *masks;
    hipMalloc(&in, NS * 64 * 16); hipMemset(in, 0, NS * 64 * 16);
    hipMalloc(&out, (size_t)256 * 4 * 8 * 64 * 4); hipMalloc(&masks, 64 * 16);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        run<0>("bitcmp + branch + 4 v_add", wps, 32, in, masks, out);
        run<0>("bitcmp + branch + 4 v_add", wps, 3, in, masks, out);
        run<0>("bitcmp + branch + 4 v_add", wps, 1, in, masks, out);
        run<1>("4 v_add, straight line", wps, 32, in, masks, out);
        run<3>("bitcmp + branch + 2 v_pk_add", wps, 32, in, masks, out);
    }
    return 0;
}
