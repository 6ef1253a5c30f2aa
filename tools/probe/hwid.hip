// Probe: which SIMD does wave w of a 256-thread workgroup land on?  (decides how roles of wave-specialised kernels
// spread over the four SIMDs of a CU)   hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid && ./hwid
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void probe(uint32_t *out, int spin)
{
    __shared__ double pad[4700];
    const int wave = threadIdx.x >> 6;
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    pad[threadIdx.x] = hw;
    // keep the WG resident for a while so that 4 WGs share a CU
    long t0 = clock64();
    while (clock64() - t0 < spin) { pad[threadIdx.x + 256] += 1.0; }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + wave) * 2] = hw;
        out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc;
    }
    if (pad[threadIdx.x + 256] == 12345.0) out[0] = 0;
}
int main()
{
    const int nb = 2048;
    uint32_t *d;
    hipMalloc(&d, nb * 8 * 4);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    static uint32_t h[nb * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    int hist[4][4] = {{0}};
    int distinct = 0;
    for (int b = 0; b < nb; ++b) {
        int seen = 0;
        for (int w = 0; w < 4; ++w) {
            const uint32_t hw = h[(b * 4 + w) * 2];
            const int simd = (hw >> 4) & 3;
            hist[w][simd]++;
            seen |= 1 << simd;
        }
        distinct += seen == 15;
    }
    printf("workgroups whose 4 waves sit on 4 distinct SIMDs: %d / %d\n", distinct, nb);
    for (int w = 0; w < 4; ++w) printf("wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 24; ++b) {
        printf("block %4d xcc %u:", b, h[b * 8 + 1] & 15);
        for (int w = 0; w < 4; ++w) {
            const uint32_t hw = h[(b * 4 + w) * 2];
            printf("  [se %u cu %2u simd %u slot %u]", (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    return 0;
}
