// Probe: raw-buffer range check semantics on gfx950 - is a 16-byte load checked per dword, does a negative
// (wrapped) voffset read as zero, and is soffset outside the check?   hipcc --offload-arch=gfx950 -O2 bufrange.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t *src, uint32_t *out, uint32_t *dst)
{
    // descriptor: base = src, num_records = 10 dwords (one "row"); rows selected by soffset
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src) + 16, 0, 40, 0x00020000);
    const int lane = threadIdx.x;
    const int voff = 4 * (2 * lane - 3);     // -12, -4, 4, 12, 20, 28, 36, 44 ...
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    out[lane * 4 + 0] = v.x; out[lane * 4 + 1] = v.y; out[lane * 4 + 2] = v.z; out[lane * 4 + 3] = v.w;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + 16, 0, 40, 0x00020000);
    u32x4 w = {1000u + lane, 2000u + lane, 3000u + lane, 4000u + lane};
    if (lane < 8) __builtin_amdgcn_raw_buffer_store_b128(w, rd, 4 * (3 * lane - 2), 0, 0);
}
int main()
{
    uint32_t h[64], *d, *o, *dst;
    for (int i = 0; i < 64; ++i) h[i] = 100 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 16); hipMalloc(&dst, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(dst, 0, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, dst);
    static uint32_t r[256], rd[64];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    hipMemcpy(rd, dst, sizeof(rd), hipMemcpyDeviceToHost);
    printf("row 1 holds 116..125 (10 in-range dwords), then 126.. belongs to the next row\n");
    for (int l = 0; l < 9; ++l)
        printf("lane %d voffset %3d dwords: %u %u %u %u\n", l, 4 * (2 * l - 3), r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    printf("stores (lane l writes 4 dwords at dword 3l-2 of row 1; in range: dwords 0..9):\n");
    for (int i = 12; i < 32; ++i) printf("%u ", rd[i]);
    printf("\n");
    return 0;
}
