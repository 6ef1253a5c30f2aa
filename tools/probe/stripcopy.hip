// Probe: what does the memory system give a "strip streaming" access pattern?  One wave per (strip, plane) walks
// down the rows of a [D][H][W] float volume reading 1 KiB row segments (256 columns, 16 B per lane) and writing 224
// of them, PF rows of loads in flight - the traffic of the streaming CBCA kernel without any of its on-chip work.
//   hipcc --offload-arch=gfx950 -O3 stripcopy.hip -o stripcopy && ./stripcopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PF, int AUXL, int AUXS>
__global__ __launch_bounds__(64) void stripcopy(const float *in, float *out, int D, int H, int W, int nstrips, int total)
{
    int id;
    {
        const int b = blockIdx.x, q = total >> 3, r = total & 7, x = b & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int strip = id % nstrips, d = id / nstrips, lane = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in + d * plane), 0, (int)(plane * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(out + d * plane, 0, (int)(plane * 4), 0x00020000);
    const int w0 = strip * 224;
    const int vb = 4 * max(w0 - 16 + 4 * lane, 0);
    const int c0 = w0 + 4 * lane;
    const int ob = (lane < 56 && c0 + 3 < W) ? 4 * c0 : 0x7ffffff0;
    u32x4 v[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, vb, min(k, H - 1) * 4 * W, AUXL);
    for (int y = 0; y < H; y += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            u32x4 o = v[k];
            o.x += 1;
            if (y + k < H) __builtin_amdgcn_raw_buffer_store_b128(o, rd, ob, (y + k) * 4 * W, AUXS);
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, vb, min(y + k + PF, H - 1) * 4 * W, AUXL);
        }
    }
}
template <int PF, int AUXL, int AUXS>
static void run(const char *name, const float *in, float *out, int D, int H, int W)
{
    const int nstrips = (W + 223) / 224, total = nstrips * D;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stripcopy<PF, AUXL, AUXS>), dim3(total), dim3(64), 0, 0, in, out, D, H, W, nstrips, total);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-28s %dx%dx%d  %.4f ms  %.0f GB/s algorithmic\n", name, W, H, D, ms, 8.0 * D * H * W / ms / 1e6);
}
int main()
{
    const int shapes[][3] = {{256, 500, 750}, {256, 500, 896}, {192, 375, 1242}};
    for (auto &s : shapes) {
        const int D = s[0], H = s[1], W = s[2];
        float *in, *out;
        hipMalloc(&in, (size_t)D * H * W * 4); hipMalloc(&out, (size_t)D * H * W * 4);
        hipMemset(in, 0, (size_t)D * H * W * 4);
        run<4, 0, 0>("PF4", in, out, D, H, W);
        run<8, 0, 0>("PF8", in, out, D, H, W);
        run<16, 0, 0>("PF16", in, out, D, H, W);
        run<16, 2, 0>("PF16 nt loads", in, out, D, H, W);
        run<16, 0, 2>("PF16 nt stores", in, out, D, H, W);
        run<16, 2, 2>("PF16 nt both", in, out, D, H, W);
        run<32, 0, 0>("PF32", in, out, D, H, W);
        hipFree(in); hipFree(out);
    }
    return 0;
}
