// Probe: what does the 256 MiB memory-side cache (Infinity Cache / MALL) give a kernel that streams at full rate?
// Three access patterns over a working set of S bytes, each launched REPS times back to back so that launch n+1
// meets what launch n left behind; one line per S.
//   read     : every launch reads the same S bytes (16 B per lane, 8 loads in flight per lane)
//   pingpong : launch n reads X and writes Y, launch n+1 reads Y and writes X (S bytes each: the CBCA iteration shape)
//   copy     : every launch reads X and writes Y
// The cold column of each pattern walks a ring of buffers > 2 GiB so that nothing a launch touches was touched by
// the ones before it.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mallstream.hip -o /tmp/mallstream && /tmp/mallstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// n16 = number of 16-byte words; grid-stride over chunks of 256 threads x 8 words
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ in, uint32_t *sink, size_t n16)
{
    const size_t chunk = 256 * 8;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        const u32x4 *p = in + c * chunk + threadIdx.x;
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + 256 * k);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
__global__ __launch_bounds__(256) void k_read_plain(const u32x4 *__restrict__ in, uint32_t *sink, size_t n16)
{
    const size_t chunk = 256 * 8;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        const u32x4 *p = in + c * chunk + threadIdx.x;
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16)
{
    const size_t chunk = 256 * 8;
    for (size_t c = blockIdx.x; c * chunk < n16; c += gridDim.x) {
        const u32x4 *p = in + c * chunk + threadIdx.x;
        u32x4 *q = out + c * chunk + threadIdx.x;
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k].x += 1;
            if (NT) __builtin_nontemporal_store(v[k], q + 256 * k); else q[256 * k] = v[k];
        }
    }
}

int main()
{
    const size_t MB = 1024 * 1024;
    const size_t ring_bytes = 3072 * MB;           // cold ring
    char *ring; uint32_t *sink;
    CK(hipMalloc(&ring, ring_bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(ring, 1, ring_bytes));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 8;
    const int sizes[] = {16, 32, 48, 64, 96, 128, 160, 192, 256, 384, 768};
    printf("%6s | %-31s | %-31s | %-31s | %-31s\n", "S MB", "read same S (nt / plain) TB/s", "read cold TB/s", "pingpong X<->Y: warm / cold", "copy X->Y nt stores: warm / cold");
    for (int smb : sizes) {
        const size_t S = smb * MB, n16 = S / 16;
        const int REPS = (int)(4096 / smb) < 8 ? 8 : (int)(4096 / smb);
        float ms;
        double r_nt, r_plain, r_cold, pp_warm, pp_cold, cp_warm, cp_cold;
        // read same
        for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) k_read<<<grid, 256>>>((const u32x4 *)ring, sink, n16); hipEventRecord(b); hipEventSynchronize(b); }
        hipEventElapsedTime(&ms, a, b); r_nt = (double)S * REPS / ms / 1e9;
        for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) k_read_plain<<<grid, 256>>>((const u32x4 *)ring, sink, n16); hipEventRecord(b); hipEventSynchronize(b); }
        hipEventElapsedTime(&ms, a, b); r_plain = (double)S * REPS / ms / 1e9;
        // read cold: walk the ring
        { const int slots = (int)(ring_bytes / S);
          for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) k_read_plain<<<grid, 256>>>((const u32x4 *)(ring + (size_t)(i % slots) * S), sink, n16); hipEventRecord(b); hipEventSynchronize(b); }
          hipEventElapsedTime(&ms, a, b); r_cold = (double)S * REPS / ms / 1e9; }
        // ping-pong warm: X = ring[0..S), Y = ring[S..2S)
        for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) { char *x = ring + (size_t)(i & 1) * S, *y = ring + (size_t)((i & 1) ^ 1) * S; k_copy<false><<<grid, 256>>>((const u32x4 *)x, (u32x4 *)y, n16); } hipEventRecord(b); hipEventSynchronize(b); }
        hipEventElapsedTime(&ms, a, b); pp_warm = 2.0 * S * REPS / ms / 1e9;
        // ping-pong cold: pairs walk the ring
        { const int slots = (int)(ring_bytes / (2 * S));
          for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) { char *x = ring + (size_t)(i % slots) * 2 * S, *y = x + S; k_copy<false><<<grid, 256>>>((const u32x4 *)x, (u32x4 *)y, n16); } hipEventRecord(b); hipEventSynchronize(b); }
          hipEventElapsedTime(&ms, a, b); pp_cold = 2.0 * S * REPS / ms / 1e9; }
        // copy with nt stores, warm (same X -> same Y) and cold
        for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) k_copy<true><<<grid, 256>>>((const u32x4 *)ring, (u32x4 *)(ring + S), n16); hipEventRecord(b); hipEventSynchronize(b); }
        hipEventElapsedTime(&ms, a, b); cp_warm = 2.0 * S * REPS / ms / 1e9;
        { const int slots = (int)(ring_bytes / (2 * S));
          for (int w = 0; w < 2; ++w) { hipEventRecord(a); for (int i = 0; i < REPS; ++i) { char *x = ring + (size_t)(i % slots) * 2 * S, *y = x + S; k_copy<true><<<grid, 256>>>((const u32x4 *)x, (u32x4 *)y, n16); } hipEventRecord(b); hipEventSynchronize(b); }
          hipEventElapsedTime(&ms, a, b); cp_cold = 2.0 * S * REPS / ms / 1e9; }
        printf("%6d | nt %6.2f  plain %6.2f           | %6.2f                          | %6.2f / %6.2f  (r+w bytes)      | %6.2f / %6.2f\n", smb, r_nt, r_plain, r_cold, pp_warm, pp_cold, cp_warm, cp_cold);
        fflush(stdout);
    }
    return 0;
}
