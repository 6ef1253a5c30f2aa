// Probe: is it safe on gfx950 to overwrite the first data register of a 16-byte buffer store right behind the store?
// LLVM's hazard recogniser leaves 2 wait states when the store's soffset is an immediate, and NONE when it is an SGPR
// (GCNHazardRecognizer::createsVALUHazard).  cbca_hwd_kernel's epilogue once stored the next division's intermediate in
// lanes 12-15 of every 16 (first component only) exactly for the stores with an SGPR soffset.  This probe issues
// NS stores per wave, each followed after N wait states by a v_mov into the first data register, and counts lanes
// whose stored first component is the poison value.
//   hipcc --offload-arch=gfx950 -O3 storehazard.hip -o storehazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int NS = 10;

#define STORE_SEQ(NOPS, AUX, SOFF)                                                                         \
    asm volatile("v_mov_b32 v20, %[x]\n v_mov_b32 v21, %[y]\n v_mov_b32 v22, %[y]\n v_mov_b32 v23, %[y]\n"   \
                 "buffer_store_dwordx4 v[20:23], %[vo], %[rs], " SOFF " offen" AUX "\n" NOPS                  \
                 "v_mov_b32 v20, %[p]\n"                                                                       \
                 :                                                                                             \
                 : [x] "v"(x), [y] "v"(y), [p] "v"(poison), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff)       \
                 : "v20", "v21", "v22", "v23", "memory")

// narrower stores: 8 bytes (below LLVM's "more than 8 bytes" rule) and 12 bytes
#define STORE_SEQ_N(WIDTH, REGS, NOPS)                                                                      \
    asm volatile("v_mov_b32 v20, %[x]\n v_mov_b32 v21, %[y]\n v_mov_b32 v22, %[y]\n v_mov_b32 v23, %[y]\n"   \
                 "buffer_store_" WIDTH " " REGS ", %[vo], %[rs], %[so] offen\n" NOPS                           \
                 "v_mov_b32 v20, %[p]\n"                                                                       \
                 :                                                                                             \
                 : [x] "v"(x), [y] "v"(y), [p] "v"(poison), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff)       \
                 : "v20", "v21", "v22", "v23", "memory")

// the data registers come out of LDS (ds_read_b128 + wait) instead of out of VALU instructions: conv3x3_split's epilogue
#define STORE_SEQ_LDS(NOPS)                                                                                \
    asm volatile("ds_read_b128 v[20:23], %[la]\n s_waitcnt lgkmcnt(0)\n"                                    \
                 "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen\n" NOPS                           \
                 "v_mov_b32 v20, %[p]\n"                                                                      \
                 :                                                                                             \
                 : [la] "v"(laddr), [p] "v"(poison), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(soff)              \
                 : "v20", "v21", "v22", "v23", "memory")

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int stride_bytes)
{
    const uint64_t pa = (uint64_t)out;
    u32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((uint32_t)pa);
    rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32) & 0xffffu);
    rs.z = 0x7fffffffu;
    rs.w = 0x00020000u;
    const int voff = threadIdx.x * 16;
    const float poison = -12345.f, y = 2.f;
    __shared__ float lds[NS * 256];
    for (int i = 0; i < NS; ++i) {
        lds[i * 256 + threadIdx.x * 4] = 1.f + (float)i;
        lds[i * 256 + threadIdx.x * 4 + 1] = lds[i * 256 + threadIdx.x * 4 + 2] = lds[i * 256 + threadIdx.x * 4 + 3] = y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const float x = 1.f + (float)i;
        const int soff = (blockIdx.x * NS + i) * stride_bytes;
        if (MODE == 0) STORE_SEQ("", "", "%[so]");
        if (MODE == 1) STORE_SEQ("s_nop 0\n", "", "%[so]");
        if (MODE == 2) STORE_SEQ("s_nop 1\n", "", "%[so]");
        if (MODE == 3) STORE_SEQ("s_nop 3\n", "", "%[so]");
        if (MODE == 4) STORE_SEQ("s_nop 7\n", "", "%[so]");
        if (MODE == 5) STORE_SEQ("", " nt", "%[so]");
        if (MODE == 6) STORE_SEQ("s_nop 1\n", " nt", "%[so]");
        if (MODE == 7) STORE_SEQ("s_nop 7\n", " nt", "%[so]");
        if (MODE == 8) STORE_SEQ("s_nop 7\n s_nop 7\n", " nt", "%[so]");
        if (MODE == 11) STORE_SEQ_N("dwordx2", "v[20:21]", "");
        if (MODE == 12) STORE_SEQ_N("dwordx3", "v[20:22]", "");
        if (MODE == 13) STORE_SEQ_N("dword", "v20", "");
        if (MODE == 14) STORE_SEQ_N("dwordx3", "v[20:22]", "s_nop 0\n");
        if (MODE == 9 || MODE == 10) {
            const int laddr = (int)(uintptr_t)(__attribute__((address_space(3))) float *)lds + i * 1024 + threadIdx.x * 16;
            if (MODE == 9) STORE_SEQ_LDS("");
            if (MODE == 10) STORE_SEQ_LDS("s_nop 0\n");
        }
    }
}
template <int MODE> static void run(const char *name, int blocks)
{
    const int stride = 1024;
    float *out;
    const size_t n = (size_t)blocks * NS * 256;
    hipMalloc(&out, n * 4);
    hipMemset(out, 0, n * 4);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, stride);
    hipDeviceSynchronize();
    std::vector<float> h(n);
    hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, lanes[64] = {0};
    for (size_t s = 0; s < (size_t)blocks * NS; ++s)
        for (int l = 0; l < 64; ++l)
            if (h[s * 256 + l * 4] != 1.f + (float)(s % NS)) { ++bad; ++lanes[l]; }
    printf("%-44s blocks %6d: %8ld of %ld first components wrong; lanes:", name, blocks, bad, (long)blocks * NS * 64);
    for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d", l);
    printf("\n");
    hipFree(out);
}
int main()
{
    for (int blocks : {1024, 65536}) {
        run<0>("sgpr soffset, 0 wait states", blocks);
        run<1>("sgpr soffset, 1 wait state", blocks);
        run<2>("sgpr soffset, 2 wait states", blocks);
        run<3>("sgpr soffset, 4 wait states", blocks);
        run<4>("sgpr soffset, 8 wait states", blocks);
        run<5>("sgpr soffset, nt, 0 wait states", blocks);
        run<6>("sgpr soffset, nt, 2 wait states", blocks);
        run<7>("sgpr soffset, nt, 8 wait states", blocks);
        run<8>("sgpr soffset, nt, 16 wait states", blocks);
        run<9>("data from ds_read, sgpr soffset, 0 wait states", blocks);
        run<10>("data from ds_read, sgpr soffset, 1 wait state", blocks);
        run<11>("dwordx2, sgpr soffset, 0 wait states", blocks);
        run<12>("dwordx3, sgpr soffset, 0 wait states", blocks);
        run<14>("dwordx3, sgpr soffset, 1 wait state", blocks);
        run<13>("dword, sgpr soffset, 0 wait states", blocks);
    }
    return 0;
}
