// The 64 -> 64 map 3x3 VALID convolutions of the feature stack (model.py:51-61 of the reference; a1 in SURVEY 8) on the
// matrix cores, as an opt-in alternative to the float32 library convolutions.
//
// Arithmetic.  gfx950 has no float32-input MFMA faster than v_mfma_f32_32x32x2_f32 (157 TFLOP/s, less than the
// library's Winograd kernels deliver in effective terms), but 2.5 PFLOP/s of f16.  Every float32 operand x is carried
// as two f16 numbers, x * s = hi + lo with hi = f16(x * s), lo = f16(x * s - hi) (s a power of two chosen so that
// both parts stay normal numbers): 22 significand bits.  A product a * b becomes three MFMA terms
//     a_hi * b_hi + a_hi * b_lo + a_lo * b_hi          (the a_lo * b_lo term is below 2^-22 relative)
// accumulated in float32 by the matrix core - the main term and the two cross terms in accumulators of their own
// (round 4; see the kernel).  Measured against a float64 evaluation of the stack the unit feature vectors are as close
// as the float32 library path's (tests/test_features_split_gpu.py records both; round 2's single accumulator was twice
// as far: 5.3e-7 vs 2.5e-7) - float32-class arithmetic on f16 hardware, not a reduced-precision mode, but NOT
// bit-identical to any float32 evaluation order.
//
// Activations live in HBM as "split records": 256 bytes per pixel, [q = channel / 16][hi 16 x f16 | lo 16 x f16],
// pixel-major ([N][H][W][256 B]).  A record has the size of the pixel's 64 float32 values; the last layer writes those
// (L2-normalised, tf.nn.l2_normalize, model.py:64) in the same place-value layout [N][H][W][64] the cost volume reads.
//
// conv3x3_split_kernel - implicit GEMM, M = 64 output maps (A = weights), N = pixels (B = activations), K = 9 taps x
// 64 input maps, v_mfma_f32_32x32x16_f16.  A workgroup (4 waves, one per SIMD) owns a tile of 16 x 32 output pixels;
// wave w the rows 4w .. 4w+3, one row of 32 pixels per MFMA column block: 4 pixel blocks x 2 map blocks = 8
// accumulators (128 registers).  The K loop runs over channel groups q of 16 (one MFMA K step) and the 9 taps:
//   * B: the 18 x 34 input pixels of the tile, channel group q only, are staged in LDS (80-byte slots: hi 32 B, lo
//     32 B, 16 B pad - conflict-free ds_read_b128 for 32 consecutive pixels); a tap is an immediate offset.  Two
//     buffers: group q+1 (or the next tile's group 0) is fetched into registers while group q is multiplied, and
//     written to the other buffer under the MFMAs of the group's last taps; one barrier per group.
//   * A: packed on the device once per weight set (mccnn_conv3x3_split_pack) in fragment order, 1 KiB per (K step,
//     part, map block), read from global memory (L1/L2 resident: 144 KiB per layer) three K steps ahead; B
//     fragments one step of 6 MFMAs ahead.
//   * per K step and wave: 4 A fragments, 8 B fragments, 24 MFMAs (768 cycles of the SIMD's matrix core).
// Epilogue: x / (s_a s_w) + bias, ReLU, split again (or L2-normalise), through a per-wave LDS scratch so that the
// stores are whole 256-byte records, 1 KiB per wave instruction.
// Persistent launch: one workgroup per CU walks tiles in an XCD-contiguous order (neighbouring tiles share halo
// pixels in one L2).
#include "common.h"

namespace mccnn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef uint32_t w4 __attribute__((ext_vector_type(4)));
typedef uint32_t w2 __attribute__((ext_vector_type(2)));

#ifndef CONV_STAGE_TAP
#define CONV_STAGE_TAP 0                     // the tap behind whose fragment loads the next group's activations are requested (-1: in front of tap 0)
#endif
#ifndef CONV_NBW
#define CONV_NBW 4                           // rows of 32 pixels per wave (4: one workgroup per CU; 2: two)
#endif
namespace cs {
constexpr int NBW = CONV_NBW;
constexpr int TH = 4 * NBW, TW = 32;         // output tile
constexpr int IH = TH + 2, IW = TW + 2;      // input pixels of a tile
constexpr int NPIX = IH * IW;                // 612
constexpr int SLOT = 80;                     // LDS bytes per staged pixel (one channel group): hi 32, lo 32, pad 16
constexpr int QBUF = NPIX * SLOT;            // 48 960
constexpr int NITEM = NPIX * 4;              // 16-byte pieces per staged group
constexpr int NST = (NITEM + 255) / 256;     // 10 per thread
constexpr int EPITCH = 272;                  // epilogue scratch: 256-byte record + 16 (keeps b128 alignment)
constexpr int EPX = NBW >= 4 ? 32 : 16;       // pixels of a row that pass through the epilogue scratch at a time
constexpr int EPI = EPX * EPITCH;            // per wave
constexpr int LDS_BYTES = 2 * QBUF + 4 * EPI;   // 132 736
constexpr int REC = 256;                     // bytes per pixel record
constexpr int WFRAG = 1024;                  // bytes per packed weight fragment
constexpr int WLAYER = 36 * 4 * WFRAG;       // packed weights of one layer
}  // namespace cs

__device__ __forceinline__ void split4(const float y[4], float scale, w2 &hi, w2 &lo)
{
    h4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = fminf(fmaxf(y[j] * scale, -65504.f), 65504.f);   // saturate instead of inf (|x| < 65504 / s)
        const _Float16 h = (_Float16)v;
        a[j] = h;
        b[j] = (_Float16)(v - (float)h);
    }
    hi = __builtin_bit_cast(w2, a);
    lo = __builtin_bit_cast(w2, b);
}

__device__ __forceinline__ void split4_scaled(const float v[4], w2 &hi, w2 &lo)   // values already scaled and clamped
{
    h4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)v[j];
        a[j] = h;
        b[j] = (_Float16)(v[j] - (float)h);
    }
    hi = __builtin_bit_cast(w2, a);
    lo = __builtin_bit_cast(w2, b);
}

// weights [64 out][64 in][3][3] float32 (torch layout) -> fragment order:
// [ks = q*9 + tap][part: hi, lo][map block mb][lane][8 x f16], lane: out = 32 mb + (lane & 31), in = 16 q + 8 (lane >> 5) + j
__global__ __launch_bounds__(256) void conv3x3_split_pack_kernel(const float *__restrict__ w, float scale,
                                                                 _Float16 *__restrict__ packed)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // one (ks, mb, lane)
    if (i >= 36 * 2 * 64) return;
    const int lane = i & 63, mb = (i >> 6) & 1, ks = i >> 7;
    const int q = ks / 9, tap = ks - q * 9;
    const int oc = 32 * mb + (lane & 31), ic0 = 16 * q + 8 * (lane >> 5);
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = w[((size_t)oc * 64 + ic0 + j) * 9 + tap] * scale;
        const _Float16 h = (_Float16)v;
        hi[j] = h;
        lo[j] = (_Float16)(v - (float)h);
    }
    h8 *p = reinterpret_cast<h8 *>(packed);
    p[((ks * 2 + 0) * 2 + mb) * 64 + lane] = hi;
    p[((ks * 2 + 1) * 2 + mb) * 64 + lane] = lo;
}

// Layer 1 (1 -> 64 maps) on the zero-padded image (process_functional.py:20-25), bias, ReLU, written as split records.
// Thread = one output pixel looping over the maps (weights and bias indexed uniformly: scalar loads), the same float32
// multiply-adds as mccnn_conv1_pad_bias_relu, then the split; a lane writes its pixel's record in 16-byte pieces.
__global__ __launch_bounds__(256) void conv1_split_kernel(const float *__restrict__ img, const float *__restrict__ w,
                                                          const float *__restrict__ bias, char *__restrict__ out, int H,
                                                          int W, int pad, int Ho, int Wo, float act_scale,
                                                          int *__restrict__ sat_flag)
{
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int yo = blockIdx.y, n = blockIdx.z;
    if (xo >= Wo) return;
    float v[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int y = yo + i - pad, x = xo + j - pad;
            v[i * 3 + j] = (y >= 0 && y < H && x >= 0 && x < W) ? img[((size_t)n * H + y) * W + x] : 0.f;
        }
    char *rec = out + (((size_t)n * Ho + yo) * Wo + xo) * cs::REC;
    bool sat = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w2 hi[4], lo[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 16 * q + 4 * g + j;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc += w[c * 9 + k] * v[k];
                acc += bias[c];
                y[j] = fmaxf(acc, 0.f);
                // above the records' range - or a NaN, tested BEFORE the ReLU (fmaxf drops a NaN operand, and a NaN
                // pixel must not come out as a clean 0: the pair then goes to the library path, which propagates it)
                sat |= !(acc * act_scale <= 65504.f);
            }
            split4(y, act_scale, hi[g], lo[g]);
        }
        w4 *p = reinterpret_cast<w4 *>(rec + q * 64);
        p[0] = w4{hi[0].x, hi[0].y, hi[1].x, hi[1].y};
        p[1] = w4{hi[2].x, hi[2].y, hi[3].x, hi[3].y};
        p[2] = w4{lo[0].x, lo[0].y, lo[1].x, lo[1].y};
        p[3] = w4{lo[2].x, lo[2].y, lo[3].x, lo[3].y};
    }
    if (sat_flag && sat) atomicOr(sat_flag, 1);              // a stored activation left the f16 range (see mccnn.h)
}

__device__ __forceinline__ void pin_loads()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0x78f);
}

// MODE 0: bias + ReLU, split records out.  MODE 1: bias, L2 normalisation over the 64 maps, float32 [N][Ho][Wo][64] out.
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_split_kernel(const char *__restrict__ in, const char *__restrict__ wpk,
                                                            const float *__restrict__ bias, char *__restrict__ out,
                                                            int N, int Hi, int Wi, float inv_scale, float act_scale,
                                                            int tiles_x, int tiles_y, int total,
                                                            int *__restrict__ sat_flag)
{
    using namespace cs;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = Hi - 2, Wo = Wi - 2;
    const size_t in_bytes = (size_t)N * Hi * Wi * REC, out_bytes = (size_t)N * Ho * Wo * REC;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(in), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wpk), 0, WLAYER, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        out, 0, (int)out_bytes, 0x00020000);

    // staging pieces of this thread: piece i = r*256 + tid -> pixel i >> 2 of the 18 x 34 input tile, 16-byte part i & 3
    int goff[NST], loff[NST];
#pragma unroll
    for (int r = 0; r < NST; ++r) {
        const int i = r * 256 + tid, p = min(i >> 2, NPIX - 1), part = i & 3;
        const int py = p / IW, px = p - py * IW;
        goff[r] = (py * Wi + px) * REC + part * 16;
        loff[r] = p * SLOT + part * 16;
    }
    const bool last_piece = (NST - 1) * 256 + tid < NITEM;

    // XCD-contiguous tile order (see cross_cbca.hip): virtual block v = blockIdx + i * gridDim keeps its XCD
    auto xcd_order = [](int b, int n) {
        const int q = n >> 3, r = n & 7, x = b & 7;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    };
    auto tile_base = [&](int v, int &n, int &ty0, int &tx0) {
        const int t = xcd_order(v, total);
        const int tx = t % tiles_x, r = t / tiles_x;
        const int ty = r % tiles_y;
        n = r / tiles_y;
        ty0 = ty * TH;
        tx0 = tx * TW;
    };

    w4 st[NST];
    auto fetch = [&](int n, int ty0, int tx0, int q) {
        const int so = ((n * Hi + ty0) * Wi + tx0) * REC + q * 64;     // wave-uniform
#ifdef CONV_ABL_NOSTAGE   // timing experiment: no activation loads (the LDS tile keeps whatever it holds)
        if (so == 0x7fffffff)
#endif
#pragma unroll
        for (int r = 0; r < NST; ++r) st[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, goff[r], so, 0);
    };
    auto commit = [&](int buf) {
        char *b = lds + buf * QBUF;
#pragma unroll
        for (int r = 0; r < NST; ++r)
            if (r < NST - 1 || last_piece) *reinterpret_cast<w4 *>(b + loff[r]) = st[r];
    };

    int v = blockIdx.x;
    if (v >= total) return;
    int n, ty0, tx0;
    tile_base(v, n, ty0, tx0);
    fetch(n, ty0, tx0, 0);
    commit(0);
    __syncthreads();

    // A fragments: slot ks % NSLOT, fetched NSLOT - 1 K steps ahead (the weights are the same for every tile, so the
    // stream simply wraps).  vmcnt counts in order: the activation pieces fetched at the start of a channel group
    // have to land before the first A fragment issued after them is needed, i.e. within NSLOT - 1 K steps.
    constexpr int NSLOT = MODE == 0 ? 3 : 4;     // (MODE 0 has the heavier epilogue: with four slots and the second accumulator set it spills)
    w4 af[NSLOT][4];
    auto fetch_a = [&](int slot, int ks) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            af[slot][f] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, (ks * 4 + f) * WFRAG, 0);
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) fetch_a(i, i);

    const int bfrag = (wave * NBW * IW + (lane & 31)) * SLOT + 16 * (lane >> 5);
    char *const epi = lds + 2 * QBUF + wave * EPI;

    while (true) {
        // Two accumulator sets.  The matrix core rounds its float32 accumulator once per MFMA; with the three products
        // of a K step chained into ONE accumulator that is 108 roundings at the full magnitude of the sum per output,
        // against 36 for a float32 convolution - and that, not the 22-bit operands, was what made round 2's kernel
        // twice as far from a float64 evaluation as the library (5.3e-7 vs 2.5e-7).  The cross terms
        // a_lo b_hi + a_hi b_lo are 2^-11 of the main term, so they get an accumulator of their own (its roundings
        // are 2^-11 as large) and meet the main sum once, in the epilogue: float32-accumulation accuracy.
        f16x acc[NBW][2], accx[NBW][2];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nb][mb][i] = accx[nb][mb][i] = 0.f;
        const int vn = v + gridDim.x;
        int nn = 0, nty0 = 0, ntx0 = 0;
        if (vn < total) tile_base(vn, nn, nty0, ntx0);
        if (vn >= total) {       // last tile: the prefetch of "the next tile" re-reads this one (never used) - the
            nn = n;              // loads stay unconditional, so the compiler's vmcnt counts stay exact
            nty0 = ty0;
            ntx0 = tx0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // next channel group of this tile, or group 0 of the next tile, into registers: issued BEHIND the first tap's
            // weight-fragment loads (CONV_STAGE_TAP, round 6) - vector-memory loads retire in order, so the first
            // fragment wait behind these loads waits out their HBM latency; one tap later they have three K steps to land
            // instead of two.
            // Loads stay where they are written: without the compiler-level memory barrier instruction selection
            // places these (unchained, read-only) loads next to their first use, and without the scheduling barrier
            // (vector-memory instructions may not cross, everything else may) the scheduler sinks them there - either
            // way the prefetch is gone.
            auto stage_next = [&]() {
                if (q < 3)
                    fetch(n, ty0, tx0, q + 1);
                else
                    fetch(nn, nty0, ntx0, 0);
                pin_loads();
            };
            if (CONV_STAGE_TAP < 0) stage_next();
            const char *bq = lds + (q & 1) * QBUF + bfrag;
            char *const other = lds + ((q + 1) & 1) * QBUF;
            // B fragments one step (= one row of 32 pixels at one tap, 6 MFMAs) ahead: the wave issues in order, so a
            // fragment read right before its MFMAs waits out the LDS latency with at most two MFMAs in the pipe
            w4 bf[2][2];
            auto read_b = [&](int slot, int step) {
                const int tap = step / NBW, nb = step % NBW;
                const int dy = tap / 3, dx = tap - dy * 3;
                const char *pb = bq + ((nb + dy) * IW + dx) * SLOT;
                bf[slot][0] = *reinterpret_cast<const w4 *>(pb);
                bf[slot][1] = *reinterpret_cast<const w4 *>(pb + 32);
            };
            read_b(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ks = q * 9 + tap;
                fetch_a((ks + NSLOT - 1) % NSLOT, (ks + NSLOT - 1) % 36);
                pin_loads();
                if (tap == CONV_STAGE_TAP) stage_next();
                const int slot = ks % NSLOT;
                const h8 a_hi0 = __builtin_bit_cast(h8, af[slot][0]), a_hi1 = __builtin_bit_cast(h8, af[slot][1]);
                const h8 a_lo0 = __builtin_bit_cast(h8, af[slot][2]), a_lo1 = __builtin_bit_cast(h8, af[slot][3]);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const int step = tap * NBW + nb;
                    if (step + 1 < 9 * NBW) read_b((step + 1) & 1, step + 1);
                    __builtin_amdgcn_sched_barrier(0);   // the reads above are issued before the MFMAs below
                    const h8 b_hi = __builtin_bit_cast(h8, bf[step & 1][0]);
                    const h8 b_lo = __builtin_bit_cast(h8, bf[step & 1][1]);
                    accx[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo0, b_hi, accx[nb][0], 0, 0, 0);
                    accx[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo1, b_hi, accx[nb][1], 0, 0, 0);
                    accx[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi0, b_lo, accx[nb][0], 0, 0, 0);
                    accx[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi1, b_lo, accx[nb][1], 0, 0, 0);
                    acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi0, b_hi, acc[nb][0], 0, 0, 0);
                    acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi1, b_hi, acc[nb][1], 0, 0, 0);
                }
                // the pieces fetched at the start of this group go to the other buffer two per tap from tap 4 on (they
                // have landed: the A fragment waits since tap 3 are behind them in the in-order vmcnt queue), so the
                // LDS writes run under the MFMAs instead of in front of the barrier
                if (tap >= 9 - (NST + 1) / 2) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = 2 * (tap - (9 - (NST + 1) / 2)) + j;
                        if (r < NST && (r < NST - 1 || last_piece)) *reinterpret_cast<w4 *>(other + loff[r]) = st[r];
                    }
                }
            }
            __syncthreads();
        }

        // ---- epilogue: this wave's four rows of 32 pixels ----
        const int px = lane & 31, half = lane >> 5;
        bool sat = false;
#ifdef CONV_ABL_NOEPI   // timing experiment: one store per accumulator so the MFMAs stay live
        {
            float t = 0.f;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[nb][mb][r] + accx[nb][mb][r];
            if (t == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rs_out, lane * 4, 0, 0);
        }
#else
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float y[2][16];      // MODE 0: relu(x) * act_scale, clamped to the f16 range; MODE 1: x
            float ss = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (MODE == 0) {
                        // computed where it is split and stored (below): 32 live registers fewer - with the second
                        // accumulator set this kernel otherwise spills
                    } else {
                        const float t = fmaf(acc[nb][mb][r] + accx[nb][mb][r], inv_scale, bias[ch]);
                        y[mb][r] = t;
                        ss = fmaf(t, t, ss);
                    }
                }
            float nrm = 1.f;
            if (MODE == 1) {
                ss += __shfl_xor(ss, 32);
                nrm = 1.f / sqrtf(fmaxf(ss, 1e-12f));
            }
            // the row passes through the wave's scratch EPX pixels at a time (all 32, or two halves of 16 where two
            // workgroups share a CU's LDS): the lanes owning those pixels write, the wave reads whole records back
            const int row = ty0 + wave * NBW + nb;
            const int so = ((n * Ho + row) * Wo + tx0) * REC;            // wave-uniform
#pragma unroll
            for (int hf = 0; hf < 32 / EPX; ++hf) {
                __builtin_amdgcn_wave_barrier();
                if (EPX == 32 || (px >> 4) == hf) {
                    const int pl = px & (EPX - 1);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int ch = 32 * mb + 8 * g + 4 * half;          // first of 4 consecutive maps
                            if (MODE == 0) {
                                w2 hi, lo;
                                float y4[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int r = 4 * g + q;
                                    const float t = fmaf(acc[nb][mb][r] + accx[nb][mb][r], inv_scale * act_scale,
                                                         bias[ch + q] * act_scale);
                                    // above the records' range, or a NaN of either sign (a negative value is not: the
                                    // ReLU makes it 0) - the same test as conv1_split's
                                    sat |= !(t <= 65504.f);
                                    y4[q] = __builtin_amdgcn_fmed3f(t, 0.f, 65504.f);
                                }
                                split4_scaled(y4, hi, lo);
                                char *p = epi + pl * EPITCH + (ch >> 4) * 64 + (ch & 15) * 2;
                                *reinterpret_cast<w2 *>(p) = hi;
                                *reinterpret_cast<w2 *>(p + 32) = lo;
                            } else {
                                w4 o;
                                o.x = __float_as_uint(y[mb][4 * g + 0] * nrm);
                                o.y = __float_as_uint(y[mb][4 * g + 1] * nrm);
                                o.z = __float_as_uint(y[mb][4 * g + 2] * nrm);
                                o.w = __float_as_uint(y[mb][4 * g + 3] * nrm);
                                *reinterpret_cast<w4 *>(epi + pl * EPITCH + ch * 4) = o;
                            }
                        }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < EPX / 4; ++it) {
                    const int i = it * 64 + lane, pl = i >> 4, part = i & 15, p = hf * EPX + pl;
                    const w4 o = *reinterpret_cast<const w4 *>(epi + pl * EPITCH + part * 16);
                    const bool ok = row < Ho && tx0 + p < Wo;
                    // padded store (common.h): the compiler reuses o's registers for the next piece's index right behind
                    // the store, and the gfx950 store-data hazard also exists - at 4e-7 per store instead of 1e-2 - when
                    // the data came out of LDS (tools/probe/storehazard.hip)
                    buffer_store_b128<0>(o, rs_out, ok ? p * REC + part * 16 : 0x7ffffff0, so);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // a stored activation left the f16 range of the records (|x| * act_scale > 65504): the clamp keeps the data
        // finite, the flag tells the host that this pair's features are not float32-accurate (mccnn.h)
        if (MODE == 0 && sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0)
            atomicOr(sat_flag, 1);
#endif
        if (vn >= total) break;
        v = vn;
        n = nn;
        ty0 = nty0;
        tx0 = ntx0;
    }
}

}  // namespace mccnn

extern "C" size_t mccnn_conv3x3_split_weights_bytes(void) { return (size_t)mccnn::cs::WLAYER; }

extern "C" int mccnn_conv3x3_split_pack(const float *weights, float weight_scale, void *packed, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(weights && packed, MCCNN_E_INVALID, "mccnn_conv3x3_split_pack: null pointer");
    MCCNN_REQUIRE(weight_scale > 0.f, MCCNN_E_INVALID, "mccnn_conv3x3_split_pack: scale must be positive");
    hipLaunchKernelGGL(conv3x3_split_pack_kernel, dim3(cdiv(36 * 2 * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       weights, weight_scale, reinterpret_cast<_Float16 *>(packed));
    return check_launch("mccnn_conv3x3_split_pack");
}

extern "C" int mccnn_conv1_split(const float *images, const float *weights, const float *bias, void *out, int N, int H,
                                 int W, int pad, float act_scale, int *saturation_flag, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(images && weights && bias && out, MCCNN_E_INVALID, "mccnn_conv1_split: null pointer");
    MCCNN_REQUIRE(N > 0 && H > 0 && W > 0 && pad >= 0, MCCNN_E_INVALID, "mccnn_conv1_split: bad size");
    MCCNN_REQUIRE(act_scale > 0.f, MCCNN_E_INVALID, "mccnn_conv1_split: scale must be positive");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    MCCNN_REQUIRE(Ho > 0 && Wo > 0 && Ho <= 65535 && N <= 65535, MCCNN_E_UNSUPPORTED,
                  "mccnn_conv1_split: output %dx%d outside the grid", Wo, Ho);
    hipLaunchKernelGGL(conv1_split_kernel, dim3(cdiv(Wo, 256), Ho, N), dim3(256), 0, (hipStream_t)stream, images,
                       weights, bias, reinterpret_cast<char *>(out), H, W, pad, Ho, Wo, act_scale, saturation_flag);
    return check_launch("mccnn_conv1_split");
}

extern "C" int mccnn_conv3x3_split(const void *in, const void *packed_weights, const float *bias, void *out, int N,
                                   int Hi, int Wi, float weight_scale, float act_scale, int last,
                                   int *saturation_flag, mccnn_stream_t stream)
{
    using namespace mccnn;
    using namespace cs;
    MCCNN_REQUIRE(in && packed_weights && bias && out, MCCNN_E_INVALID, "mccnn_conv3x3_split: null pointer");
    MCCNN_REQUIRE(N > 0 && Hi > 2 && Wi > 2, MCCNN_E_INVALID, "mccnn_conv3x3_split: input %dx%d too small", Wi, Hi);
    MCCNN_REQUIRE(weight_scale > 0.f && act_scale > 0.f, MCCNN_E_INVALID, "mccnn_conv3x3_split: scales must be positive");
    MCCNN_REQUIRE((size_t)N * Hi * Wi * REC <= 0x7ffffff0u, MCCNN_E_UNSUPPORTED,
                  "mccnn_conv3x3_split: %d x %dx%d records exceed 32-bit buffer offsets", N, Wi, Hi);
    const int Ho = Hi - 2, Wo = Wi - 2;
    const int tiles_x = cdiv(Wo, TW), tiles_y = cdiv(Ho, TH);
    const long total = (long)tiles_x * tiles_y * N;
    MCCNN_REQUIRE(total <= 0x7fffffffL, MCCNN_E_UNSUPPORTED, "mccnn_conv3x3_split: too many tiles");
    const int cus = device_cus8();
    const long slots = (long)cus * (LDS_BYTES <= 80 * 1024 ? 2 : 1);     // resident workgroups
    const int grid = (int)(total < slots ? total : slots);
    const float inv = 1.f / (weight_scale * act_scale);
    hipStream_t s = (hipStream_t)stream;
    if (last) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_split_kernel<1>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        MCCNN_REQUIRE(attr == hipSuccess, MCCNN_E_UNSUPPORTED, "mccnn_conv3x3_split: %d bytes of LDS refused", LDS_BYTES);
        hipLaunchKernelGGL(conv3x3_split_kernel<1>, dim3(grid), dim3(256), LDS_BYTES, s,
                           reinterpret_cast<const char *>(in), reinterpret_cast<const char *>(packed_weights), bias,
                           reinterpret_cast<char *>(out), N, Hi, Wi, inv, act_scale, tiles_x, tiles_y, (int)total,
                           saturation_flag);
    } else {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_split_kernel<0>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        MCCNN_REQUIRE(attr == hipSuccess, MCCNN_E_UNSUPPORTED, "mccnn_conv3x3_split: %d bytes of LDS refused", LDS_BYTES);
        hipLaunchKernelGGL(conv3x3_split_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, s,
                           reinterpret_cast<const char *>(in), reinterpret_cast<const char *>(packed_weights), bias,
                           reinterpret_cast<char *>(out), N, Hi, Wi, inv, act_scale, tiles_x, tiles_y, (int)total,
                           saturation_flag);
    }
    return check_launch("mccnn_conv3x3_split");
}
