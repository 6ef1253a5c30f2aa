// a6 semi_global_matching (/root/reference/src/process_functional.py:476-568) on gfx950, plus the DHW <-> HWD
// layout changes around it.
//
// Recurrence along r = (rh, rw), one scanline per wavefront, disparities on lanes:
//     m      = min_k V[k, p-r]
//     V[d,p] = (V[d,p] + min(V[d,p-r], V[d-1,p-r]+P1, V[d+1,p-r]+P1, m+P2)) - m          (pf:547-566)
// with (P1,P2)(d,p) in {P, P/Q1, P/Q2} chosen by two threshold tests on image differences (pf:509-541).
// float32 add / min / sub only, never fused: bit-exact against the reference.
//
// Why the pixel-major "HWD" layout: the recurrence consumes, at each step, the whole disparity vector of ONE
// pixel.  In DHW those D floats are H*W*4 bytes apart; in HWD they are one contiguous run (1 KiB at D = 256), so a
// wave issues one global_load_dwordx4 per lane per step (4 disparities per lane), both for horizontal (row) and
// vertical (column) scanlines, and can keep PF steps in flight in registers - no LDS, no barriers.  The four
// passes of SGM_average compose in place on the HWD copy; mccnn_dhw_to_hwd / mccnn_hwd_to_dhw convert once.
//
// Penalty classes without P1/P2/D2 volumes (the reference allocates three [D,H,W] temporaries, pf:504-507):
//   a(p)   = |I_self(p)  - I_self(p-r)|          >= thr   one byte per pixel   ("A plane", from the volume's image)
//   b(d,p) = |I_other(h,x) - I_other(h-rh,x-rw)| >= thr   x = w-d (left volume) or w+d (right volume); 0 where
//            the reference skips (pf:517-518, 530-531)  -> one byte per pixel of the OTHER image ("B plane"),
//            looked up at column w -/+ d.  Planes are padded by >= D columns of the "skipped" value on both sides
//            so the lookup needs no bounds test.  class = a + b: 0 -> P, 1 -> P/Q1, 2 -> P/Q2.
#include "common.h"

namespace mccnn {
// Cache policy of the in-place scanline passes: every voxel is read once and written once per pass and the next pass
// (another direction) comes back to it only after 1.5 GB of other traffic, so both the load and the store carry the
// non-temporal hint (aux bit 1 = nt on gfx940+): 0.342 -> 0.319 ms horizontal, 0.312 -> 0.294 ms vertical.  The same
// hint on the first pass's plane-major gathers (+28 %) or on the layout transposes (+8 % / no change) is a loss.
constexpr int kNT = 2;


// ---- flag planes ----------------------------------------------------------------------------------------------
// blockIdx.z selects the image (0: left, 1: right): both flag planes of a pass in one launch
__global__ __launch_bounds__(256) void sgm_flags_kernel(const float *__restrict__ img_l, const float *__restrict__ img_r,
                                                        int H, int W, int rh, int rw, float thr, int pitch, int pad,
                                                        uint8_t *__restrict__ plane_l, uint8_t *__restrict__ plane_r)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // padded column
    const int h = blockIdx.y;
    if (i >= pitch) return;
    const float *__restrict__ img = blockIdx.z ? img_r : img_l;
    uint8_t *__restrict__ plane = blockIdx.z ? plane_r : plane_l;
    const int x = i - pad;
    float diff = 0.f;  // D2 stays 0 where the reference `continue`s (pf:507)
    const int xp = x - rw, hp = h - rh;
    if (x >= 0 && x < W && xp >= 0 && xp < W && hp >= 0 && hp < H) {
        const float t = img[(size_t)h * W + x] - img[(size_t)hp * W + xp];
        const float sq = t * t;
        diff = sqrtf(sq);  // np.linalg.norm of a 1-vector (pf:512,520)
    }
    plane[(size_t)h * pitch + i] = diff >= thr ? 1 : 0;
}

// ---- wave helpers ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf,
                                                      false));
}

// Single-instruction float minima.  fminf() lowers to v_max (canonicalise) + v_min because clang must honour signalling
// NaNs; cost volumes are finite, so the bare instructions are used (same value for every non-NaN input, and
// min is associative, so min3(a,b,c) then min with d equals the reference's min(min(a,b),min(c,d)), pf:559).
__device__ __forceinline__ float vmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// x = min(x, x from another lane) in ONE instruction (v_min_f32 with a DPP operand; the builtin path costs a v_mov_dpp
// plus the min).  s_nop 1 = the two wait states a DPP read needs after the VALU write of its source; hipcc does not
// pad inside an asm statement.  Lanes without a valid source keep x.
#define MCCNN_DPP_MIN(name, ctrl)                                                                          \
    __device__ __forceinline__ float name(float x)                                                        \
    {                                                                                                      \
        float r = x;                                                                                       \
        asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 " ctrl : "+v"(r) : "v"(x));                               \
        return r;                                                                                          \
    }
MCCNN_DPP_MIN(min_xor1, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
MCCNN_DPP_MIN(min_xor2, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
MCCNN_DPP_MIN(min_hmirror, "row_half_mirror row_mask:0xf bank_mask:0xf")
MCCNN_DPP_MIN(min_mirror, "row_mirror row_mask:0xf bank_mask:0xf")
MCCNN_DPP_MIN(min_bcast15, "row_bcast:15 row_mask:0xa bank_mask:0xf")
MCCNN_DPP_MIN(min_bcast31, "row_bcast:31 row_mask:0xc bank_mask:0xf")

// minimum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ float wave_min(float x)
{
    x = min_xor1(x);
    x = min_xor2(x);
    x = min_hmirror(x);
    x = min_mirror(x);    // every lane of a 16-row holds the row minimum
    x = min_bcast15(x);   // rows 1 and 3 fold in the row below
    x = min_bcast31(x);   // rows 2 and 3 fold in lanes 0..31 -> lane 63 holds the minimum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

struct SgmJob {
    float *vol;             // HWD volume, updated in place
    const uint8_t *aplane;  // flags of the volume's own image
    const uint8_t *bplane;  // flags of the other image
    int dsign;              // -1: x = w - d (left volume), +1: x = w + d (right volume)
};

struct SgmParams {
    SgmJob job[2];
    int D, Dp, H, W, pitch, pad, rh, rw;
    float p1[3], p2[3];  // indexed by class a+b: {P, P/Q1, P/Q2}
};

constexpr float kInf = __builtin_huge_valf();
typedef uint32_t sgm_u32x4 __attribute__((ext_vector_type(4)));

// NG = number of 64 * VPL-disparity groups per lane (lane l of group g owns d = 64 VPL g + VPL l .. + VPL - 1), PF =
// steps in flight, FULL = every lane of every group holds VPL real disparities (D == 64 * VPL * NG): no tail masking.
// VPL = 4 disparities per lane (16-byte accesses); VPL = 3 (12-byte accesses, one group) serves 129 .. 192 disparities
// whose padded count is a multiple of 3 - KITTI's D = 192 then runs on all 64 lanes instead of 48.
// All memory traffic goes through raw buffer instructions: per-lane byte offset (constant for the whole scanline) +
// wave-uniform step offset in an SGPR, so a step spends no VALU on addresses; lanes past the disparity range get an
// out-of-range offset (loads return 0, stores are dropped) instead of a branch.
typedef uint32_t sgm_u32x3 __attribute__((ext_vector_type(3)));
template <int VPL> struct SgmVec {
    float v[VPL];
};
template <int NG, int PF, bool FULL, int VPL = 4>
__global__ __launch_bounds__(64) void sgm_pass_kernel(const SgmParams P)
{
    static_assert(VPL == 4 || (VPL == 3 && NG == 1), "three disparities per lane: one group only");
    typedef SgmVec<VPL> vec;
    const SgmJob J = P.job[blockIdx.y];
    const int lane = threadIdx.x;
    const int line = blockIdx.x;
    const bool horiz = P.rh == 0;
    const int nsteps = horiz ? P.W - 1 : P.H - 1;
    const bool fwd = horiz ? P.rw > 0 : P.rh > 0;
    const int D = P.D;
    // position along the scan axis of step t: pos(t) = fwd ? t : nsteps - t  (non-negative, so it can ride in soffset)
    const unsigned vstride = (horiz ? 1u : (unsigned)P.W) * (unsigned)P.Dp * 4u;   // bytes between scan positions
    const unsigned fstride = horiz ? 1u : (unsigned)P.pitch;
    const size_t line_pix = horiz ? (size_t)line * P.W : (size_t)line;              // pixel at scan position 0
    const size_t line_flag = horiz ? (size_t)line * P.pitch + P.pad : (size_t)P.pad + line;
    const unsigned span = (unsigned)min((size_t)0xFFFFFFFFu, (size_t)nsteps * vstride + (size_t)P.Dp * 4u);
    const __amdgpu_buffer_rsrc_t rs_vol =
        __builtin_amdgcn_make_buffer_rsrc(J.vol + line_pix * P.Dp, 0, (int)span, 0x00020000);
    const unsigned fspan = (unsigned)((size_t)nsteps * fstride + 1u);
    // the B lookups reach up to pad bytes to either side of the pixel: base the descriptor pad bytes early
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(J.aplane + line_flag), 0, (int)fspan, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(J.bplane + line_flag - P.pad), 0, (int)(fspan + 2u * P.pad), 0x00020000);

    constexpr int kDrop = 0x7ffffff0;
    int voff[NG], boff[NG];   // per-lane byte offsets: volume vector, packed B flags (relative to the B descriptor)
    int dlane[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        dlane[g] = g * 64 * VPL + lane * VPL;
        const bool act = FULL || dlane[g] < D;
        voff[g] = act ? 4 * dlane[g] : kDrop;
        // four flag bytes: x = w + d .. w + d + 3 (right volume) or x = w - d - 3 .. w - d (left volume); with three
        // disparities per lane the fourth byte is fetched and ignored
        boff[g] = act ? P.pad + (J.dsign > 0 ? dlane[g] : -dlane[g] - 3) : kDrop;
    }
    const int shl = J.dsign > 0 ? 0 : 24;  // byte j of the packed flags sits at bit 8*j (dsign>0) or 8*(3-j)
    const int sdir = J.dsign > 0 ? 8 : -8;

    auto mask_tail = [&](vec v, int g) {  // disparities >= D behave as +inf (never win a min; their stores are pads)
        if (!FULL) {
            const int d = dlane[g];
#pragma unroll
            for (int j = 0; j < VPL; ++j)
                if (d + j >= D) v.v[j] = kInf;
        }
        return v;
    };
    auto vec_min = [&](const vec &v) {
        float r = v.v[0];
#pragma unroll
        for (int j = 1; j < VPL; ++j) r = vmin(r, v.v[j]);
        return r;
    };
    auto pos = [&](int t) { return (unsigned)(fwd ? t : nsteps - t); };
    auto load_vol = [&](int g, int t) {
        vec r;
        if constexpr (VPL == 4) {
            const sgm_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs_vol, voff[g], pos(t) * vstride, kNT);
            r.v[0] = __uint_as_float(u.x); r.v[1] = __uint_as_float(u.y); r.v[2] = __uint_as_float(u.z); r.v[3] = __uint_as_float(u.w);
        } else {
            const sgm_u32x3 u = __builtin_amdgcn_raw_buffer_load_b96(rs_vol, voff[g], pos(t) * vstride, kNT);
            r.v[0] = __uint_as_float(u.x); r.v[1] = __uint_as_float(u.y); r.v[2] = __uint_as_float(u.z);
        }
        return r;
    };

    vec prev[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) prev[g] = mask_tail(load_vol(g, 0), g);
    float m;
    {
        float lm = kInf;
#pragma unroll
        for (int g = 0; g < NG; ++g) lm = vmin(lm, vec_min(prev[g]));
        m = wave_min(lm);
    }

    vec cbuf[PF][NG];
    uint32_t fbuf[PF][NG];
    uint32_t abuf[PF];
    auto issue = [&](int slot, int t) {
        const unsigned fo = pos(t) * fstride;
        abuf[slot] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_a, 0, fo, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            cbuf[slot][g] = load_vol(g, t);
            fbuf[slot][g] = __builtin_amdgcn_raw_buffer_load_b32(rs_b, boff[g], fo, 0);
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) issue(k, min(1 + k, nsteps));

    for (int t0 = 1; t0 <= nsteps; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            if (t > nsteps) continue;   // (not `break`: a loop with an early exit is only unrolled up to 8 times)
            const int a = __builtin_amdgcn_readfirstlane((int)abuf[k]);
            // class a+b: b = 0 -> index a, b = 1 -> index a+1
            const float p1lo = a ? P.p1[1] : P.p1[0], p1hi = a ? P.p1[2] : P.p1[1];
            const float p2lo = a ? P.p2[1] : P.p2[0], p2hi = a ? P.p2[2] : P.p2[1];
            vec nw[NG];
            float lm = kInf;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const vec pv = prev[g];
                // neighbours d-1 / d+1 across lanes; the ends of the disparity range see +inf (pf:552,566)
                float below = dpp_mov<0x138>(kInf, pv.v[VPL - 1]);  // wave_shr:1 - lane l gets lane l-1
                float above = dpp_mov<0x130>(kInf, pv.v[0]);        // wave_shl:1 - lane l gets lane l+1
                if (NG > 1) {
                    if (g > 0 && lane == 0)
                        below = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prev[g > 0 ? g - 1 : 0].v[VPL - 1]), 63));
                    if (g + 1 < NG && lane == 63)
                        above = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prev[g + 1 < NG ? g + 1 : g].v[0]), 0));
                }
                const vec c = mask_tail(cbuf[k][g], g);
                const uint32_t fb = fbuf[k][g];
                vec o;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const bool bj = (fb >> (shl + j * sdir)) & 1u;
                    const float q1 = bj ? p1hi : p1lo, q2 = bj ? p2hi : p2lo;
                    const float lo = j > 0 ? pv.v[j > 0 ? j - 1 : 0] : below;
                    const float hi = j + 1 < VPL ? pv.v[j + 1 < VPL ? j + 1 : j] : above;
                    const float best = vmin(vmin3(pv.v[j], lo + q1, hi + q1), m + q2);
                    const float sum = c.v[j] + best;
                    o.v[j] = sum - m;
                }
                nw[g] = o;
                if constexpr (VPL == 4) {
                    sgm_u32x4 ou;
                    ou.x = __float_as_uint(o.v[0]); ou.y = __float_as_uint(o.v[1]);
                    ou.z = __float_as_uint(o.v[2]); ou.w = __float_as_uint(o.v[3]);
                    buffer_store_b128<kNT>(ou, rs_vol, voff[g], pos(t) * vstride);
                } else {
                    sgm_u32x3 ou;
                    ou.x = __float_as_uint(o.v[0]); ou.y = __float_as_uint(o.v[1]); ou.z = __float_as_uint(o.v[2]);
                    buffer_store_b96<kNT>(ou, rs_vol, voff[g], pos(t) * vstride);
                }
                lm = vmin(lm, vec_min(o));
            }
            issue(k, min(t + PF, nsteps));   // past the end: a harmless re-read of the last line (keeps the code branch-free)
            m = wave_min(lm);
#pragma unroll
            for (int g = 0; g < NG; ++g) prev[g] = nw[g];
        }
    }
}

// ---- first direction r = (0,1) fused with the DHW -> HWD layout change -------------------------------------------
// SGM_average always starts with the left-to-right pass (pf:194-195), so that pass can read the plane-major volume
// the aggregation left behind and write the pixel-major copy the other three passes work on: one full read + write
// of the volume (dhw_to_hwd) disappears.  A wave still owns one image row.  Its disparity vectors are gathered 16
// columns at a time: 16 x global_load_dwordx4 fetch a [256 d][16 w] tile (each plane row contributes one 64-byte
// segment), registers -> LDS as [w][d], and each step reads its 4 disparities back with one ds_read_b128.  Tiles are
// double buffered and the loads of tile k+2 are in flight while tile k is consumed.  D <= 256.
struct SgmFirstJob {
    const float *src;       // DHW volume (read only)
    float *dst;             // HWD volume (written, including the untouched first column)
    const uint8_t *aplane;
    const uint8_t *bplane;
    int dsign;
};
struct SgmFirstParams {
    SgmFirstJob job[2];
    int D, Dp, H, W, pitch, pad;
    float p1[3], p2[3];
};

template <bool FULL>
__global__ __launch_bounds__(64) void sgm_first_pass_kernel(const SgmFirstParams P)
{
    constexpr int TC = 32;          // columns per tile: 128 bytes of every plane row = whole cache lines
    constexpr int DP = 260;         // LDS pitch of one tile column (floats): 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float tile[TC * DP];
    const SgmFirstJob J = P.job[blockIdx.y];
    const int lane = threadIdx.x;
    const int h = blockIdx.x;
    const int D = P.D, W = P.W;
    const size_t plane = (size_t)P.H * W;
    const int ntiles = (W + TC - 1) / TC;

    // source: descriptor based at row h of plane 0; per-lane offset selects plane d and the 4-column group
    const unsigned src_span = (unsigned)min((size_t)0xFFFFFFFFu, ((size_t)D * plane - (size_t)h * W) * 4);
    const __amdgpu_buffer_rsrc_t rs_src =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(J.src + (size_t)h * W), 0, (int)src_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(
        J.dst + (size_t)h * W * P.Dp, 0, (int)((size_t)W * P.Dp * 4), 0x00020000);
    const size_t line_flag = (size_t)h * P.pitch + P.pad;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(J.aplane + line_flag),
                                                                          0, W, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(J.bplane + line_flag - P.pad), 0, W + 2 * P.pad, 0x00020000);

    constexpr int kDrop = 0x7ffffff0;
    const int dl = lane * 4;
    const bool act = FULL || dl < D;
    const int voff = act ? 4 * dl : kDrop;                                   // HWD vector of this lane
    const int boff = act ? P.pad + (J.dsign > 0 ? dl : -dl - 3) : kDrop;     // packed B flags
    const int shl = J.dsign > 0 ? 0 : 24, sdir = J.dsign > 0 ? 8 : -8;
    // Tile gather: instruction i covers plane rows d = 8 i + (lane >> 3), columns 4 (lane & 7) .. +3 - eight lanes
    // share one 128-byte line, so an instruction asks for 8 whole lines (the 16-column tiles of the first version
    // asked for 64 half lines per instruction and ran into the vector-memory pipe: 0.47 ms).
    const int gr = lane >> 3, gc = (lane & 7) * 4;
    unsigned goff[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int d = 8 * i + gr;
        goff[i] = d < D ? (unsigned)((size_t)d * plane * 4) + 4u * gc : (unsigned)kDrop;
    }
    auto mask_tail = [&](float4 v) {
        if (!FULL) {
            if (dl + 0 >= D) v.x = kInf;
            if (dl + 1 >= D) v.y = kInf;
            if (dl + 2 >= D) v.z = kInf;
            if (dl + 3 >= D) v.w = kInf;
        }
        return v;
    };

    // One tile lives in LDS (being consumed, 32 steps) while the next one is in flight in registers; when the LDS
    // tile is used up the registers are spilled over it.  33 KB of LDS per wave = 4 waves per CU, as many scanline
    // waves as a 750x500 pair offers per CU anyway (1000 waves on 256 CUs).
    sgm_u32x4 ld[32];
    auto issue_tile = [&](int k) {      // tile k -> registers (clamped past the last tile: harmless re-read)
        const unsigned w0 = (unsigned)min(k, ntiles - 1) * TC * 4u;
#pragma unroll
        for (int i = 0; i < 32; ++i) ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, goff[i], w0, 0);
    };
    auto spill_tile = [&]() {           // registers -> LDS [w][d]
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int d = 8 * i + gr;
            tile[(gc + 0) * DP + d] = __uint_as_float(ld[i].x);
            tile[(gc + 1) * DP + d] = __uint_as_float(ld[i].y);
            tile[(gc + 2) * DP + d] = __uint_as_float(ld[i].z);
            tile[(gc + 3) * DP + d] = __uint_as_float(ld[i].w);
        }
    };

    uint32_t fb[TC], fa[TC];    // flags of the tile consumed next
    auto issue_flags = [&](int k) {
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            const unsigned w = (unsigned)min(k * TC + c, W - 1);
            fa[c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_a, 0, w, 0);
            fb[c] = __builtin_amdgcn_raw_buffer_load_b32(rs_b, boff, w, 0);
        }
    };

    issue_tile(0);
    issue_flags(0);
    float4 prev = make_float4(kInf, kInf, kInf, kInf);
    float m = 0.f;
    for (int k = 0; k < ntiles; ++k) {
        uint32_t cfa_[TC], cfb_[TC];
#pragma unroll
        for (int c = 0; c < TC; ++c) { cfa_[c] = fa[c]; cfb_[c] = fb[c]; }
        __syncthreads();   // single-wave workgroup: the previous tile's LDS reads are done
        spill_tile();
        issue_tile(k + 1);
        issue_flags(k + 1);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < TC; ++c) {
            const int w = k * TC + c;
            if (w >= W) continue;
            const uint32_t cfa = cfa_[c], cfb = cfb_[c];
            const float4 cv = mask_tail(*reinterpret_cast<const float4 *>(&tile[c * DP + dl]));
            float4 o;
            if (w == 0) {
                o = cv;     // the first line of the scan is untouched (it only seeds the recurrence)
            } else {
                const int a = __builtin_amdgcn_readfirstlane((int)cfa);
                const float p1lo = a ? P.p1[1] : P.p1[0], p1hi = a ? P.p1[2] : P.p1[1];
                const float p2lo = a ? P.p2[1] : P.p2[0], p2hi = a ? P.p2[2] : P.p2[1];
                const float below = dpp_mov<0x138>(kInf, prev.w);
                const float above = dpp_mov<0x130>(kInf, prev.x);
                const uint32_t f = cfb;
                const bool b0 = (f >> shl) & 1u, b1 = (f >> (shl + sdir)) & 1u, b2 = (f >> (shl + 2 * sdir)) & 1u,
                           b3 = (f >> (shl + 3 * sdir)) & 1u;
                {
                    const float q1 = b0 ? p1hi : p1lo, q2 = b0 ? p2hi : p2lo;
                    const float s = cv.x + vmin(vmin3(prev.x, below + q1, prev.y + q1), m + q2);
                    o.x = s - m;
                }
                {
                    const float q1 = b1 ? p1hi : p1lo, q2 = b1 ? p2hi : p2lo;
                    const float s = cv.y + vmin(vmin3(prev.y, prev.x + q1, prev.z + q1), m + q2);
                    o.y = s - m;
                }
                {
                    const float q1 = b2 ? p1hi : p1lo, q2 = b2 ? p2hi : p2lo;
                    const float s = cv.z + vmin(vmin3(prev.z, prev.y + q1, prev.w + q1), m + q2);
                    o.z = s - m;
                }
                {
                    const float q1 = b3 ? p1hi : p1lo, q2 = b3 ? p2hi : p2lo;
                    const float s = cv.w + vmin(vmin3(prev.w, prev.z + q1, above + q1), m + q2);
                    o.w = s - m;
                }
            }
            sgm_u32x4 ou;
            ou.x = __float_as_uint(o.x); ou.y = __float_as_uint(o.y);
            ou.z = __float_as_uint(o.z); ou.w = __float_as_uint(o.w);
            buffer_store_b128<0>(ou, rs_dst, voff, (unsigned)w * (unsigned)P.Dp * 4u);
            prev = o;
            m = wave_min(vmin(vmin(o.x, o.y), vmin(o.z, o.w)));
        }
    }
}

// ---- DHW <-> HWD: a [D x N] <-> [N x Dp] matrix transpose through a padded 64x64 LDS tile ------------------------
__global__ __launch_bounds__(256) void dhw_to_hwd_kernel(const float *__restrict__ dhw, float *__restrict__ hwd, int D,
                                                         long N, int Dp)
{
    __shared__ float tile[64][65];
    const long n0 = (long)blockIdx.x * 64;
    const int d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {  // read rows of 64 pixels of plane d0 + r
        const int d = d0 + r;
        const long n = n0 + tx;
        tile[r][tx] = (d < D && n < N) ? dhw[(size_t)d * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {  // write rows of 64 disparities of pixel n0 + r
        const long n = n0 + r;
        const int d = d0 + tx;
        if (n < N && d < Dp) hwd[(size_t)n * Dp + d] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void hwd_to_dhw_kernel(const float *__restrict__ hwd, float *__restrict__ dhw, int D,
                                                         long N, int Dp)
{
    __shared__ float tile[64][65];
    const long n0 = (long)blockIdx.x * 64;
    const int d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const long n = n0 + r;
        const int d = d0 + tx;
        tile[r][tx] = (n < N && d < D) ? hwd[(size_t)n * Dp + d] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int d = d0 + r;
        const long n = n0 + tx;
        if (d < D && n < N) dhw[(size_t)d * N + n] = tile[tx][r];
    }
}

// out[c][r] = in[r][c] for r < R, c < Cw, with 16-byte accesses on both global sides (the vector-memory pipe costs the
// same per wave instruction whatever its width).  Requires C, ip, op multiples of 4 and 16-byte aligned bases; Rw is
// the number of out columns to write rounded up to 4 (pad columns receive zeros).
template <int TRI>   // input rows per tile = length of the contiguous output runs (in floats)
__global__ __launch_bounds__(256) void transpose_f4_kernel(const float *__restrict__ in, float *__restrict__ out, long R,
                                                           long C, long ip, long op, long Cw, long Rw)
{
    __shared__ float tile[TRI * 65];
    const long r0 = (long)blockIdx.y * TRI, c0 = (long)blockIdx.x * 64;
    {
        const int q = threadIdx.x & 15, p = threadIdx.x >> 4;   // 16 float4 per 64-wide input row, 16 rows per pass
#pragma unroll
        for (int pass = 0; pass < TRI / 16; ++pass) {
            const int r = p + 16 * pass;
            const long gr = r0 + r, gc = c0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < R && gc < C) v = *reinterpret_cast<const float4 *>(in + gr * ip + gc);
            float *t = &tile[r * 65 + 4 * q];
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
    }
    __syncthreads();
    constexpr int QN = TRI / 4;          // float4 per output row of the tile
    constexpr int PN = 256 / QN;         // output rows per pass
    const int q = threadIdx.x % QN, p = threadIdx.x / QN;
#pragma unroll
    for (int pass = 0; pass < 64 / PN; ++pass) {
        const int c = p + PN * pass;
        const long gc = c0 + c, gr = r0 + 4 * q;
        if (gc < Cw && gr < Rw) {
            float4 v;
            v.x = tile[(4 * q + 0) * 65 + c];
            v.y = tile[(4 * q + 1) * 65 + c];
            v.z = tile[(4 * q + 2) * 65 + c];
            v.w = tile[(4 * q + 3) * 65 + c];
            *reinterpret_cast<float4 *>(out + gc * op + gr) = v;
        }
    }
}

static inline int flag_pad(int D) { return (D + 3 + 15) & ~15; }  // >= D+3: the packed 4-byte read may start 3 early

}  // namespace mccnn

extern "C" int mccnn_hwd_pitch(int D) { return (D + 3) & ~3; }

extern "C" int mccnn_dhw_to_hwd(const float *dhw, float *hwd, int D, int H, int W, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(dhw && hwd, MCCNN_E_INVALID, "mccnn_dhw_to_hwd: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_dhw_to_hwd: non-positive size");
    const long N = (long)H * W;
    const int Dp = mccnn_hwd_pitch(D);
    if ((N & 3) == 0 && ((uintptr_t)dhw & 15) == 0 && ((uintptr_t)hwd & 15) == 0)   // in: [D][N], out: [N][Dp]
        hipLaunchKernelGGL(transpose_f4_kernel<64>, dim3(cdiv(N, 64), cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream,
                           dhw, hwd, (long)D, N, N, (long)Dp, N, (long)Dp);
    else
        hipLaunchKernelGGL(dhw_to_hwd_kernel, dim3(cdiv(N, 64), cdiv(Dp, 64)), dim3(256), 0, (hipStream_t)stream, dhw,
                           hwd, D, N, Dp);
    return check_launch("mccnn_dhw_to_hwd");
}

extern "C" int mccnn_hwd_to_dhw(const float *hwd, float *dhw, int D, int H, int W, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(dhw && hwd, MCCNN_E_INVALID, "mccnn_hwd_to_dhw: null pointer");
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "mccnn_hwd_to_dhw: non-positive size");
    const long N = (long)H * W;
    const int Dp = mccnn_hwd_pitch(D);
    if ((N & 3) == 0 && ((uintptr_t)dhw & 15) == 0 && ((uintptr_t)hwd & 15) == 0)   // in: [N][Dp], out: [D][N]
#ifndef SGM_T_BACK
#define SGM_T_BACK 128
#endif
        hipLaunchKernelGGL(transpose_f4_kernel<SGM_T_BACK>, dim3(cdiv(Dp, 64), cdiv(N, SGM_T_BACK)), dim3(256), 0,
                           (hipStream_t)stream, hwd, dhw, N, (long)Dp, (long)Dp, N, (long)D, N);   // 512-byte runs
    else
        hipLaunchKernelGGL(hwd_to_dhw_kernel, dim3(cdiv(N, 64), cdiv(D, 64)), dim3(256), 0, (hipStream_t)stream, hwd,
                           dhw, D, N, Dp);
    return check_launch("mccnn_hwd_to_dhw");
}

extern "C" size_t mccnn_sgm_scratch_bytes(int H, int W, int D)
{
    if (H <= 0 || W <= 0 || D <= 0) return 0;
    const size_t pitch = (size_t)W + 2 * (size_t)mccnn::flag_pad(D);
    return 2 * (size_t)H * pitch + 256;  // one flag plane per image
}

// The flag planes of one direction (both images) into `flags` (layout of the scratch buffer of mccnn_sgm_pass).
static int sgm_launch_flags(const char *who, const float *image_left, const float *image_right, int D, int H, int W, int rh,
                            int rw, float thr, void *flags, size_t flags_bytes, hipStream_t s)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image_left && image_right && flags, MCCNN_E_INVALID, "%s: null pointer", who);
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "%s: non-positive size", who);
    MCCNN_REQUIRE(D >= 2 && D <= 512, MCCNN_E_UNSUPPORTED,
                  "%s: D=%d outside [2,512] (the reference itself needs D >= 2, pf:550)", who, D);
    MCCNN_REQUIRE((rh == 0 && (rw == 1 || rw == -1)) || (rw == 0 && (rh == 1 || rh == -1)), MCCNN_E_INVALID,
                  "%s: r=(%d,%d) is not an axis-aligned unit step (pf:484)", who, rh, rw);
    MCCNN_REQUIRE(flags_bytes >= mccnn_sgm_scratch_bytes(H, W, D), MCCNN_E_SCRATCH, "%s: scratch %zu < %zu bytes", who,
                  flags_bytes, mccnn_sgm_scratch_bytes(H, W, D));
    const int pad = flag_pad(D);
    const int pitch = W + 2 * pad;
    uint8_t *plane_l = reinterpret_cast<uint8_t *>(flags);
    uint8_t *plane_r = plane_l + (((size_t)H * pitch + 127) & ~(size_t)127);
    const dim3 fgrid(cdiv(pitch, 256), H, 2), fblock(256);
    hipLaunchKernelGGL(sgm_flags_kernel, fgrid, fblock, 0, s, image_left, image_right, H, W, rh, rw, thr, pitch, pad,
                       plane_l, plane_r);
    return check_launch(who);
}

// One direction on 1 or 2 volumes with the flag planes already in `flags`.
static int sgm_launch_pass(const char *who, float *const *vol_hwd, const int *side, int n_jobs, int D, int H, int W, int rh,
                           int rw, float p1, float p2, float q1, float q2, const void *flags, size_t flags_bytes,
                           hipStream_t s)
{
    using namespace mccnn;
    MCCNN_REQUIRE(vol_hwd && side && flags, MCCNN_E_INVALID, "%s: null pointer", who);
    MCCNN_REQUIRE(n_jobs == 1 || n_jobs == 2, MCCNN_E_INVALID, "%s: n_jobs=%d must be 1 or 2", who, n_jobs);
    MCCNN_REQUIRE(H > 0 && W > 0, MCCNN_E_INVALID, "%s: non-positive size", who);
    MCCNN_REQUIRE(D >= 2 && D <= 512, MCCNN_E_UNSUPPORTED,
                  "%s: D=%d outside [2,512] (the reference itself needs D >= 2, pf:550)", who, D);
    MCCNN_REQUIRE((rh == 0 && (rw == 1 || rw == -1)) || (rw == 0 && (rh == 1 || rh == -1)), MCCNN_E_INVALID,
                  "%s: r=(%d,%d) is not an axis-aligned unit step (pf:484)", who, rh, rw);
    MCCNN_REQUIRE(flags_bytes >= mccnn_sgm_scratch_bytes(H, W, D), MCCNN_E_SCRATCH, "%s: scratch %zu < %zu bytes", who,
                  flags_bytes, mccnn_sgm_scratch_bytes(H, W, D));
    const int pad = flag_pad(D);
    const int pitch = W + 2 * pad;
    const uint8_t *plane_l = reinterpret_cast<const uint8_t *>(flags);
    const uint8_t *plane_r = plane_l + (((size_t)H * pitch + 127) & ~(size_t)127);
    SgmParams P;
    for (int j = 0; j < 2; ++j) {
        const int jj = j < n_jobs ? j : 0;
        MCCNN_REQUIRE(vol_hwd[jj] != nullptr, MCCNN_E_INVALID, "%s: null volume", who);
        MCCNN_REQUIRE(side[jj] == MCCNN_SIDE_LEFT || side[jj] == MCCNN_SIDE_RIGHT, MCCNN_E_INVALID,
                      "%s: side must be MCCNN_SIDE_LEFT or MCCNN_SIDE_RIGHT", who);
        P.job[j].vol = vol_hwd[jj];
        const bool left = side[jj] == MCCNN_SIDE_LEFT;
        P.job[j].aplane = left ? plane_l : plane_r;
        P.job[j].bplane = left ? plane_r : plane_l;
        P.job[j].dsign = left ? -1 : +1;
    }
    P.D = D; P.Dp = mccnn_hwd_pitch(D); P.H = H; P.W = W; P.pitch = pitch; P.pad = pad; P.rh = rh; P.rw = rw;
    P.p1[0] = p1; P.p1[1] = p1 / q1; P.p1[2] = p1 / q2;  // pf:538-541 (float32 divisions)
    P.p2[0] = p2; P.p2[1] = p2 / q1; P.p2[2] = p2 / q2;
    const int nlines = rh == 0 ? H : W;
    if ((rh == 0 ? W : H) < 2) return 0;  // nothing to scan
    const dim3 grid(nlines, n_jobs), block(64);
    MCCNN_REQUIRE((size_t)H * W * P.Dp * 4 < ((size_t)1 << 32), MCCNN_E_UNSUPPORTED,
                  "%s: %dx%dx%d volume exceeds the 4 GiB reach of a buffer descriptor", who, W, H, D);
    // steps in flight: 8, 12 and 16 measure the same at 750x500x256 (0.30 / 0.29 ms per pass: 1000-1500 scanline waves);
    // a 1242x375 pair has only 750 row scanlines - fewer waves than SIMDs - and its horizontal passes gain from 24 steps
    // (0.388 -> 0.342 ms); two disparity groups per lane (D > 256) take 12 (vertical 2.17 -> 2.06 ms at 1500x1000x400)
#ifndef SGM_PF_PARTIAL
#define SGM_PF_PARTIAL 24
#endif
#ifndef SGM_PF_2G
#define SGM_PF_2G 12
#endif
    if (D == 256)
        hipLaunchKernelGGL((sgm_pass_kernel<1, 16, true>), grid, block, 0, s, P);
    else if (D == 192)                                   // three disparities per lane: all 64 lanes, no tail masks
        hipLaunchKernelGGL((sgm_pass_kernel<1, SGM_PF_PARTIAL, true, 3>), grid, block, 0, s, P);
    else if (D > 128 && D < 192 && P.Dp % 3 == 0)
        hipLaunchKernelGGL((sgm_pass_kernel<1, SGM_PF_PARTIAL, false, 3>), grid, block, 0, s, P);
    else if (D < 256)
        hipLaunchKernelGGL((sgm_pass_kernel<1, SGM_PF_PARTIAL, false>), grid, block, 0, s, P);
    else if (D == 512)
        hipLaunchKernelGGL((sgm_pass_kernel<2, SGM_PF_2G, true>), grid, block, 0, s, P);
    else
        hipLaunchKernelGGL((sgm_pass_kernel<2, SGM_PF_2G, false>), grid, block, 0, s, P);
    return check_launch(who);
}

extern "C" int mccnn_sgm_pass(const float *image_left, const float *image_right, float *const *vol_hwd, const int *side,
                              int n_jobs, int D, int H, int W, int rh, int rw, float p1, float p2, float q1, float q2,
                              float thr, void *scratch, size_t scratch_bytes, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image_left && image_right && vol_hwd && side && scratch, MCCNN_E_INVALID,
                  "mccnn_sgm_pass: null pointer");
    MCCNN_REQUIRE(n_jobs == 1 || n_jobs == 2, MCCNN_E_INVALID, "mccnn_sgm_pass: n_jobs=%d must be 1 or 2", n_jobs);
    const int rc = sgm_launch_flags("mccnn_sgm_pass", image_left, image_right, D, H, W, rh, rw, thr, scratch, scratch_bytes,
                                    (hipStream_t)stream);
    if (rc) return rc;
    return sgm_launch_pass("mccnn_sgm_pass", vol_hwd, side, n_jobs, D, H, W, rh, rw, p1, p2, q1, q2, scratch, scratch_bytes,
                           (hipStream_t)stream);
}

// The two halves of mccnn_sgm_pass as calls of their own (round 6): the flag planes depend on the images, the direction
// and the threshold only - a caller that advances the two volumes of a pair in separate launches (or on separate
// streams) builds them once per direction and hands them to every pass.
extern "C" int mccnn_sgm_flags(const float *image_left, const float *image_right, int D, int H, int W, int rh, int rw,
                               float thr, void *flags, size_t flags_bytes, mccnn_stream_t stream)
{
    return sgm_launch_flags("mccnn_sgm_flags", image_left, image_right, D, H, W, rh, rw, thr, flags, flags_bytes,
                            (hipStream_t)stream);
}

extern "C" int mccnn_sgm_pass_flagged(float *const *vol_hwd, const int *side, int n_jobs, int D, int H, int W, int rh, int rw,
                                      float p1, float p2, float q1, float q2, const void *flags, size_t flags_bytes,
                                      mccnn_stream_t stream)
{
    return sgm_launch_pass("mccnn_sgm_pass_flagged", vol_hwd, side, n_jobs, D, H, W, rh, rw, p1, p2, q1, q2, flags,
                           flags_bytes, (hipStream_t)stream);
}

extern "C" int mccnn_sgm_first_pass(const float *image_left, const float *image_right, const float *const *vol_dhw,
                                    float *const *vol_hwd, const int *side, int n_jobs, int D, int H, int W, float p1,
                                    float p2, float q1, float q2, float thr, void *scratch, size_t scratch_bytes,
                                    mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(image_left && image_right && vol_dhw && vol_hwd && side && scratch, MCCNN_E_INVALID,
                  "mccnn_sgm_first_pass: null pointer");
    MCCNN_REQUIRE(n_jobs == 1 || n_jobs == 2, MCCNN_E_INVALID, "mccnn_sgm_first_pass: n_jobs=%d must be 1 or 2", n_jobs);
    MCCNN_REQUIRE(H > 0 && W > 1, MCCNN_E_INVALID, "mccnn_sgm_first_pass: bad size");
    MCCNN_REQUIRE(D >= 2 && D <= 256, MCCNN_E_UNSUPPORTED,
                  "mccnn_sgm_first_pass: D=%d outside [2,256]; use mccnn_dhw_to_hwd + mccnn_sgm_pass", D);
    MCCNN_REQUIRE((size_t)D * H * W * 4 < ((size_t)1 << 32), MCCNN_E_UNSUPPORTED,
                  "mccnn_sgm_first_pass: volume exceeds the 4 GiB reach of a buffer descriptor");
    MCCNN_REQUIRE(scratch_bytes >= mccnn_sgm_scratch_bytes(H, W, D), MCCNN_E_SCRATCH,
                  "mccnn_sgm_first_pass: scratch %zu < %zu bytes", scratch_bytes, mccnn_sgm_scratch_bytes(H, W, D));
    hipStream_t s = (hipStream_t)stream;
    const int pad = flag_pad(D);
    const int pitch = W + 2 * pad;
    uint8_t *plane_l = reinterpret_cast<uint8_t *>(scratch);
    uint8_t *plane_r = plane_l + (((size_t)H * pitch + 127) & ~(size_t)127);
    const dim3 fgrid(cdiv(pitch, 256), H, 2), fblock(256);
    hipLaunchKernelGGL(sgm_flags_kernel, fgrid, fblock, 0, s, image_left, image_right, H, W, 0, 1, thr, pitch, pad,
                       plane_l, plane_r);
    int rc = check_launch("mccnn_sgm_first_pass(flags)");
    if (rc) return rc;
    SgmFirstParams P;
    for (int j = 0; j < 2; ++j) {
        const int jj = j < n_jobs ? j : 0;
        MCCNN_REQUIRE(vol_dhw[jj] && vol_hwd[jj], MCCNN_E_INVALID, "mccnn_sgm_first_pass: null volume");
        MCCNN_REQUIRE(side[jj] == MCCNN_SIDE_LEFT || side[jj] == MCCNN_SIDE_RIGHT, MCCNN_E_INVALID,
                      "mccnn_sgm_first_pass: side must be MCCNN_SIDE_LEFT or MCCNN_SIDE_RIGHT");
        const bool left = side[jj] == MCCNN_SIDE_LEFT;
        P.job[j].src = vol_dhw[jj];
        P.job[j].dst = vol_hwd[jj];
        P.job[j].aplane = left ? plane_l : plane_r;
        P.job[j].bplane = left ? plane_r : plane_l;
        P.job[j].dsign = left ? -1 : +1;
    }
    P.D = D; P.Dp = mccnn_hwd_pitch(D); P.H = H; P.W = W; P.pitch = pitch; P.pad = pad;
    P.p1[0] = p1; P.p1[1] = p1 / q1; P.p1[2] = p1 / q2;
    P.p2[0] = p2; P.p2[1] = p2 / q1; P.p2[2] = p2 / q2;
    const dim3 grid(H, n_jobs), block(64);
    if (D == 256)
        hipLaunchKernelGGL((sgm_first_pass_kernel<true>), grid, block, 0, s, P);
    else
        hipLaunchKernelGGL((sgm_first_pass_kernel<false>), grid, block, 0, s, P);
    return check_launch("mccnn_sgm_first_pass");
}
