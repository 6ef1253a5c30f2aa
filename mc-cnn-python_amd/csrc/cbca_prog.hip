// a4 cost_volume_aggregation in the reference's summation order (/root/reference/src/process_functional.py:149-163) on
// pixel-major volumes, program-driven: the per-image control of cbca_hwd_kernel (cbca_hwd.hip) - which region rows a
// K x G patch of anchors sweeps, which pixels each row needs, which arms each anchor adds - is compiled ONCE per image
// into a linear program per patch (cbca_prog_build_kernel below, one wave per patch, csrc/cbca_prog_build.h), and
// the aggregation itself is a threaded-code interpreter written in gfx950 assembly (csrc/asm/cbca_prog_gen.py
// generates it: VGPR index mode, computed entries into straight lines of v_pk_add_f32, no test or branch per region
// element).  Same additions in the same order per (pixel, disparity) as the reference: bit-identical to
// mccnn_cbca_iter_hwd, 18 % faster at 750x500x256 (0.45 vs 0.55 ms per two-volume iteration) because a patch is 4 x 5
// anchors instead of 2 x 5 in fewer registers (relative window addressing): 0.69 x the region rows from L2.
//
// The assembled code objects are embedded in this library (build/asm/cbca_prog_v{2,3,4}.inc, Makefile) and loaded
// through the HIP module API on first use, once per device.
#include <algorithm>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cbca_prog_build.h"
#include "cbca_prog_layout_v2.h"
#include "cbca_prog_layout_v3.h"
#include "cbca_prog_layout_v4.h"
#include "support.h"

namespace mccnn {
namespace prog {

static const Layout kLayouts[3] = {CBCA_PROG_V2_LAYOUT, CBCA_PROG_V3_LAYOUT, CBCA_PROG_V4_LAYOUT};
alignas(4096) static const unsigned char kCodeV2[] = {
#include "cbca_prog_v2.inc"
};
alignas(4096) static const unsigned char kCodeV3[] = {
#include "cbca_prog_v3.inc"
};
alignas(4096) static const unsigned char kCodeV4[] = {
#include "cbca_prog_v4.inc"
};
alignas(4096) static const unsigned char kCodeV2w[] = {
#include "cbca_prog_v2w.inc"
};
alignas(4096) static const unsigned char kCodeV3w[] = {
#include "cbca_prog_v3w.inc"
};
alignas(4096) static const unsigned char kCodeV4w[] = {
#include "cbca_prog_v4w.inc"
};
alignas(4096) static const unsigned char kCodeV2s[] = {
#include "cbca_prog_v2s.inc"
};
alignas(4096) static const unsigned char kCodeV3s[] = {
#include "cbca_prog_v3s.inc"
};
alignas(4096) static const unsigned char kCodeV4s[] = {
#include "cbca_prog_v4s.inc"
};
alignas(4096) static const unsigned char kCodeV2r[] = {
#include "cbca_prog_v2r.inc"
};
alignas(4096) static const unsigned char kCodeV3r[] = {
#include "cbca_prog_v3r.inc"
};
alignas(4096) static const unsigned char kCodeV4r[] = {
#include "cbca_prog_v4r.inc"
};
// [kernel: 0 plain, 1 with WTA, 2 skip, 3 refresh][disparities per lane - 2]
enum { kPlain = 0, kWta = 1, kSkip = 2, kRefresh = 3 };
static const unsigned char *const kCode[4][3] = {{kCodeV2, kCodeV3, kCodeV4}, {kCodeV2w, kCodeV3w, kCodeV4w},
                                                 {kCodeV2s, kCodeV3s, kCodeV4s}, {kCodeV2r, kCodeV3r, kCodeV4r}};
static const char *const kName[4][3] = {{"mccnn_cbca_prog_v2", "mccnn_cbca_prog_v3", "mccnn_cbca_prog_v4"},
                                        {"mccnn_cbca_prog_v2_wta", "mccnn_cbca_prog_v3_wta", "mccnn_cbca_prog_v4_wta"},
                                        {"mccnn_cbca_prog_v2_skip", "mccnn_cbca_prog_v3_skip", "mccnn_cbca_prog_v4_skip"},
                                        {"mccnn_cbca_prog_v2_refresh", "mccnn_cbca_prog_v3_refresh",
                                         "mccnn_cbca_prog_v4_refresh"}};

// disparities per lane: 2 up to 128, 3 where that fills the lanes exactly (padded D a multiple of 3 up to 192), else 4
// with 256-disparity chunks - the same rule as cbca_hwd.hip
static int vpl_of(int Dp)
{
#ifdef MCCNN_PROG_FORCE_VPL      // experiments (tools/build_prog_variant.sh): e.g. 256 disparities as two 128-disparity chunks
    return MCCNN_PROG_FORCE_VPL;
#endif
    return Dp <= 128 ? 2 : (Dp <= 192 && Dp % 3 == 0) ? 3 : 4;
}

struct Loaded {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};
static Loaded g_loaded[64][4][3];
static std::mutex g_mu;

static int kernel_for(int vpl, int which, hipFunction_t *fn)
{
    int dev = 0;
    MCCNN_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, MCCNN_E_UNSUPPORTED,
                  "mccnn_cbca_iter_prog_pair: no current device");
    std::lock_guard<std::mutex> lock(g_mu);
    Loaded &l = g_loaded[dev][which][vpl - 2];
    if (!l.fn) {
        hipError_t e = hipModuleLoadData(&l.mod, kCode[which][vpl - 2]);
        MCCNN_REQUIRE(e == hipSuccess, (int)e, "mccnn_cbca_iter_prog_pair: hipModuleLoadData: %s", hipGetErrorString(e));
        e = hipModuleGetFunction(&l.fn, l.mod, kName[which][vpl - 2]);
        MCCNN_REQUIRE(e == hipSuccess, (int)e, "mccnn_cbca_iter_prog_pair: hipModuleGetFunction(%s): %s",
                      kName[which][vpl - 2], hipGetErrorString(e));
    }
    *fn = l.fn;
    return 0;
}

constexpr int kStageOps = 32;     // ops one sweep step can have with the layouts this library is built with (checked)
struct StageEmitter {
    uint32_t *col;                // this lane's column of the staging area
    int n;
    __device__ __forceinline__ void op(uint32_t v)
    {
        if (n < kStageOps) col[n * 64] = v;
        ++n;
    }
};

// One wave per patch (column group, row group, image); a lane builds one sweep step (a patch has at most
// 2 K + 2 R - 1 = 33) into an LDS staging column, a wave scan turns the lanes' op counts into positions, and the ops
// go out (raw op i lives at dword i + i / 63, a REFILL closes every 64-op chunk).  (One thread per patch took
// 0.91 ms per 750x500 pair: 600 waves of serial, divergent code.)
// mode 0: the full programs; 1: the second set (set_dwords further on: the programs without the anchors whose support
// region is the pixel itself, cbca_prog_build.h unit_region); 2 (round 5): BOTH sets from one pass over the support
// words - the sweep steps of the full program on lanes 0 .., those of the skip program on the lanes behind them (the two
// fit the 64 lanes unless both sweeps are longer than 32 steps: then one after the other), one scan, two outputs.
__global__ __launch_bounds__(64) void cbca_prog_build_kernel(const Layout Larg, const uint32_t *__restrict__ sup0,
                                                             const uint32_t *__restrict__ sup1, uint32_t *__restrict__ prog0,
                                                             uint32_t *__restrict__ prog1, int H, int W, int ngroups,
                                                             int stride, size_t set_dwords, int mode)
{
    // the handler offsets are looked up with per-lane indices: from LDS (a copy of the kernel argument), not from the
    // kernarg segment in memory (a dependent ~0.4 us load per op)
    __shared__ Layout L;
    const int lane = threadIdx.x, cg = blockIdx.x, rg = blockIdx.y, job = blockIdx.z;   // blockIdx.z = image
    {
        const int *src = reinterpret_cast<const int *>(&Larg);
        int *dst = reinterpret_cast<int *>(&L);
        for (int i = lane; i < (int)(sizeof(Layout) / 4); i += 64) dst[i] = src[i];
        __syncthreads();
    }
    const int y0 = rg * L.K;
    if (y0 >= H) return;
    const uint32_t *sup = job ? sup1 : sup0;
    uint32_t *const base = (job ? prog1 : prog0) + ((size_t)rg * ngroups + cg) * stride;
    // the patch's anchors (the same for every lane) and the lanes' work arrays live in LDS, not in scratch memory
    __shared__ Patch P2[2];                    // [0] every anchor, [1] without the unit-region anchors
    __shared__ uint8_t tmp[5 * MAXG][64];
    int nsteps[2];
    {   // patch_setup (cbca_prog_build.h) with one anchor per lane: 20 independent loads instead of 20 in a row
        const int K = L.K, G = L.G, x0 = cg * L.G;
        const int k = lane / G, j = lane - k * G;
        int up = 0, dn = 0;
        bool ok[2] = {false, false};
        int lowest[2] = {y0, y0}, highest[2] = {y0, y0};
        if (lane < K * G) {
            const int y = y0 + k, x = x0 + j;
            if (x < W && y < H) {
                const uint32_t a = sup[(size_t)y * W + x];
                ok[0] = true;
                ok[1] = !unit_region(a);
                const int u = (int)(a & 31u), d = (int)((a >> 5) & 31u);
                up = u < y ? u : y;
                dn = d < H - 1 - y ? d : H - 1 - y;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (ok[s]) {
                        lowest[s] = y - up;
                        highest[s] = dn > 0 ? y + dn : y0;
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                lowest[s] = min(lowest[s], __shfl_xor(lowest[s], d));
                highest[s] = max(highest[s], __shfl_xor(highest[s], d));
            }
            const int nd = y0 + K - 1 - lowest[s] + 1, na = highest[s] - y0;
            if (lane < K * G) P2[s].sched[k][j] = sched_of(ok[s], K, k, ok[s] ? up : 0, ok[s] ? dn : 0, nd);
            if (lane == 0) {
                P2[s].y0 = y0;
                P2[s].x0 = x0;
                P2[s].nd = nd;
                P2[s].na = na;
                P2[s].row0 = y0 - R > 0 ? y0 - R : 0;
            }
            nsteps[s] = nd + na;
        }
    }
    __syncthreads();
    const RowTmpT<uint8_t> T = {{&tmp[0 * MAXG][lane], 64}, {&tmp[1 * MAXG][lane], 64}, {&tmp[2 * MAXG][lane], 64},
                                {&tmp[3 * MAXG][lane], 64}, {&tmp[4 * MAXG][lane], 64}};
    // one pass: a lane's ops go to an LDS staging column first (their position in the program is known only once every
    // lane has counted its own), then a wave scan places them
    __shared__ uint32_t stage[kStageOps][64];
    const uint32_t refill = (uint32_t)L.refill | ((uint32_t)L.M0 << 16);
    const uint32_t endop = (uint32_t)L.end | ((uint32_t)L.M0 << 16);
    // lanes [0, nA) build set sA, lanes [nA, nA + nB) set sB (nB = 0: one set only)
    auto pass = [&](int sA, int nA, int sB, int nB) {
        StageEmitter c = {&stage[0][lane], 0};
        const bool inB = lane >= nA;
        const int sel = inB ? sB : sA, t = inB ? lane - nA : lane;
        if (lane < nA + nB) emit_row(L, P2[sel], sup, W, t, T, c);
        int incl = c.n;                                               // inclusive scan over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const int total = __shfl(incl, 63);
        const int totalA = nA > 0 ? __shfl(incl, nA - 1) : 0;
        uint32_t *out = base + (sel ? set_dwords : 0);
        WriteEmitter w = {out, incl - c.n - (inB ? totalA : 0), stride, refill};
        for (int i = 0; i < c.n; ++i) w.op(stage[i][lane]);
        if (lane == 0) {
            WriteEmitter e = {base + (sA ? set_dwords : 0), totalA, stride, refill};
            e.op(endop);
        }
        if (lane == 1 && sB != sA) {
            WriteEmitter e = {base + (sB ? set_dwords : 0), total - totalA, stride, refill};
            e.op(endop);
        }
    };
    if (mode == 0) {
        pass(0, nsteps[0], 0, 0);
    } else if (mode == 1) {
        pass(1, nsteps[1], 1, 0);
    } else if (nsteps[0] + nsteps[1] <= 64) {
        pass(0, nsteps[0], 1, nsteps[1]);
    } else {
        pass(0, nsteps[0], 0, 0);
        __syncthreads();
        pass(1, nsteps[1], 1, 0);
    }
}

struct Shape {
    int vpl, Dp, nchunks, band_rows, band_groups, ngroups, stride;
};

// What mccnn_cbca_prog_build_pair last wrote where: the interpreter kernel follows whatever its program buffer holds,
// so a buffer that was never built (or built for another shape, or from support arms that have been overwritten since)
// must never reach it - it would not fail, it would run away.
struct Built {
    int D, H, W;
    const void *support;
    unsigned long long gen;       // generation of the support arms the programs were built from
    unsigned long long used;      // last build or successful look-up (eviction is least-recently-used)
};
static std::unordered_map<const void *, Built> g_built[2];      // [0] the full programs, [1] the skip programs
static std::mutex g_built_mu;
static unsigned long long g_built_tick = 0;

static int check_built(const void *prog, const mccnn_support_t *support, int D, int H, int W, int set)
{
    std::lock_guard<std::mutex> lock(g_built_mu);
    const auto it = g_built[set].find(prog);
    MCCNN_REQUIRE(it != g_built[set].end(), MCCNN_E_INVALID,
                  "mccnn_cbca_iter_prog_pair: this program buffer has not been written by mccnn_cbca_prog_build_%spair",
                  set ? "skip_" : "");
    Built &b = it->second;
    b.used = ++g_built_tick;
    MCCNN_REQUIRE(b.D == D && b.H == H && b.W == W, MCCNN_E_INVALID,
                  "mccnn_cbca_iter_prog_pair: programs were built for %dx%dx%d, called with %dx%dx%d", b.W, b.H, b.D, W, H, D);
    MCCNN_REQUIRE(b.support == support && b.gen == support_generation(support), MCCNN_E_INVALID,
                  "mccnn_cbca_iter_prog_pair: programs are stale: mccnn_cross_arms has rewritten the support buffer (or "
                  "another one is passed) since mccnn_cbca_prog_build_pair");
    return 0;
}

// 0 when the program-driven kernels serve this shape (then *s is filled), else the reason as an error code
static int shape_of(int D, int H, int W, Shape *s, const char *who)
{
    MCCNN_REQUIRE(D > 0 && H > 0 && W > 0, MCCNN_E_INVALID, "%s: non-positive size", who);
    const int Dp = mccnn_hwd_pitch(D);
    const int vpl = vpl_of(Dp);
    const Layout &L = kLayouts[vpl - 2];
    // a LOAD op carries a 16-bit pixel index relative to the patch's first region row
    MCCNN_REQUIRE((long)(2 * R + L.K) * W + L.G + 2 * R < 65536, MCCNN_E_UNSUPPORTED,
                  "%s: %d columns exceed the 16-bit pixel index of a program op (use mccnn_cbca_iter_hwd_pair)", who, W);
    MCCNN_REQUIRE((size_t)(2 * R + L.K) * W * Dp * 4 < ((size_t)1 << 31), MCCNN_E_UNSUPPORTED,
                  "%s: %d columns x %d disparities exceed a buffer descriptor's reach", who, W, D);
    MCCNN_REQUIRE((size_t)H * W * 4 < ((size_t)1 << 31), MCCNN_E_UNSUPPORTED, "%s: %dx%d image too large", who, W, H);
    s->vpl = vpl;
    s->Dp = Dp;
    s->nchunks = cdiv(Dp, 64 * vpl);
    s->band_rows = band_rows_of(H, L.K);
    s->band_groups = s->band_rows / L.K;
    s->ngroups = cdiv(W, L.G);
    s->stride = stride_dwords(L.K, L.G, L.W);
    {   // ops of one sweep step: windows + arm runs (cbca_prog_build.h, stride_dwords)
        const int groups = L.K >= 2 ? L.K / 2 : 1, pieces = (R + 1 + L.W - 1) / L.W;
        MCCNN_REQUIRE(pieces * (2 * L.G + 2 * L.G * groups) <= kStageOps, MCCNN_E_UNSUPPORTED,
                      "%s: this layout's sweep steps exceed the builder's staging area", who);
    }
    MCCNN_REQUIRE(s->ngroups <= 65535 && s->nchunks * 2 <= 65535, MCCNN_E_UNSUPPORTED, "%s: %dx%dx%d exceeds the grid", who,
                  W, H, D);
    return 0;
}

static size_t set_bytes(const Shape &s) { return (size_t)8 * s.band_groups * s.ngroups * s.stride * 4; }

}  // namespace prog
}  // namespace mccnn

extern "C" size_t mccnn_cbca_prog_bytes(int D, int H, int W)
{
    using namespace mccnn;
    prog::Shape s;
    if (prog::shape_of(D, H, W, &s, "mccnn_cbca_prog_bytes")) return 0;
    return 2 * prog::set_bytes(s);      // the full programs, then the skip programs
}

// mode 0: the full programs, 1: the skip programs, 2: both sets in one launch
static int prog_build(const char *who, int mode, const mccnn_support_t *support_left, const mccnn_support_t *support_right,
                      int D, int H, int W, int L, void *prog_left, void *prog_right, mccnn_stream_t stream)
{
    using namespace mccnn;
    MCCNN_REQUIRE(support_left && support_right && prog_left && prog_right, MCCNN_E_INVALID, "%s: null pointer", who);
    MCCNN_REQUIRE(L >= 1 && L <= 14, MCCNN_E_UNSUPPORTED, "%s: L=%d outside [1,14]", who, L);
    prog::Shape s;
    int rc = prog::shape_of(D, H, W, &s, who);
    if (rc) return rc;
    rc = check_support_record(support_left, H, W, L, who, true);
    if (rc) return rc;
    rc = check_support_record(support_right, H, W, L, who, true);
    if (rc) return rc;
    MCCNN_REQUIRE(8 * s.band_groups <= 65535, MCCNN_E_UNSUPPORTED, "%s: %d rows exceed the grid", who, H);
    const dim3 grid(s.ngroups, 8 * s.band_groups, 2);
    hipLaunchKernelGGL(prog::cbca_prog_build_kernel, grid, dim3(64), 0, (hipStream_t)stream, prog::kLayouts[s.vpl - 2],
                       reinterpret_cast<const uint32_t *>(support_left), reinterpret_cast<const uint32_t *>(support_right),
                       reinterpret_cast<uint32_t *>(prog_left), reinterpret_cast<uint32_t *>(prog_right), H, W, s.ngroups,
                       s.stride, prog::set_bytes(s) / 4, mode);
    rc = check_launch(who);
    for (int set = 0; rc == 0 && set < 2; ++set) {
        if (mode != 2 && mode != set) continue;
        const unsigned long long gl = support_generation(support_left), gr = support_generation(support_right);
        std::lock_guard<std::mutex> lock(prog::g_built_mu);
        auto &reg = prog::g_built[set];
        // bounded like the support registry: past 4096 entries the least recently used half goes (a look-up by an
        // aggregation launch refreshes an entry, so program buffers that are in use stay)
        if (reg.size() > 4096) {
            std::vector<unsigned long long> used;
            used.reserve(reg.size());
            for (const auto &kv : reg) used.push_back(kv.second.used);
            std::nth_element(used.begin(), used.begin() + used.size() / 2, used.end());
            const unsigned long long keep_from = used[used.size() / 2];
            for (auto it = reg.begin(); it != reg.end();) it = it->second.used < keep_from ? reg.erase(it) : std::next(it);
        }
        reg[prog_left] = prog::Built{D, H, W, support_left, gl, ++prog::g_built_tick};
        reg[prog_right] = prog::Built{D, H, W, support_right, gr, ++prog::g_built_tick};
    }
    return rc;
}

extern "C" int mccnn_cbca_prog_build_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right,
                                          int D, int H, int W, int L, void *prog_left, void *prog_right,
                                          mccnn_stream_t stream)
{
    return prog_build("mccnn_cbca_prog_build_pair", 0, support_left, support_right, D, H, W, L, prog_left, prog_right, stream);
}

extern "C" int mccnn_cbca_prog_build_both_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right,
                                               int D, int H, int W, int L, void *prog_left, void *prog_right,
                                               mccnn_stream_t stream)
{
    return prog_build("mccnn_cbca_prog_build_both_pair", 2, support_left, support_right, D, H, W, L, prog_left, prog_right,
                      stream);
}

extern "C" int mccnn_cbca_prog_build_skip_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right,
                                               int D, int H, int W, int L, void *prog_left, void *prog_right,
                                               mccnn_stream_t stream)
{
    return prog_build("mccnn_cbca_prog_build_skip_pair", 1, support_left, support_right, D, H, W, L, prog_left, prog_right,
                      stream);
}

static int prog_iter(const char *who, const float *in_left, float *out_left, const mccnn_support_t *support_left,
                     const void *prog_left, const float *in_right, float *out_right, const mccnn_support_t *support_right,
                     const void *prog_right, int D, int H, int W, int L, float *disp_left, float *disp_right,
                     int store_right, bool wta, bool skip_unit, mccnn_stream_t stream, bool single = false,
                     bool refresh = false)
{
    using namespace mccnn;
    if (single) {       // one volume per launch: the launch has no second job, its slots repeat the first one's
        in_right = in_left;
        support_right = support_left;
        prog_right = prog_left;
    }
    MCCNN_REQUIRE(in_left && out_left && support_left && prog_left && in_right && support_right && prog_right,
                  MCCNN_E_INVALID, "%s: null pointer", who);
    MCCNN_REQUIRE(single || out_right || (wta && !store_right), MCCNN_E_INVALID, "%s: out_right is null", who);
    MCCNN_REQUIRE(!wta || (disp_left && disp_right), MCCNN_E_INVALID, "%s: null disparity map", who);
    if (!out_right || single) out_right = out_left;     // never dereferenced: an empty descriptor / no second job
    MCCNN_REQUIRE(in_left != out_left && (single || (in_right != out_right && (out_left != out_right || !store_right) &&
                                                     in_left != out_right && in_right != out_left)),
                  MCCNN_E_INVALID, "%s: outputs must not alias an input or each other", who);
    MCCNN_REQUIRE(L >= 1 && L <= 14, MCCNN_E_UNSUPPORTED, "%s: L=%d outside [1,14]", who, L);
    prog::Shape s;
    int rc = prog::shape_of(D, H, W, &s, who);
    if (rc) return rc;
    MCCNN_REQUIRE(!wta || s.nchunks == 1, MCCNN_E_UNSUPPORTED,
                  "%s: D=%d spans more than one chunk of a wave (use mccnn_wta_hwd)", who, D);
    rc = check_support_record(support_left, H, W, L, who, true);
    if (rc) return rc;
    rc = check_support_record(support_right, H, W, L, who, true);
    if (rc) return rc;
    rc = prog::check_built(prog_left, support_left, D, H, W, skip_unit ? 1 : 0);
    if (rc) return rc;
    rc = prog::check_built(prog_right, support_right, D, H, W, skip_unit ? 1 : 0);
    if (rc) return rc;
    hipFunction_t fn;
    rc = prog::kernel_for(s.vpl, wta ? prog::kWta : skip_unit ? prog::kSkip : refresh ? prog::kRefresh : prog::kPlain, &fn);
    if (rc) return rc;
    if (skip_unit) {      // the second program set of both buffers
        prog_left = static_cast<const char *>(prog_left) + prog::set_bytes(s);
        prog_right = static_cast<const char *>(prog_right) + prog::set_bytes(s);
    }
    struct {
        const void *in0, *in1;
        void *out0, *out1;
        const void *prog0, *prog1, *sup0, *sup1;
        int Dp, H, W, nchunks, band_rows, band_groups, prog_stride_bytes, ngroups;
        void *disp0, *disp1;
        int D, store1, pad0, pad1;
    } args = {in_left, in_right, out_left, out_right, prog_left, prog_right, support_left, support_right,
              s.Dp, H, W, s.nchunks, s.band_rows, s.band_groups, s.stride * 4, s.ngroups,
              disp_left, disp_right, D, store_right ? 1 : 0, 0, 0};
    static_assert(sizeof(args) == 0x80, "kernarg layout of csrc/asm/cbca_prog_gen.py");
    size_t size = wta ? 0x80 : 0x60;
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const hipError_t e = hipModuleLaunchKernel(fn, 8 * s.band_groups, s.ngroups, s.nchunks * (single ? 1 : 2), 64, 1, 1, 0,
                                               (hipStream_t)stream, nullptr, extra);
    MCCNN_REQUIRE(e == hipSuccess, (int)e, "%s: %s", who, hipGetErrorString(e));
    return 0;
}

extern "C" int mccnn_cbca_iter_prog_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                         const void *prog_left, const float *in_right, float *out_right,
                                         const mccnn_support_t *support_right, const void *prog_right, int D, int H, int W,
                                         int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_pair", in_left, out_left, support_left, prog_left, in_right, out_right,
                     support_right, prog_right, D, H, W, L, nullptr, nullptr, 1, false, false, stream);
}

extern "C" int mccnn_cbca_iter_prog_pair_skip(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                              const void *prog_left, const float *in_right, float *out_right,
                                              const mccnn_support_t *support_right, const void *prog_right, int D, int H,
                                              int W, int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_pair_skip", in_left, out_left, support_left, prog_left, in_right, out_right,
                     support_right, prog_right, D, H, W, L, nullptr, nullptr, 1, false, true, stream);
}

extern "C" int mccnn_cbca_iter_prog(const float *in, float *out, const mccnn_support_t *support, const void *prog, int D,
                                    int H, int W, int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog", in, out, support, prog, nullptr, nullptr, nullptr, nullptr, D, H, W, L, nullptr,
                     nullptr, 1, false, false, stream, true);
}

extern "C" int mccnn_cbca_iter_prog_skip(const float *in, float *out, const mccnn_support_t *support, const void *prog,
                                         int D, int H, int W, int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_skip", in, out, support, prog, nullptr, nullptr, nullptr, nullptr, D, H, W, L,
                     nullptr, nullptr, 1, false, true, stream, true);
}

extern "C" int mccnn_cbca_iter_prog_pair_wta(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                             const void *prog_left, const float *in_right, float *out_right,
                                             const mccnn_support_t *support_right, const void *prog_right, int D, int H,
                                             int W, int L, float *disparity_left, float *disparity_right, int store_right,
                                             mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_pair_wta", in_left, out_left, support_left, prog_left, in_right, out_right,
                     support_right, prog_right, D, H, W, L, disparity_left, disparity_right, store_right, true, false, stream);
}

// The first iteration of an aggregation whose later iterations run the skip programs (round 6): the full kernel, which
// also writes the value of every unit-region pixel - v1 = (0 + v0) / 1 - back into `in`.
extern "C" int mccnn_cbca_iter_prog_refresh(float *in, float *out, const mccnn_support_t *support, const void *prog, int D,
                                            int H, int W, int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_refresh", in, out, support, prog, nullptr, nullptr, nullptr, nullptr, D, H, W, L,
                     nullptr, nullptr, 1, false, false, stream, true, true);
}

extern "C" int mccnn_cbca_iter_prog_pair_refresh(float *in_left, float *out_left, const mccnn_support_t *support_left,
                                                 const void *prog_left, float *in_right, float *out_right,
                                                 const mccnn_support_t *support_right, const void *prog_right, int D, int H,
                                                 int W, int L, mccnn_stream_t stream)
{
    return prog_iter("mccnn_cbca_iter_prog_pair_refresh", in_left, out_left, support_left, prog_left, in_right, out_right,
                     support_right, prog_right, D, H, W, L, nullptr, nullptr, 1, false, false, stream, false, true);
}
