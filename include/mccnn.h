/*
 * mccnn.h - C ABI of libmccnn_hip.so: the MI355X (gfx950) stereo-matching hot path.
 *
 * The reference (Jackie-Chou/MC-CNN-python) has no FFI layer: its hot path is eleven NumPy functions in
 * src/process_functional.py called by src/match.py:132-175.  Each entry point below replaces the interpreted
 * loop nest of one of those functions; the host-side mirror in mc-cnn-python_amd/src/process_functional.py keeps
 * the reference's Python signatures and calls these through ctypes (see INTEGRATION.md for the binding).
 * "pf:" = /root/reference/src/process_functional.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller; nothing is retained after return
 *   - no allocation, no synchronisation: work is enqueued on `stream` (a hipStream_t passed as void*);
 *     scratch, where needed, is passed in and sized by the matching *_scratch_bytes() query
 *   - return 0 on success, otherwise a negative MCCNN_E_* code or the positive hipError_t of the failed launch;
 *     mccnn_last_error_string() describes the last failure on the calling thread
 *   - float32 everywhere (the reference's dtype); sizes are ints, byte offsets are 64-bit inside
 *   - ONE piece of host-side state, the exception to "nothing is retained": mccnn_cross_arms / mccnn_cross_arms_pair
 *     remember, per support-buffer ADDRESS, the (H, W, L) they built it for (a mutex-guarded process-global map of at
 *     most 4096 entries, cleared when full).  The aggregation entry points consult it only to REFUSE a plane built for
 *     another image size or with longer arms than the L they are told (MCCNN_E_INVALID); addresses it has never seen
 *     pass.  It is keyed by address, not by allocation: a buffer that is freed and whose address is reused keeps its
 *     stale entry until the next mccnn_cross_arms on that address - which every correct use performs before
 *     aggregating with it.  No device memory is referenced by it.
 *
 * HBM layouts
 *   image     [H][W]        float32 (the reference's [H,W,1])
 *   features  [H][W][C]     float32, C = 64 (NET.features, NHWC)
 *   volume    "DHW" [D][H][W]   - the reference's layout; cost volume, the streaming CBCA and its WTA / sub-pixel
 *             "HWD" [H][W][Dp]  - pixel-major, Dp = mccnn_hwd_pitch(D); the SGM scanline kernels, and the whole
 *                                 bit-exact variant from the first aggregation on (mccnn_cbca_iter_hwd, mccnn_wta_hwd,
 *                                 mccnn_subpixel_hwd)
 *   support   mccnn_support_bytes(H,W) bytes: plane 0 [H][W] uint32 words (four 5-bit cross-arm lengths + 12-bit
 *             region size, mccnn_support_t), then four derived planes private to the aggregation kernels
 *   maps      [H][W]        float32 disparity maps, int32 status / region counts
 */
#ifndef MCCNN_H
#define MCCNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *mccnn_stream_t; /* hipStream_t */

#define MCCNN_ABI_VERSION 7 /* 2: window-mask plane, *_hwd entry points; 3: saturation flags; 4: program-driven CBCA; 5: skip programs; 6: one-volume launches; 7: refresh launches, SGM flag planes as a call of their own */

#define MCCNN_E_INVALID (-1)     /* bad argument (null pointer, non-positive size, unsupported shape) */
#define MCCNN_E_UNSUPPORTED (-2) /* shape outside what the kernels were built for (e.g. D > 512 for SGM) */
#define MCCNN_E_SCRATCH (-3)     /* scratch buffer too small */

#define MCCNN_SIDE_LEFT 0  /* the reference's choice == "L" */
#define MCCNN_SIDE_RIGHT 1 /* choice == "R" */

int mccnn_version(void);
const char *mccnn_last_error_string(void);

/* ---- a2  compute_cost_volume (pf:78-113) ------------------------------------------------------------------
 * lcv[d,h,w] = -<fl[h,w,:], fr[h,w-d,:]> for w >= d, border columns filled by the reference's 3-tap mean
 * recurrences, rcv[d,h,w] = lcv[d,h,w+d] (+ its own border recurrence).  Requires 1 <= D <= W-2.
 * mode MCCNN_CV_EXACT reproduces NumPy's pairwise float32 summation order bit for bit (C must be 64);
 * mode MCCNN_CV_MFMA contracts the 64 channels on the matrix cores (every float32 operand as two f16 numbers, three
 * v_mfma_f32_32x32x16_f16 products per K step, float32 accumulation): |diff| <= 2e-6 on unit-norm features - the
 * features must be unit vectors (|x| <= 1), as NET produces them. */
#define MCCNN_CV_EXACT 0
#define MCCNN_CV_MFMA 1
int mccnn_cost_volume(const float *fl, const float *fr, int H, int W, int C, int D, float *lcv, float *rcv, int mode,
                      mccnn_stream_t stream);

/* The same volumes written pixel-major ("HWD" [H][W][Dp], the layout the bit-exact variant keeps from here to WTA),
 * bit-identical to mccnn_cost_volume(mode) followed by mccnn_dhw_to_hwd for d < D (the Dp - D pad entries of a pixel
 * are not written), for MCCNN_CV_EXACT and (ABI 5) MCCNN_CV_MFMA alike; D <= 512. */
int mccnn_cost_volume_hwd(const float *fl, const float *fr, int H, int W, int C, int D, float *lcv_hwd, float *rcv_hwd,
                          int mode, mccnn_stream_t stream);

/* ---- a3  compute_cross_region (pf:571-657) ----------------------------------------------------------------
 * Per pixel: the four arm lengths (<= L-1 per side, anchor-relative threshold |I(q)-I(p)| < tau) and the region
 * size count = sum over the vertical arm of (left+right+1), packed in one 32-bit word so that the aggregation
 * kernel fetches everything about two neighbouring pixels with one 8-byte load:
 *     bits 0-4 up | 5-9 down | 10-14 left | 15-19 right | 20-31 count        (arms <= 31, count <= 63*63)
 * hence L <= 32.  The reference's explicit coordinate list [H][W][(2L)^2][2] (padded with -1) is produced by
 * mccnn_cross_region_list for API compatibility only.
 * The support buffer holds four derived planes behind the first (each 16-byte aligned), private to the kernels
 * of mccnn_cbca_iter / mccnn_cbca_iter_hwd.  Two are written for the streaming kernel's LDS layout: [H][W] uint32 "hsum words" (LDS byte addresses of the two row-prefix
 * entries whose difference is the pixel's horizontal-arm sum) and [H][W] uint64 "emit words" (the float64 reciprocal
 * 1/count with the vertical arms in its 12 low mantissa bits; the 40 upper mantissa bits are chosen so that the word as
 * stored is the float64 nearest to 1/count).  The third lists, for every 16 x 64 pixel tile, its pixels in order of
 * falling region size (uint16 tile-local indices): the order in which the plane-major reference-order kernel deals
 * pixels to lanes, so that a wave's lanes walk regions of similar size.  The fourth, [H][W] uint32 "window masks", is
 * what the pixel-major reference-order kernel (mccnn_cbca_iter_hwd) walks: one bit per column of its register window
 * that the pixel's horizontal arm covers.  Allocate mccnn_support_bytes(H, W) bytes (22 per pixel + alignment);
 * mccnn_cross_arms fills all planes, and consumers that only want the arms / counts read the first H*W words. */
typedef uint32_t mccnn_support_t; /* plane 0 layout [H][W] */
size_t mccnn_support_bytes(int H, int W); /* whole buffer: all planes (0 for non-positive sizes) */
#define MCCNN_SUPPORT_UP(s) ((s) & 31u)
#define MCCNN_SUPPORT_DOWN(s) (((s) >> 5) & 31u)
#define MCCNN_SUPPORT_LEFT(s) (((s) >> 10) & 31u)
#define MCCNN_SUPPORT_RIGHT(s) (((s) >> 15) & 31u)
#define MCCNN_SUPPORT_COUNT(s) ((s) >> 20)
int mccnn_cross_arms(const float *image, int H, int W, float tau, int L, mccnn_support_t *support,
                     mccnn_stream_t stream);
/* Both views in one call (compute_cross_region is called once per image right after each other, pf:120-121): the
 * same results as two mccnn_cross_arms calls, half the launches. */
int mccnn_cross_arms_pair(const float *image_left, const float *image_right, int H, int W, float tau, int L,
                          mccnn_support_t *support_left, mccnn_support_t *support_right, mccnn_stream_t stream);
int mccnn_cross_region_list(const mccnn_support_t *support, int H, int W, int L, int32_t *region,
                            mccnn_stream_t stream);

/* ---- a4  cost_volume_aggregation, ONE iteration on ONE volume (pf:149-163) -----------------------------------
 * out[d,p] = (sum_{q in U(p)} in[d,q]) / count[p];  in != out (ping-pong; the reference does not mutate either).
 * L is the distance threshold the support plane was built with (arms < L; supported: L <= 32); a plane that this
 * library's mccnn_cross_arms built with a larger distance, or for another image size, is refused (MCCNN_E_INVALID).
 * order MCCNN_CBCA_SEPARABLE: horizontal-arm sums then vertical-arm sums evaluated through float64 prefix sums -
 * the correctly rounded region sum, within <= 1e-6 per iteration (O(1) costs) of the reference's sequential float32
 * sum, cost independent of the arm lengths; volumes must be finite (an inf/nan would poison its whole row).
 * MCCNN_CBCA_REFERENCE_ORDER: the reference's flat running sum (vertical arm self,up..,down.. x horizontal arm
 * self,left..,right..), bit-exact, slower. */
#define MCCNN_CBCA_SEPARABLE 0
#define MCCNN_CBCA_REFERENCE_ORDER 1
int mccnn_cbca_iter(const float *in, float *out, const mccnn_support_t *support, int D, int H, int W, int L, int order,
                    mccnn_stream_t stream);

/* One iteration on the left AND the right volume (same D, H, W; each with its own support planes) - the reference's
 * cost_volume_aggregation takes both views in one call (match.py:142, 154; pf:116-180 loops over the two).  Same
 * results as two mccnn_cbca_iter calls; for MCCNN_CBCA_SEPARABLE with L <= 14 it is ONE launch whose workgroups are dealt
 * the work of both volumes (fewer, fuller rounds: 750x500x256 is exactly three rounds of 256 workgroups), every
 * other variant runs the two single-volume launches. */
int mccnn_cbca_iter_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                         const float *in_right, float *out_right, const mccnn_support_t *support_right, int D, int H,
                         int W, int L, int order, mccnn_stream_t stream);

/* The paper's support regions from BOTH views (sec. 4.1; the reference skips it, pf:122-144, 661-729 dead code).
 * Opt-in, changes the output: at disparity d every arm used at a pixel q is min(own arm at q, the other view's arm at
 * the partner q -/+ d) (side LEFT: x - d, RIGHT: x + d; partner outside the image: own arms).  Flat float32 running
 * sum in the reference's order, region size counted on the way.  support_self / support_other: planes built by
 * mccnn_cross_arms on the volume's own view and on the other view. */
int mccnn_cbca_iter_both(const float *in, float *out, const mccnn_support_t *support_self,
                         const mccnn_support_t *support_other, int D, int H, int W, int L, int side,
                         mccnn_stream_t stream);

/* ---- a4 on the pixel-major layout: ONE iteration in the reference's summation order, bit-exact (pf:149-163) ----
 * in_hwd / out_hwd are "HWD" volumes [H][W][Dp]; out[p,d] = the reference's flat float32 running sum over the region
 * list of p (vertical arm self,up..,down.. x horizontal arm self,left..,right..) divided by count[p].  A wave owns a
 * few neighbouring pixels and all their disparities (disparities on lanes), so the region walk runs on the scalar
 * unit and every region element is one coalesced load: same bits as mccnn_cbca_iter(..., MCCNN_CBCA_REFERENCE_ORDER)
 * on the plane-major volume, an order of magnitude faster.  The kernel reads plane 0 of the support buffer AND its
 * window-mask plane (one word per pixel, behind the derived planes): `support` must be the start of the WHOLE
 * mccnn_support_bytes(H, W) buffer that mccnn_cross_arms wrote - a copy of plane 0 alone is refused
 * (MCCNN_E_INVALID: the *_hwd entry points only accept pointers mccnn_cross_arms has written).  L <= 14; in != out.
 * The _pair form takes the left and the right volume in one launch (pf:116-180 loops over the two views). */
int mccnn_cbca_iter_hwd(const float *in_hwd, float *out_hwd, const mccnn_support_t *support, int D, int H, int W, int L,
                        mccnn_stream_t stream);
int mccnn_cbca_iter_hwd_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                             const float *in_right, float *out_right, const mccnn_support_t *support_right, int D,
                             int H, int W, int L, mccnn_stream_t stream);

/* The LAST iteration of an aggregation fused with a7 (pf:239-272): like mccnn_cbca_iter_hwd_pair, and the first strict
 * minimum over d of every pixel of the two results goes to disparity_left / disparity_right ([H][W] float32, -1 where
 * no disparity wins), exactly what mccnn_wta_hwd returns on the stored volumes.  store_right = 0 leaves out_right
 * unwritten (it may then be NULL): match.py needs the final right volume for nothing but its WTA.  One chunk of
 * disparities per wave: D <= 256 (MCCNN_E_UNSUPPORTED beyond; run mccnn_wta_hwd then). */
int mccnn_cbca_iter_hwd_pair_wta(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                 const float *in_right, float *out_right, const mccnn_support_t *support_right, int D,
                                 int H, int W, int L, float *disparity_left, float *disparity_right, int store_right,
                                 mccnn_stream_t stream);

/* ---- a4 on the pixel-major layout, program-driven (pf:149-163; round 4) ---------------------------------------------
 * Bit-identical to mccnn_cbca_iter_hwd_pair and faster (0.45 vs 0.55 ms per two-volume iteration at 750x500x256): the
 * per-image control of that kernel - region rows, pixel windows and arm runs of every 4 x 5 patch of anchors - is
 * compiled once per image into a linear program per patch (mccnn_cbca_prog_build_pair, after mccnn_cross_arms), and the
 * iteration is an interpreter of those programs written in gfx950 assembly (csrc/asm/cbca_prog_gen.py).  The programs
 * depend on the image, on D (disparities per lane) and on nothing else: one build serves all 18 iterations of a pair.
 *   mccnn_cbca_prog_bytes: size of ONE image's program buffer (room for two program sets: the full programs and the
 *     ones mccnn_cbca_iter_prog_pair_skip runs); 0 when the shape is outside what the programs encode
 *     (W > 2180 columns, or volumes beyond a buffer descriptor's reach) - callers then stay with mccnn_cbca_iter_hwd_pair.
 *   support_*: the whole buffers mccnn_cross_arms wrote (plane 0 is read).  L <= 14; outputs must not alias inputs.
 *   mccnn_cbca_iter_prog_pair refuses (MCCNN_E_INVALID) program buffers mccnn_cbca_prog_build_pair has not written, built
 *   for another shape, or built from support arms that mccnn_cross_arms has overwritten since: the kernel follows its
 *   programs blindly. */
size_t mccnn_cbca_prog_bytes(int D, int H, int W);
int mccnn_cbca_prog_build_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right, int D, int H,
                               int W, int L, void *prog_left, void *prog_right, mccnn_stream_t stream);
int mccnn_cbca_iter_prog_pair(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                              const void *prog_left, const float *in_right, float *out_right,
                              const mccnn_support_t *support_right, const void *prog_right, int D, int H, int W, int L,
                              mccnn_stream_t stream);
/* The last iteration fused with a7 (pf:239-272), like mccnn_cbca_iter_hwd_pair_wta: also writes the first strict minimum
 * over d of both results to disparity_left / disparity_right ([H][W] float32, -1 where nothing wins); store_right = 0
 * leaves out_right unwritten (it may then be NULL).  One chunk of disparities per wave: D <= 256. */
int mccnn_cbca_iter_prog_pair_wta(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                  const void *prog_left, const float *in_right, float *out_right,
                                  const mccnn_support_t *support_right, const void *prog_right, int D, int H, int W, int L,
                                  float *disparity_left, float *disparity_right, int store_right, mccnn_stream_t stream);

/* A later iteration of the SAME ping-pong pair of buffers, without the pixels that cannot change any more.  A pixel
 * whose support region is the pixel itself (all four arms 0: count 1) gets v1 = (0 + v0) / 1 in the first iteration
 * (pf:156-161) - v0 except that -0.0 becomes +0.0 and a signalling NaN is quieted - and (0 + v1) / 1 = v1 bit for bit
 * ever after.  This entry point may replace mccnn_cbca_iter_prog_pair from the SECOND consecutive iteration on: it runs
 * the second program set, which mccnn_cbca_prog_build_skip_pair writes into the same buffers (a launch of its own, so
 * that it can run beside the first iterations); such anchors take no part - not loaded for their own sake, not divided,
 * NOT STORED (on the synthetic benchmark pair 46 % of the pixels: 27 % less HBM traffic per iteration).  What the
 * caller must know: out_* keeps, for every such pixel, what it held before.  After a first full iteration in -> out
 * that is v1 in `out` and still v0 in `in`; as an operand of somebody else's sum the two are interchangeable (a sum
 * that began as 0 + x is never -0.0, so adding -0.0 or +0.0 gives the same bits; either NaN gives the same quiet NaN),
 * but a caller that READS such a pixel from the buffer that was the first iteration's input must first run one full
 * iteration into it.  match.py's 16-iteration aggregation (pf:117-183 with max_average_time = 16) therefore runs
 * iteration 1 full, iterations 2 .. 15 this way, and the last one - which carries the WTA and writes every pixel
 * into the first iteration's input buffer - full again (stereo_device.cbca_prog_pair states the rule for any count). */
int mccnn_cbca_prog_build_skip_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right, int D,
                                    int H, int W, int L, void *prog_left, void *prog_right, mccnn_stream_t stream);
/* Both program sets from ONE pass over the support words (round 5): what mccnn_cbca_prog_build_pair followed by
 * mccnn_cbca_prog_build_skip_pair write, word for word, in one launch and about half their time (the sweep steps of the
 * two programs of a patch share a wave's lanes). */
int mccnn_cbca_prog_build_both_pair(const mccnn_support_t *support_left, const mccnn_support_t *support_right, int D,
                                    int H, int W, int L, void *prog_left, void *prog_right, mccnn_stream_t stream);
int mccnn_cbca_iter_prog_pair_skip(const float *in_left, float *out_left, const mccnn_support_t *support_left,
                                   const void *prog_left, const float *in_right, float *out_right,
                                   const mccnn_support_t *support_right, const void *prog_right, int D, int H, int W,
                                   int L, mccnn_stream_t stream);

/* One volume per launch (round 5): the same kernels with half the grid.  A stereo pair's two volumes are independent
 * chains of iterations; issued as two such chains on two streams, one chain's launch fills the compute units that the
 * other chain's launch leaves idle while its last and heaviest patches finish (the skip launches, whose work sits in a
 * few image regions, end with most of the chip idle): 8.7 % less time for 16 iterations at 750x500x256 than one
 * two-volume launch per iteration, same bits.  `prog` is one image's buffer as mccnn_cbca_prog_build_pair /
 * _build_skip_pair wrote it (either position of the pair); the same refusals as the two-volume entry points. */
int mccnn_cbca_iter_prog(const float *in, float *out, const mccnn_support_t *support, const void *prog, int D, int H, int W,
                         int L, mccnn_stream_t stream);
int mccnn_cbca_iter_prog_skip(const float *in, float *out, const mccnn_support_t *support, const void *prog, int D, int H,
                              int W, int L, mccnn_stream_t stream);

/* The FIRST iteration of an aggregation whose later iterations run the skip programs (round 6): exactly
 * mccnn_cbca_iter_prog(_pair) - the full programs, every pixel of `out` written - and, besides, the value v1 = (0 + v0) / 1
 * of every unit-region pixel is written back into `in` (pf:156-161 with aver_num = 1: `in` is NOT const here).  After
 * it both buffers of the ping-pong pair hold v1 at those pixels, so EVERY later iteration may be a _skip launch,
 * whichever buffer the last one writes: without it `in` keeps v0, which differs from v1 where v0 is -0.0, and an
 * aggregation with an even number of iterations (match.py's first: 2) had to end with a full launch.  Waves that read
 * such a pixel as a neighbour while it is being rewritten see v0 or v1 - as an operand the two give the same bits (see
 * above), so the race cannot be observed.  Same programs, same refusals as mccnn_cbca_iter_prog(_pair). */
int mccnn_cbca_iter_prog_refresh(float *in, float *out, const mccnn_support_t *support, const void *prog, int D, int H, int W,
                                 int L, mccnn_stream_t stream);
int mccnn_cbca_iter_prog_pair_refresh(float *in_left, float *out_left, const mccnn_support_t *support_left,
                                      const void *prog_left, float *in_right, float *out_right,
                                      const mccnn_support_t *support_right, const void *prog_right, int D, int H, int W,
                                      int L, mccnn_stream_t stream);

/* ---- layout changes between DHW and HWD ------------------------------------------------------------------- */
int mccnn_hwd_pitch(int D); /* Dp: D rounded up to a multiple of 4 (16-byte rows) */
int mccnn_dhw_to_hwd(const float *dhw, float *hwd, int D, int H, int W, mccnn_stream_t stream);
int mccnn_hwd_to_dhw(const float *hwd, float *dhw, int D, int H, int W, mccnn_stream_t stream);

/* ---- a6  semi_global_matching, one axis-aligned direction, in place (pf:476-568) -----------------------------
 * vol_hwd is updated in place along r = (rh, rw) in {(0,1),(0,-1),(-1,0),(1,0)}; the first line of the scan
 * axis is left untouched.  Penalties are recomputed from the two images (no P1/P2/D2 volumes):
 *   (P1,P2) = (p1,p2) if D1<thr and D2<thr; /q2 if both >= thr; /q1 otherwise      (pf:535-541)
 * p1, p2, q1, q2, thr are the float32 roundings NumPy applies to the Python scalars (pass float(sgm_P1/sgm_V)
 * as p1 for the vertical directions, pf:204).  float32 add/min/sub only, un-fused: bit-exact vs the reference.
 * Up to two volumes (e.g. left + right) are advanced by one launch so the chip sees 2x the scanlines:
 * n_jobs in {1,2}; job j updates vol_hwd[j] as side[j].  scratch >= mccnn_sgm_scratch_bytes(H,W,D). 2<=D<=512. */
size_t mccnn_sgm_scratch_bytes(int H, int W, int D);
int mccnn_sgm_pass(const float *image_left, const float *image_right, float *const *vol_hwd, const int *side,
                   int n_jobs, int D, int H, int W, int rh, int rw, float p1, float p2, float q1, float q2, float thr,
                   void *scratch, size_t scratch_bytes, mccnn_stream_t stream);
/* The two halves of mccnn_sgm_pass as calls of their own (ABI 7, round 6).  The flag planes (pf:504-533: which pixels'
 * intensity step along r reaches thr) depend on the two images, r and thr only: mccnn_sgm_flags writes them into `flags`
 * (>= mccnn_sgm_scratch_bytes(H,W,D)), mccnn_sgm_pass_flagged is the pass itself on planes built for the same H, W, D
 * and r.  A caller that advances the two volumes of a pair in separate launches (StereoMatcher: two free-running chains
 * on two streams) builds the planes of each direction once, off the critical path, and both chains read them. */
int mccnn_sgm_flags(const float *image_left, const float *image_right, int D, int H, int W, int rh, int rw, float thr,
                    void *flags, size_t flags_bytes, mccnn_stream_t stream);
int mccnn_sgm_pass_flagged(float *const *vol_hwd, const int *side, int n_jobs, int D, int H, int W, int rh, int rw, float p1,
                           float p2, float q1, float q2, const void *flags, size_t flags_bytes, mccnn_stream_t stream);

/* The first direction of SGM_average, r = (0,1) (pf:194-195, 216-217), fused with the layout change: reads the
 * plane-major volumes vol_dhw[j] (left untouched) and writes the pixel-major vol_hwd[j], i.e. it replaces
 * mccnn_dhw_to_hwd + mccnn_sgm_pass(rh=0, rw=1) and saves one full read + write of every volume.  2 <= D <= 256. */
int mccnn_sgm_first_pass(const float *image_left, const float *image_right, const float *const *vol_dhw,
                         float *const *vol_hwd, const int *side, int n_jobs, int D, int H, int W, float p1, float p2,
                         float q1, float q2, float thr, void *scratch, size_t scratch_bytes, mccnn_stream_t stream);

/* ---- a7  disparity_prediction, one volume (pf:239-272): first strict minimum over d, as float32 --------------
 * A pixel whose D costs are all NaN / +inf gets -1 (the reference asserts there, pf:253); mccnn_lr_status treats a
 * negative or NaN disparity as an occlusion. */
int mccnn_wta(const float *vol_dhw, int D, int H, int W, float *disparity, mccnn_stream_t stream);

/* The same on a pixel-major volume [H][W][Dp] (one 1 KiB run per pixel at D = 256): same index, same -1 rule. */
int mccnn_wta_hwd(const float *vol_hwd, int D, int H, int W, float *disparity, mccnn_stream_t stream);

/* ---- a8  interpolation (pf:279-378) -------------------------------------------------------------------------
 * mccnn_lr_status: 0 match, 1 mismatch, 2 occlusion (pf:285-307).  mccnn_interpolate: status 1 -> median of the
 * nearest status-0 pixel right/left/below/above, status 2 -> nearest status-0 pixel to the right, else raw. */
int mccnn_lr_status(const float *disp_left, const float *disp_right, int H, int W, int D, int32_t *status,
                    mccnn_stream_t stream);
int mccnn_interpolate(const float *disp_left, const int32_t *status, int H, int W, float *out,
                      mccnn_stream_t stream);
/* The two rules of the paper that the reference names and leaves out (pf:318, pf:361), opt-in: directions = 16 takes
 * the median of the nearest matches along 16 rays instead of the 4 axis directions; occlusion_from_left = 1 fills an
 * occlusion from the nearest match to its left instead of its right.  (4, 0) == mccnn_interpolate. */
int mccnn_interpolate_ex(const float *disp_left, const int32_t *status, int H, int W, int directions,
                         int occlusion_from_left, float *out, mccnn_stream_t stream);

/* ---- a9  subpixel_enhance (pf:381-400), float32 as NumPy 2 evaluates it ------------------------------------- */
int mccnn_subpixel(const float *disp, const float *vol_dhw, int D, int H, int W, float *out, mccnn_stream_t stream);
/* numpy1_promotion = 1: the scalar promotion of NumPy < 2 (the reference's own Python 2.7 environment): the
 * denominator, the quotient and the subtraction of pf:396 run in float64 and are rounded once on the store;
 * differs from the default by <= 2.5e-5 px.  0 == mccnn_subpixel (what the golden vectors pin). */
int mccnn_subpixel_ex(const float *disp, const float *vol_dhw, int D, int H, int W, int numpy1_promotion, float *out,
                      mccnn_stream_t stream);

/* mccnn_subpixel_ex on a pixel-major volume [H][W][Dp]: the same arithmetic, three neighbouring floats per pixel. */
int mccnn_subpixel_hwd(const float *disp, const float *vol_hwd, int D, int H, int W, int numpy1_promotion, float *out,
                       mccnn_stream_t stream);

/* ---- a10 median_filter (pf:403-421): clipped fh x fw window (odd sizes, fh*fw <= 49), np.median ------------- */
int mccnn_median(const float *disp, int H, int W, int fh, int fw, float *out, mccnn_stream_t stream);

/* ---- a11 bilateral_filter (pf:424-470) ----------------------------------------------------------------------
 * table: device [fh][fw] float32 spatial kernel (util.normal evaluated on the host, pf:428-436); thr gates
 * |I(q)-I(p)| < thr.  Sums follow NumPy's pairwise order over the clipped window (pf:463,466): bit-exact. */
int mccnn_bilateral(const float *image, const float *disp, int H, int W, int fh, int fw, const float *table,
                    float thr, float *out, mccnn_stream_t stream);

/* ---- a1 epilogues of the conv stack (model.py:51-64, 111-125) ------------------------------------------------
 * mccnn_bias_act: x[n][c][i] = act(x[n][c][i] + bias[c]) in place on an NCHW tensor (plane = H*W elements) -
 *   tf.nn.bias_add + tf.nn.relu of model.py:118-123 in ONE pass over the activations (relu != 0), or the bias
 *   alone (relu == 0).  The convolution itself runs in PyTorch-ROCm/MIOpen without bias.
 * mccnn_l2norm_chw_to_hwc: tf.nn.l2_normalize over channels (model.py:64), x * rsqrt(max(sum x^2, 1e-12)), fused
 *   with the last layer's bias (bias may be NULL) and the layout change NCHW [C][H][W] -> NHWC [H][W][C]. */
int mccnn_bias_act(float *x, const float *bias, int N, int C, long plane, int relu, mccnn_stream_t stream);
/* First layer fused with the input padding (pf:20-25, model.py:51-53): images [N][H][W] -> out [N][C][H+2pad-2]
 * [W+2pad-2] = relu(conv3x3_valid(zero_pad(image, pad), weights [C][1][3][3]) + bias [C]). */
int mccnn_conv1_pad_bias_relu(const float *images, const float *weights, const float *bias, float *out, int N, int H,
                              int W, int pad, int C, mccnn_stream_t stream);
int mccnn_l2norm_chw_to_hwc(const float *chw, const float *bias, float *hwc, int C, int H, int W,
                            mccnn_stream_t stream);

/* ---- a1 on the matrix cores: split-operand 3x3 convolutions (the default feature path; model.py:51-64) ---------
 * The 64 -> 64 map layers as an implicit GEMM on v_mfma_f32_32x32x16_f16 with every float32 operand carried as two
 * f16 numbers (x * s = hi + lo, 22 significand bits; products hi*hi and hi*lo + lo*hi accumulated in float32 in two
 * accumulator sets that meet once per output): as close to a float64 evaluation as the float32 library convolutions
 * (2.5e-7 .. 3e-7 on the unit feature vectors for both), not bit-identical to them.
 * saturation_flag (device int, may be NULL): set to 1 when a stored activation exceeds the f16 range of the records
 * (|x| * act_scale > 65504; the value is clamped, the features are then NOT float32-accurate).  The caller zeroes
 * it, reads it back after the pair and recomputes with the float32 library path if it is set (match.py does).
 * "Split records": 256 bytes per pixel, [channel group q of 16][hi: 16 x f16 | lo: 16 x f16], pixel-major
 * [N][H][W][256 B]; act_scale (a power of two, e.g. 256) is the factor the stored activations carry (they saturate
 * at |x| * act_scale = 65504).
 * mccnn_conv3x3_split_pack: weights [64][64][3][3] float32 (out, in, ky, kx) -> `packed`
 *   (mccnn_conv3x3_split_weights_bytes() bytes) in MFMA fragment order, scaled by weight_scale (a power of two that
 *   brings max |w| near 1024).
 * mccnn_conv1_split: layer 1 (1 -> 64 maps) fused with the zero padding like mccnn_conv1_pad_bias_relu, writing
 *   split records [N][H+2pad-2][W+2pad-2].
 * mccnn_conv3x3_split: in [N][Hi][Wi] records -> VALID conv + bias; last == 0: ReLU, split records
 *   [N][Hi-2][Wi-2]; last != 0: tf.nn.l2_normalize over the 64 maps, float32 [N][Hi-2][Wi-2][64] (what
 *   mccnn_cost_volume reads). */
size_t mccnn_conv3x3_split_weights_bytes(void);
int mccnn_conv3x3_split_pack(const float *weights, float weight_scale, void *packed, mccnn_stream_t stream);
int mccnn_conv1_split(const float *images, const float *weights, const float *bias, void *out, int N, int H, int W,
                      int pad, float act_scale, int *saturation_flag, mccnn_stream_t stream);
int mccnn_conv3x3_split(const void *in, const void *packed_weights, const float *bias, void *out, int N, int Hi, int Wi,
                        float weight_scale, float act_scale, int last, int *saturation_flag, mccnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MCCNN_H */
