// Fused multi-scale view-synthesis loss (forward, backward, forward+backward) for gfx950.
//
// Restates the loop of SFMLearner.__call__, models/base_model.py:69-124 (citations into
// pfnet/sfm-learner-chainer), per (sample b, scale s, source i, pixel):
//   depth = 1/disp (:60) -> projective_inverse_warp (:90-94, models/transform.py:156-193)
//   -> |I^ - I| with the all-channels-zero mask (:95-100,:111) -> SSIM (:112-115,:126-142)
//   -> smoothness of the disparity (:75-77,:169-185 or :144-155) -> explainability (:103-109)
// and the hand-derived backward to disp, pose, mask logits and (optionally) the source image.
//
// Execution model ("wave strip"): one 64-lane wavefront owns a strip of one image at one scale:
// 64 consecutive columns (the outer HL/HR lanes are halo) and `chunk_rows` rows (+ halo rows).
// It walks down its rows keeping the last rows in registers, so that
//   * horizontal neighbours come from DPP wave shifts (no LDS, no barrier),
//   * vertical neighbours come from a register ring,
//   * the 3x3 SSIM pools and their transposes are separable sums over those two,
//   * depth gradients of all sources are summed in a wave-private LDS tile and d_disp is
//     written exactly once, coalesced; the 12 sums of dL/dPm are reduced in-wavefront.
// All scales, sources and samples of a step are covered by ONE launch of loss_kernel (every wave builds the geometry of its
// passes itself: sfm_ssim_pass.h, build_pass_geom), followed by a tiny one: finalize_kernel reduces the per-wave partials in a
// fixed order (bitwise reproducible) and finishes d_pose.
#include <stdlib.h>
#include <string.h>

#include <hip/hip_ext.h>

#include "sfm_common.h"
#include "sfm_ssim_pass.h"

namespace sfm {

constexpr int MAX_CHUNK_ROWS = 28;   // + 4 halo rows = the 32 steps a pass can have (StepMasks)
constexpr int MIN_CHUNK_ROWS = 4;
constexpr int WAVES_PER_BLOCK = 1;   // independent wavefronts; grouped only so that a CU is filled with few workgroups

struct ScaleArgs {
  const float* tgt;
  const float* src;
  const float* disp;
  const float* mlog;
  float* d_disp;
  float* d_mask;
  float* d_src;
  float* warped;               // optional output (B,n_src,3,h,w): the warped sources, base_model.py:90-94
  int h, w, strips, chunks, tiles, item_begin, chunk_rows;
  float inv_cnt;               // 1 / (norm_B * 3 * h * w)                 base_model.py:111,115
  float c_dx2, c_dy2, c_dxy;   // smooth_reg / 2^s / element count         base_model.py:76,184-185
  float c_ex, c_ey;            // the same for the edge-aware form         base_model.py:154-155
  float c_exp;                 // exp_reg / (norm_B * h * w)               base_model.py:105,167
  // uniform factors of the backward, products with the upstream gradient gy (set_gy): kept as kernel arguments so that they
  // are scalar operands -- computed in the kernel they would be wave-uniform values held in vector registers
  float k_pix;                 // gy (1-alpha) inv_cnt          dL/d(sum |e|)        base_model.py:111,117
  float kq;                    // -gy alpha inv_cnt             -dL/d(sum ssim) = 2 kappa of a pixel  base_model.py:115,117,142
  float k_exp;                 // gy c_exp                                            base_model.py:105,167
};

struct LossArgs {
  // Header: what every wave of every kernel needs before anything else, in two 64-byte lines.  The argument block lives in memory
  // the scalar cache has not seen when a launch starts; every further line a wave touches before its first data load is another
  // round trip in front of it (0.2 - 0.7 us each: tools/trace_finalize.py), and all waves of a launch wait for it together.
  int B, n_src, n_scales, items;
  int simds_per_xcd;           // SIMDs of one XCD (dispatch rounds -> age rank, see loss_kernel)
  int prio_top;                // resident waves per SIMD - 1, at most 3
  unsigned prio_tab;           // issue priority levels: 2 bits per (phase, rank), phase = first / second half of the sources
  float alpha;                 // ssim_rate
  int tiles_of[SFM_MAX_SCALES];        // sc[s].tiles, 0 beyond n_scales
  int item_begin_of[SFM_MAX_SCALES];   // sc[s].item_begin
  const float* intrinsics;
  float* part_loss;  // [items][4]   pixel, ssim, smooth, exp
  float* part_gpm;   // [items][n_src][12]
  float gy;          // upstream gradient on total_loss
  unsigned long long* trace;   // diagnostics: per item {t_start, t_end (100 MHz), HW_ID, XCC_ID}; normally nullptr
  const float* pose[SFM_MAX_SRC];
  float* d_pose[SFM_MAX_SRC];
  ScaleArgs sc[SFM_MAX_SCALES];
};

template <bool SSIM, bool GRAD, int SMODE>
struct Halo {
  static constexpr int HS = SSIM ? (GRAD ? 2 : 1) : 0;              // reach of the photometric pass
  static constexpr int HM = SMODE == 1 ? 2 : (SMODE == 2 ? 1 : 0);  // reach of the smoothness stencil
  static constexpr int HR = HS > HM ? HS : HM;                      // right halo lanes
  static constexpr int HL = GRAD ? HR : HS;                         // left halo lanes (forward smoothness terms look right/down only)
  static constexpr int SW = 64 - HL - HR;                           // output columns per strip
};

static int strip_width(bool ssim, bool grad, int smode) {
  const int hs = ssim ? (grad ? 2 : 1) : 0;
  const int hm = smode == 1 ? 2 : (smode == 2 ? 1 : 0);
  const int hr = hs > hm ? hs : hm;
  const int hl = grad ? hr : hs;
  return 64 - hl - hr;
}

// ------------------------------------------------------------------------------------------
// smoothness passes (one per wave, before the sources)
// ------------------------------------------------------------------------------------------
// the wave-private d_disp tile: the first contribution of a wave is a plain store, later ones read-add-write
// (see geometry_backward: ds_add_f32 is the slower way)
__device__ __forceinline__ void tile_put(float* p, const float v, const bool add) {
  if (add) *p = *p + v;
  else *p = v;
}

// second-order, models/base_model.py:169-185
template <bool GRAD, bool LOSS>
__device__ __forceinline__ void smooth2_pass(const LossArgs& A, const ScaleArgs& S, const float* __restrict__ dplane, int lane,
                                             int x, bool xin, bool outl, int y0, int y1, float* gacc, float& acc_sm, const bool add) {
  // Every term of compute_smooth_loss is a forward difference anchored at one pixel (a, x):
  //   dx2(a,x)  = d(a,x+2) - 2 d(a,x+1) + d(a,x)                      valid x <= w-3
  //   dy2(a,x)  = d(a+2,x) - 2 d(a+1,x) + d(a,x)                      valid a <= h-3
  //   dxdy, dydx(a,x): the two evaluation orders of the mixed difference  valid a <= h-2, x <= w-2
  // The walk computes the anchored SIGNS of a row once (rows a, a+1, a+2 in registers) and keeps the two
  // previous rows' signs in a ring; the gradient at (q,x) gathers them with the transposed stencil:
  //   c_dx2 [s2x(q,x-2) - 2 s2x(q,x-1) + s2x(q,x)] + c_dy2 [s2y(q-2,x) - 2 s2y(q-1,x) + s2y(q,x)]
  //   + c_dxy [txy(q-1,x-1) - txy(q-1,x) - txy(q,x-1) + txy(q,x)]
  const int h = S.h, w = S.w;
  const bool vx2 = xin && (x <= w - 3);
  const bool vx1 = xin && (x <= w - 2);
  const unsigned xc = (unsigned)min(max(x, 0), w - 1);
  // always a load, from a row clamped into the image (a load under a branch would make the compiler drain every
  // outstanding load at the join); values of rows outside the image only feed terms that are masked out
  auto ldrow = [&](int r) -> float { return ldf(dplane, (unsigned)min(max(r, 0), h - 1) * (unsigned)w + xc); };
  // loop-invariant coefficients pinned in vector registers (the scalar file is full)
  float c_dx2 = S.c_dx2, c_dy2 = S.c_dy2, c_dxy = S.c_dxy, gyv = A.gy;
  asm volatile("" : "+v"(c_dx2), "+v"(c_dy2), "+v"(c_dxy), "+v"(gyv));
  float s2y_m1 = 0.f, s2y_m2 = 0.f, txy_m1 = 0.f;   // anchored signs of rows a-1, a-2
  // one row of the walk: d0, dp1 = rows a, a+1 (masked), q = row a+2 as loaded (masked in place); afterwards the register of
  // row a receives row a+5
  auto row = [&](const int a, float& d0, float& dp1, float& q) {
    q = xin ? q : 0.f;
    const float dp2 = q;
    const float dxr0 = from_right(d0) - d0;        // dx(a,x)
    const float dx2 = from_right(dxr0) - dxr0;     // dx2(a,x)
    const float dy0 = dp1 - d0, dy1 = dp2 - dp1;   // dy(a,x), dy(a+1,x)
    const float dy2 = dy1 - dy0;                   // dy2(a,x)
    const float dxrp = from_right(dp1) - dp1;      // dx(a+1,x)
    const float dxdy0 = dxrp - dxr0;               // dxdy(a,x) = dx(a+1,x) - dx(a,x)
    const float dydx0 = from_right(dy0) - dy0;     // dydx(a,x) = dy(a,x+1) - dy(a,x)
    d0 = ldrow(a + 5);                             // the register of row a is free now
    const bool va = a >= 0, va2 = va && a <= h - 3, va1 = va && a <= h - 2;   // uniform
    if (LOSS) {
      if (a >= y0) {
        float t = 0.f;
        if (vx2) t += c_dx2 * fabsf(dx2);
        if (va2) t += c_dy2 * fabsf(dy2);
        if (vx1 && va1) t += c_dxy * (fabsf(dxdy0) + fabsf(dydx0));
        if (outl) acc_sm += t;
      }
    }
    if (GRAD) {
      const float s2y = va2 ? signf(dy2) : 0.f;
      const float txy = (vx1 && va1) ? signf(dxdy0) + signf(dydx0) : 0.f;
      if (a >= y0) {
        const float s2x = (vx2 && va) ? signf(dx2) : 0.f;
        const float s2x1 = from_left(s2x);
        const float gx2 = from_left(s2x1) - 2.f * s2x1 + s2x;
        const float gy2 = s2y_m2 - 2.f * s2y_m1 + s2y;
        const float gxy = from_left(txy_m1) - txy_m1 - from_left(txy) + txy;
        tile_put(gacc + (a - y0) * 64 + lane, gyv * (c_dx2 * gx2 + c_dy2 * gy2 + c_dxy * gxy), add);
      }
      s2y_m2 = s2y_m1; s2y_m1 = s2y; txy_m1 = txy;
    }
  };
  // The rows a .. a+4 of the column live in five registers used in rotation (the body is instantiated five times: no register
  // is moved, so the wait for a load sits where its value is first used, three rows later; a rotation through moves made
  // every row wait for the load it had just issued).
  float r0 = xin ? ldrow(y0 - 2) : 0.f, r1 = xin ? ldrow(y0 - 1) : 0.f;
  float r2 = ldrow(y0), r3 = ldrow(y0 + 1), r4 = ldrow(y0 + 2);
  for (int a = y0 - 2; a < y1; a += 5) {
    row(a, r0, r1, r2);
    if (a + 1 < y1) row(a + 1, r1, r2, r3);
    if (a + 2 < y1) row(a + 2, r2, r3, r4);
    if (a + 3 < y1) row(a + 3, r3, r4, r0);
    if (a + 4 < y1) row(a + 4, r4, r0, r1);
  }
}

// edge-aware first-order, models/base_model.py:144-155 (commented out at :78-80 in the reference)
//   x term anchored at (q,x):  c_ex |d(q,x+1) - d(q,x)| wx(q,x),   wx = exp(-|mean_c (I(q,x+1) - I(q,x))|)
//   y term anchored at (q,x):  c_ey |d(q+1,x) - d(q,x)| wy(q,x),   wy = exp(-|mean_c (I(q+1,x) - I(q,x))|)
// The gradient at (q,x) gathers the signed weights of the two terms it takes part in on each axis:
//   c_ex [tx(q,x-1) - tx(q,x)] + c_ey [sy(q-1,x) - sy(q,x)],   tx = sign(d_dx) wx,  sy = sign(d_dy) wy.
// sy(q-1) is carried from the previous row (it is not recomputed from a third image row), and the rows q .. q+3 of the
// column (disparity + three channels each) live in four register sets used in rotation, as in smooth2_pass.
struct EdgeRow {
  float d, i[3];
};

template <bool GRAD, bool LOSS, bool HWC>
__device__ __forceinline__ void smooth_edge_pass(const LossArgs& A, const ScaleArgs& S, const float* __restrict__ dplane,
                                                 const float* __restrict__ tplane, int lane, int x, bool xin, bool outl, int y0,
                                                 int y1, float* gacc, float& acc_sm, const bool add) {
  const int h = S.h, w = S.w;
  const size_t P = (size_t)h * w;
  const bool vx1 = xin && (x <= w - 2);
  const float outf = outl ? 1.f : 0.f;
  const unsigned xc = (unsigned)min(max(x, 0), w - 1);
  // always a load, from a row clamped into the image (see smooth2_pass); one 12-byte load for the three channels of an HWC pixel
  auto ldrow = [&](int r, EdgeRow& o) {
    const unsigned off = (unsigned)min(max(r, 0), h - 1) * (unsigned)w + xc;
    o.d = ldf(dplane, off);
    if constexpr (HWC) {
      const Rgb t = ld_off<Rgb>(tplane, 12u * off);
#pragma unroll
      for (int c = 0; c < 3; ++c) o.i[c] = t.c[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) o.i[c] = ldf(tplane + c * P, off);
    }
  };
  // No masking of the loaded rows: every load comes from a column clamped into the image, so the values of a halo lane outside
  // the image are finite copies of the border column, and every term such a lane takes part in carries a zero weight (vx1f, xinf).
  const float vx1f = vx1 ? 1.f : 0.f, xinf = xin ? 1.f : 0.f;
  const float KE = -1.44269504088896341f / 3.0f;          // exp(-|s / 3|) = exp2(|s| KE): the mean over the channels folded in
  float gcx = A.gy * S.c_ex, gcy = A.gy * S.c_ey, cex = S.c_ex, cey = S.c_ey;
  asm volatile("" : "+v"(gcx), "+v"(gcy), "+v"(cex), "+v"(cey));   // loop-invariant, pinned in vector registers (as in smooth2_pass)
  float sy_prev = 0.f;   // sy of the row above
  // one row of the walk: c0 = row q, c1 = row q+1; afterwards the registers of row q receive row q+4.  Branch-free up to the
  // (wave-uniform) test whether the row belongs to the chunk: a lane-variant branch would split the block around the DPP reads.
  auto row = [&](const int q, EdgeRow& c0, EdgeRow& c1) {
    const float sx = ((from_right(c0.i[0]) - c0.i[0]) + (from_right(c0.i[1]) - c0.i[1])) + (from_right(c0.i[2]) - c0.i[2]);
    const float sy3 = ((c1.i[0] - c0.i[0]) + (c1.i[1] - c0.i[1])) + (c1.i[2] - c0.i[2]);
    const float ddx = from_right(c0.d) - c0.d;   // d_dx(q,x)
    const float ddy = c1.d - c0.d;               // d_dy(q,x)
    ldrow(q + 4, c0);                            // the registers of row q are free now
    const float rowf = ((unsigned)q <= (unsigned)(h - 2)) ? xinf : 0.f;   // 0 <= q <= h-2 (uniform) and the column inside the image
    const float wx = __builtin_amdgcn_exp2f(fabsf(sx) * KE) * vx1f;
    const float wy = __builtin_amdgcn_exp2f(fabsf(sy3) * KE) * rowf;
    const float sy = ksign(wy, ddy);
    // (tx and its neighbour difference are formed for every row, the one above the chunk included: a cross-lane read is not
    // moved into the branch below, and left outside on its own it costs a register copy per operand instead of riding on the subtraction)
    const float tx = ksign(wx, ddx);
    const float gx = from_left(tx) - tx;
    if (q >= y0) {
      if (LOSS) acc_sm = fmaf(outf, fmaf(cex * fabsf(ddx), wx, cey * fabsf(ddy) * wy), acc_sm);
      if (GRAD) tile_put(gacc + (q - y0) * 64 + lane, fmaf(gcy, sy_prev - sy, gcx * gx), add);
    }
    sy_prev = sy;
  };
  EdgeRow r0, r1, r2, r3;
  ldrow(y0 - 1, r0); ldrow(y0, r1); ldrow(y0 + 1, r2); ldrow(y0 + 2, r3);
  for (int q = y0 - 1; q < y1; q += 4) {   // the first row only produces sy(y0-1)
    row(q, r0, r1);
    if (q + 1 < y1) row(q + 1, r1, r2);
    if (q + 2 < y1) row(q + 2, r2, r3);
    if (q + 3 < y1) row(q + 3, r3, r0);
  }
}

// ------------------------------------------------------------------------------------------
// the main kernel: one wavefront per (scale, sample, strip, chunk)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void set_issue_prio(const int p) {   // s_setprio takes an immediate
  if (p <= 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

// WIDE: the L1 gradient kernels compiled a second time for three waves per SIMD (168 registers instead of 128).  A small launch
// (BASELINE cfg1 / cfg2: fewer waves than three rounds of SIMDs even at the smallest chunk height) is bound by how fast ONE wave
// gets through its instructions, not by how many waves a SIMD holds: cfg2's kernel 14.3 -> 13.4 us.  At B = 32 the same build is
// 8.7 % slower than the four-wave one (profiles/r03_ab_small_kernels.txt), so the plan picks per launch (Plan::wide).
// WARPED: the instantiations that also write the warped source images (SfmLossDesc.warped; LOSS kernels only).
// REF: the projection of the source passes in the reference's own evaluation order (sfm_ssim_pass.h, issue_row) -- an experiment of
// round 5 behind sfm_loss_variant, built for the pixel-interleaved SSIM kernels of sfm_loss_fwd_bwd only.
// DSRC: the instantiations a launch with SfmLossDesc.d_src bound runs (the LDS accumulation window of sfm_ssim_pass.h, dsrc_scatter).
// What a wave needs before anything else -- which item it is -- arrives PRELOADED in scalar registers (the dispatcher delivers the
// first dwords of the argument block with the wave: -mllvm -amdgpu-kernarg-preload-count, see finalize_kernel): the main kernels
// take these ten dwords in front of the by-value struct, so that the first scalar round trip of a wave is already the one for its
// scale's entry (rounds 3-5: header, then entry: two dependent round trips with the whole chip waiting at the start of a launch).
//   h_bn: B | n_src << 16 | n_scales << 20 | prio_top << 24 | COMPACT << 31;  h_t01 .. h_t67: tiles_of[] as 16-bit halves.
// COMPACT = 0 (a tile count or B beyond 16 bits): the header is read from the struct, as before.
#define SFM_HDR_PARAMS const unsigned h_bn, const int h_items, const int h_simds, const unsigned h_prio, const unsigned h_t01, const unsigned h_t23, \
                       const unsigned h_t45, const unsigned h_t67, unsigned long long* const h_trace
#define SFM_HDR_ARGS h_bn, h_items, h_simds, h_prio, h_t01, h_t23, h_t45, h_t67, h_trace
struct Hdr {
  int B, n_src, n_scales, items, simds_per_xcd, prio_top;
  unsigned prio_tab;
  int tiles_of[SFM_MAX_SCALES], item_begin_of[SFM_MAX_SCALES];
  unsigned long long* trace;
};
__device__ __forceinline__ Hdr make_hdr(const LossArgs& A, SFM_HDR_PARAMS) {
  Hdr H;
  if (h_bn >> 31) {      // (uniform)
    H.B = (int)(h_bn & 0xffffu); H.n_src = (int)((h_bn >> 16) & 0xfu); H.n_scales = (int)((h_bn >> 20) & 0xfu); H.prio_top = (int)((h_bn >> 24) & 0x3u);
    H.items = h_items; H.simds_per_xcd = h_simds; H.prio_tab = h_prio; H.trace = h_trace;
    const unsigned tw[4] = {h_t01, h_t23, h_t45, h_t67};
    int run = 0;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) {
      H.tiles_of[k] = (int)((tw[k >> 1] >> (16 * (k & 1))) & 0xffffu);
      H.item_begin_of[k] = run;          // make_plan: item_begin of a scale = B x the tiles of the scales before it
      run += H.B * H.tiles_of[k];
    }
  } else {
    // The header of the argument block is fetched in ONE batch of scalar loads, before anything branches on it: left to where each
    // field is first used, the loads end up behind one another's branches -- eight dependent round trips to a cold scalar cache at the
    // start of every wave, with the whole chip waiting.
    asm volatile("" ::"s"(A.B), "s"(A.n_src), "s"(A.n_scales), "s"(A.items), "s"(A.simds_per_xcd), "s"(A.prio_top), "s"(A.prio_tab),
                 "s"(A.tiles_of[0]), "s"(A.tiles_of[1]), "s"(A.tiles_of[2]), "s"(A.tiles_of[3]), "s"(A.tiles_of[4]), "s"(A.tiles_of[5]),
                 "s"(A.tiles_of[6]), "s"(A.tiles_of[7]), "s"(A.trace));
    H.B = A.B; H.n_src = A.n_src; H.n_scales = A.n_scales; H.items = A.items; H.simds_per_xcd = A.simds_per_xcd; H.prio_top = A.prio_top;
    H.prio_tab = A.prio_tab; H.trace = A.trace;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) { H.tiles_of[k] = A.tiles_of[k]; H.item_begin_of[k] = A.item_begin_of[k]; }
  }
  return H;
}

template <bool SSIM, bool GRAD, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED, int REF = 0, bool DSRC = false>
__device__ __forceinline__ void loss_body(const Hdr& H, const LossArgs& A) {
  static_assert(GRAD || !DSRC, "dL/d(src) is an output of the backward");
  static_assert(LOSS || !WARPED, "the warped images are an output of the forward and the fused entry points");
  static_assert(REF == 0 || (SSIM && GRAD && LOSS && HWC && !EXPL), "the reference-order variants exist for the benchmarked kernels only");
  using HH = Halo<SSIM, GRAD, SMODE>;
  __shared__ float gacc_all[GRAD ? WAVES_PER_BLOCK * MAX_CHUNK_ROWS * 64 : 64];
  const int wave = threadIdx.x >> 6;
  float* gacc = gacc_all + (GRAD ? wave * MAX_CHUNK_ROWS * 64 : 0);

  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  XCD x is given the samples
  // b = x (mod 8) at ALL scales, largest scale first: every XCD gets the same mix of work, and all planes of
  // a sample (target, sources, disparity; shared halo rows, overlapping gather footprints) meet in one L2.
  // With fewer than 8 samples the items are dealt out as 8 contiguous ranges instead.  Placement only affects
  // speed, never the result (the partial sums are indexed by the item id, not by the block).
  static_assert(WAVES_PER_BLOCK == 1, "item mapping assumes one wavefront per workgroup");
  // (the header H: preloaded kernel arguments, see make_hdr)
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  int s = 0, b, t;
  if ((int)H.prio_tab >= 0) {      // (bit 31 of prio_tab: deal the ITEMS out over the XCDs instead, see make_plan)
    // whole groups of eight samples: one sample of each group per XCD; the samples left over (B not a multiple of 8) are dealt out
    // item by item, round-robin over the XCDs (before round 3 they went to the first XCDs whole: B = 11 ran at 11/16)
    const int nb = H.B >> 3;                      // samples of the whole groups owned by this XCD
    int rem = loc;
    bool found = false;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) {
      if (k < H.n_scales && !found) {
        const int cnt = nb * H.tiles_of[k];
        if (rem < cnt) { s = k; found = true; }
        else rem -= cnt;
      }
    }
    if (found) {
      int tls = H.tiles_of[0];
#pragma unroll
      for (int k = 1; k < SFM_MAX_SCALES; ++k) tls = (s == k) ? H.tiles_of[k] : tls;
      const int bl = rem / tls;
      t = rem - bl * tls;
      b = xcd + 8 * bl;
    } else {
      int T = 0;
#pragma unroll
      for (int k = 0; k < SFM_MAX_SCALES; ++k) T += H.tiles_of[k];     // (0 beyond n_scales)
      const int q = rem * 8 + xcd;                // index among the left-over samples' items
      const int left = H.B & 7;
      if (q >= left * T) return;   // whole wavefront leaves; no block-level synchronisation anywhere in this kernel
      const int br = q / T;
      int wi = q - br * T;
      b = (H.B & ~7) + br;
      bool hit = false;
#pragma unroll
      for (int k = 0; k < SFM_MAX_SCALES; ++k) {
        if (!hit) {
          if (wi < H.tiles_of[k]) { s = k; hit = true; }
          else wi -= H.tiles_of[k];
        }
      }
      t = wi;
    }
  } else {
    const int per = (int)(gridDim.x >> 3);
    const int it = xcd * per + loc;
    if (it >= H.items) return;
#pragma unroll
    for (int k = 1; k < SFM_MAX_SCALES; ++k)
      if (k < H.n_scales && it >= H.item_begin_of[k]) s = k;
    int tls = H.tiles_of[0], ibs = H.item_begin_of[0];
#pragma unroll
    for (int k = 1; k < SFM_MAX_SCALES; ++k) { tls = (s == k) ? H.tiles_of[k] : tls; ibs = (s == k) ? H.item_begin_of[k] : ibs; }
    const int idx = it - ibs;
    b = idx / tls;
    t = idx - b * tls;
  }
  const ScaleArgs& S = A.sc[s];
  // ... and the scale's entry in a second one
  asm volatile("" ::"s"(S.tgt), "s"(S.src), "s"(S.disp), "s"(S.d_disp), "s"(S.h), "s"(S.w), "s"(S.strips), "s"(S.tiles), "s"(S.item_begin),
               "s"(S.chunk_rows), "s"(S.inv_cnt), "s"(S.k_pix), "s"(S.kq));
  const int item = S.item_begin + b * S.tiles + t;
  unsigned long long t_start = 0;
  if (H.trace) t_start = __builtin_amdgcn_s_memrealtime();
  // Issue priority.  The SIMD arbiter prefers the oldest wave, so of the co-resident waves of a SIMD one runs ahead
  // and the SIMD ends its launch with a lone wave at half its throughput (profiles/r01_wave_stage_stamps.txt).  The
  // dispatcher places workgroups j, j + S, j + 2S ... of an XCD (S = its SIMD count) on the same SIMD in that age
  // order, so the dispatch round is the age rank: the youngest is preferred during the first half of the sources,
  // the oldest during the second, and the waves of a SIMD finish closer together.  Only ever affects speed.
  const int prio_rank = min((int)(blockIdx.x >> 3) / H.simds_per_xcd, H.prio_top);
  set_issue_prio((int)((H.prio_tab >> (2 * prio_rank)) & 3u));
  const int chunk = t / S.strips;
  const int strip = t - chunk * S.strips;
  const int h = S.h, w = S.w;
  const int lane = threadIdx.x & 63;
  const int x = strip * HH::SW - HH::HL + lane;
  const bool xin = (x >= 0) && (x < w);
  const bool outl = (lane >= HH::HL) && (lane < 64 - HH::HR) && (x < w);
  const int y0 = chunk * S.chunk_rows;
  const int y1 = min(y0 + S.chunk_rows, h);
  const ScaleConst sc = make_scale_const(h, w);
  const size_t P = (size_t)h * w;

  float acc_pix = 0.f, acc_ssim = 0.f, acc_sm = 0.f, acc_exp = 0.f;
  bool first = true;
  // the LDS window of the optional dL/d(src) (dynamic LDS: allocated by the launch only when the descriptor binds d_src)
  extern __shared__ float dsrc_tile[];
  if (DSRC && S.d_src) {
#pragma unroll
    for (int k = 0; k < dsrc_tile_floats(SSIM ? 8 : 4) / 64; ++k) dsrc_tile[k * 64 + lane] = 0.f;
  }
  // the disparities every source pass of this wave starts from (see ssim_source_pass / l1_source_pass): loaded once, now
  float disp_first, disp_second;
  {
    const float* dpl = S.disp + (size_t)b * P;
    const int rfirst = y0 - HH::HS;
    const unsigned xcl = (unsigned)min(max(x, 0), w - 1);
    disp_first = ldf(dpl, (unsigned)min(max(rfirst, 0), h - 1) * (unsigned)w + xcl);
    disp_second = ldf(dpl, (unsigned)min(max(rfirst + 1, 0), h - 1) * (unsigned)w + xcl);
  }
  // the geometry of every source pass of this wave (sfm_ssim_pass.h, build_wave_geom), its loads in the same batch as the disparities
  static_assert(SFM_MAX_SRC * 8 <= 64, "one group of eight lanes per source");
  const WaveGeom WG = build_wave_geom(A.pose, H.n_src, b, A.intrinsics + (size_t)(b * H.n_scales + s) * 9, lane);
#ifdef SFM_STAMPS
  Stamps st = {0, 0, 0, 0, 0};
  unsigned long long ts0 = 0, cyc_smooth = 0, cyc_src = 0;
  SFM_STAMP(ts0);      // wave start, in shader cycles (the trace's t_start / t_end are 100 MHz ticks)
#endif
  // Phases of a wave: the smoothness pass and one pass per source.  The smoothness pass is short on arithmetic and long
  // on latency, and the co-resident waves of a SIMD start together: the middle one (by age) runs it LAST, so that it
  // does not coincide with the others'.  (One call site per kind of pass: the phase loop costs no code.)
  // (Measured round 3 with a run-time table, commit 40da7e8: every wave first +2 %, oldest first / middle between the sources /
  // youngest last +4...7 %, the other mixed orders within noise of this one: profiles/r03_smooth_position_sweep.txt.)
  const bool smooth_last = (SMODE != 0) && (prio_rank == 1);
  const int n_phases = H.n_src + (SMODE != 0 ? 1 : 0);
  for (int ph = 0; ph < n_phases; ++ph) {
    const int i = (SMODE != 0 && !smooth_last) ? ph - 1 : ph;   // source of this phase; -1 or n_src = the smoothness pass
#ifdef SFM_STAMPS
    unsigned long long tp0 = 0, tp1 = 0;
    SFM_STAMP(tp0);
#endif
    if (SMODE != 0 && (i < 0 || i >= H.n_src)) {
      if (SMODE == 1) smooth2_pass<GRAD, LOSS>(A, S, S.disp + (size_t)b * P, lane, x, xin, outl, y0, y1, gacc, acc_sm, !first);
      else smooth_edge_pass<GRAD, LOSS, HWC>(A, S, S.disp + (size_t)b * P, S.tgt + (size_t)b * 3 * P, lane, x, xin, outl, y0, y1, gacc, acc_sm, !first);
      first = false;
#ifdef SFM_STAMPS
      SFM_STAMP(tp1);
      cyc_smooth += tp1 - tp0;
#endif
      continue;
    }
    if (i * 2 >= H.n_src) set_issue_prio((int)((H.prio_tab >> (8 + 2 * prio_rank)) & 3u));
    SsimCtx C;
    const float xf = (float)x;
    C.x0 = x - lane;
    if constexpr (REF != 0) {
      // the geometry of this pass from make_geom: the reference's products and divisions, nothing fused (every lane builds it:
      // ~560 instructions per pass -- this is the variant that is measured, not the product)
      const float* pp = nullptr;
#pragma unroll
      for (int k = 0; k < SFM_MAX_SRC; ++k) pp = (i == k) ? A.pose[k] : pp;
      Geom g;
      make_geom(pp + b * 6, A.intrinsics + (size_t)(b * H.n_scales + s) * 9, g);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        C.M1[k] = uniform(g.M[k * 3 + 1]);
        C.P3[k] = uniform(g.P[k * 4 + 3]);
        C.mx[k] = uniform(g.M[k * 3 + 0]) * xf + uniform(g.M[k * 3 + 2]);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) C.Kinv[k] = uniform(g.Kinv[k]);
#pragma unroll
      for (int k = 0; k < 12; ++k) C.Pm[k] = uniform(g.P[k]);
    } else {
      // the twelve numbers of this pass out of the wave's geometry rows (lane 8 i + k: row k of source i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        C.M1[k] = from_lane(WG.M1, 8 * i + k);
        C.P3[k] = from_lane(WG.P3, 8 * i + k);
        C.mx[k] = fmaf(from_lane(WG.M0, 8 * i + k), xf, from_lane(WG.M2, 8 * i + k));
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      C.tp[k] = S.tgt + ((size_t)b * 3 + k) * P;
      C.sp[k] = S.src + (((size_t)b * H.n_src + i) * 3 + k) * P;
    }
    C.k_pix = S.k_pix;
    C.kq = S.kq;
    C.k_exp = S.k_exp;
    // (the chunk's rows pass through an opaque register once per phase: what depends on them is recomputed per pass instead of
    // being hoisted out of the phase loop and carried -- spilled -- across every pass)
    int y0p = y0, y1p = y1;
    asm volatile("" : "+s"(y0p), "+s"(y1p));
    C.h = h; C.w = w; C.y0 = y0p; C.y1 = y1p;
    C.dp = S.disp + (size_t)b * P;
    C.dsp = (DSRC && S.d_src) ? S.d_src + ((size_t)b * H.n_src + i) * 3 * P : nullptr;
    C.dtile = dsrc_tile;
    C.wp = WARPED ? S.warped + ((size_t)b * H.n_src + i) * 3 * P : nullptr;
    C.mp = EXPL ? S.mlog + ((size_t)b * H.n_src + i) * P : nullptr;
    C.dmp = (EXPL && GRAD) ? S.d_mask + ((size_t)b * H.n_src + i) * P : nullptr;
    C.P = P;
    C.sc = sc;
    C.xc = (unsigned)min(max(x, 0), w - 1);
    C.disp_first = disp_first;
    C.disp_second = disp_second;
    C.xc12 = 12u * C.xc;
    C.x12 = 12u * (unsigned)x;
    C.img12 = 12u * (unsigned)h * (unsigned)w;
    C.w12 = 12u * (unsigned)w;
    C.w12f = (float)(12 * w);
    C.xin = xin;
    C.outb = outl;
    C.xinf = xin ? 1.f : 0.f;
    C.outf = outl ? 1.f : 0.f;
    C.lane = lane;
    float* gpm_out = GRAD ? A.part_gpm + ((size_t)item * H.n_src + i) * 12 : nullptr;
    if constexpr (SSIM) {
      ssim_source_pass<GRAD, LOSS, HWC, WARPED, REF, DSRC>(C, gacc, first, acc_pix, acc_ssim, gpm_out SFM_STAMPS_PASS);
    } else {
      l1_source_pass<GRAD, LOSS, EXPL, HWC, WARPED, DSRC>(C, gacc, first, acc_pix, acc_exp, gpm_out);
    }
    first = false;
#ifdef SFM_STAMPS
    SFM_STAMP(tp1);
    cyc_src += tp1 - tp0;
#endif
  }
  // the wave's four loss sums, reduced in LOCKSTEP (same adds in the same order per value as four wave_sum calls one after the other:
  // bit-identical) and in front of the d_disp write-out, whose LDS reads and stores issue under the DPP chain's latency -- the last wave
  // of a launch is alone on its SIMD when it gets here
  if (LOSS) {
    float v[4] = {acc_pix, acc_ssim, acc_sm, acc_exp};
    wave_sums_lockstep(v);
    if (lane == 63) {
      float* o = A.part_loss + (size_t)item * 4;
      o[0] = v[0] * S.inv_cnt; o[1] = v[1] * S.inv_cnt; o[2] = v[2]; o[3] = v[3] * S.c_exp;
    }
  }
  if (GRAD) {
    if (outl) {
      float* o = S.d_disp + (size_t)b * P;
      for (int q = y0; q < y1; ++q) stf_wt(o, (unsigned)(q * w + x), gacc[(q - y0) * 64 + lane]);
    }
  }
  if (H.trace && lane == 0) {   // timing-only diagnostics; never read by any kernel
    unsigned long long* o = H.trace + (size_t)item * 4;
    o[0] = t_start;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    o[3] = (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf) | ((unsigned long long)blockIdx.x << 8);   // HW_REG_XCC_ID, workgroup
#ifdef SFM_STAMPS
    {
      unsigned long long* q = H.trace + (size_t)H.items * 4 + (size_t)item * 8;
      unsigned long long ts3 = 0;
      SFM_STAMP(ts3);
      q[0] = st.a_fin; q[1] = st.a_iss; q[2] = st.b; q[3] = st.c; q[4] = st.steps > 0 ? st.steps : 1;   // (the L1 passes carry no per-stage stamps)
      q[5] = cyc_smooth;   // the smoothness pass
      q[6] = cyc_src;      // the source passes, everything included (context set-up, prologue, row loop, pose sums)
      q[7] = ts3 - ts0;    // the whole wave in shader cycles (start-up and d_disp write-out = the rest)
    }
#endif
  }
}

template <bool SSIM, bool GRAD, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, (SSIM && GRAD) ? 3 : 4) loss_kernel(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<SSIM, GRAD, LOSS, EXPL, SMODE, HWC, WARPED>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (DSRC, see loss_body: the gradient kernels of a launch that also wants dL/d(src).  Three waves per SIMD; the SSIM ones two: at
//  three they would spill 22-34 registers to scratch)
template <bool SSIM, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, SSIM ? 2 : 3) loss_kernel_dsrc(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<SSIM, true, LOSS, EXPL, SMODE, HWC, WARPED, 0, true>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (REF, see loss_body: the fused SSIM kernels in the pixel-interleaved layout, in the reference's evaluation order)
template <int SMODE, bool WARPED, int REF>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 3) loss_kernel_ref(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<true, true, true, false, SMODE, true, WARPED, REF>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (WIDE, see above: L1 gradient kernels only)
template <bool LOSS, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 3) loss_kernel_wide(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<false, true, LOSS, false, SMODE, HWC, WARPED>(make_hdr(A, SFM_HDR_ARGS), A);
}

// ------------------------------------------------------------------------------------------
// finalize: fixed-order reduction of the per-wave partials
//   blocks [0, B*n_src): d_pose of (b, i)         (only when do_pose)
//   last block        : the five reported scalars (only when loss5 != nullptr)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#ifdef SFM_FIN_STAMPS   // diagnostic build only: 100 MHz time stamps of finalize_kernel's stages into the debug trace buffer (tools/trace_finalize.py)
#define SFM_FSTAMP(slot) do { if (A.trace && threadIdx.x == 0) A.trace[200000 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SFM_FSTAMP_F(slot) do { if (A.trace && threadIdx.x == 64) A.trace[200000 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // first FOLDING wave
#else
#define SFM_FSTAMP(slot) do { } while (0)
#define SFM_FSTAMP_F(slot) do { } while (0)
#endif

constexpr int FINALIZE_WAVES = 16;   // waves of the block that sums the loss partials (the pose blocks use one)

// The first fourteen dwords of the argument block are PRELOADED into scalar registers by the dispatcher (-mllvm
// -amdgpu-kernarg-preload-count=14 in the Makefile; every other kernel of this file takes one by-value struct, which is never preloaded):
// what a pose block needs to form the addresses of its first loads is there when the wave starts, instead of one scalar round trip
// later (tools/anyorder_probe.hip: 0.15 us per kernel on this stack; stamps of this kernel: 0.5 us from its start to "tiles known").
//   q_bsc: B | n_src << 16 | n_scales << 20 | COMPACT << 24 | POSE << 25 | LOSS << 26;  q_t01 .. q_t67: tiles_of[] as 16-bit halves;
//   q_loss_back: bytes from part_loss up to part_gpm (both lie in the caller's workspace: the loss block's pointer without a round trip).
// COMPACT = 0 (a tile count or B beyond 16 bits: planar frames of tens of megapixels): the fields are read from the struct, as before.
constexpr unsigned FIN_COMPACT = 1u << 24, FIN_POSE = 1u << 25, FIN_LOSS = 1u << 26;
__global__ void __launch_bounds__(64 * FINALIZE_WAVES) finalize_kernel(const float* __restrict__ q_gpm, const float* __restrict__ q_intr,
                                                                       const float* __restrict__ q_pose0, const float* __restrict__ q_pose1,
                                                                       const unsigned q_bsc, const unsigned q_t01, const unsigned q_t23,
                                                                       const unsigned q_t45, const unsigned q_t67, const unsigned q_loss_back,
                                                                       const LossArgs A, float* __restrict__ loss5) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  static_assert(SFM_MAX_SCALES == 8, "four packed words of tile counts");
  const bool compact = (q_bsc & FIN_COMPACT) != 0;
  int hB, h_src, h_scales, h_items, tlh[SFM_MAX_SCALES], ibh[SFM_MAX_SCALES];
  const float *h_gpm, *h_intr, *h_loss;
  if (compact) {          // (uniform)
    hB = (int)(q_bsc & 0xffffu); h_src = (int)((q_bsc >> 16) & 0xfu); h_scales = (int)((q_bsc >> 20) & 0xfu);
    const unsigned tw[4] = {q_t01, q_t23, q_t45, q_t67};
    int run = 0;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) {
      tlh[k] = (int)((tw[k >> 1] >> (16 * (k & 1))) & 0xffffu);
      ibh[k] = run;                              // make_plan: item_begin of a scale = B x the tiles of the scales before it
      run += hB * tlh[k];
    }
    h_items = run;
    h_gpm = q_gpm; h_intr = q_intr;
    h_loss = reinterpret_cast<const float*>(reinterpret_cast<const char*>(q_gpm) - q_loss_back);
  } else {
    hB = A.B; h_src = A.n_src; h_scales = A.n_scales;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) { tlh[k] = A.tiles_of[k]; ibh[k] = A.item_begin_of[k]; }
    h_gpm = A.part_gpm; h_intr = A.intrinsics; h_loss = A.part_loss; h_items = A.items;
  }
  const int n_pose_blocks = (q_bsc & FIN_POSE) ? hB * h_src : 0;
  if ((int)blockIdx.x < n_pose_blocks) {
    // d_pose of (b, i): every lane folds its tiles of every scale into K_s^T . gPm (linear), one in-register wave reduction (DPP).
    // A sample with many tiles (308 at cfg2, 376 at cfg5) is spread over the sixteen waves of the block, so that its partials are
    // fetched in ONE round of independent loads instead of up to six dependent rounds (-1 us on those steps); the waves' sums meet
    // in LDS and are added in wave order: a fixed order, the result does not depend on timing.
    const int b = blockIdx.x / h_src, i = blockIdx.x - b * h_src;
    const bool stamp = blockIdx.x == 0;
    if (stamp) SFM_FSTAMP(0);
    __shared__ float pose_red[FINALIZE_WAVES][6];
    // This block is a chain of round trips to memory with little arithmetic between them, so it is written to need ONE: what the
    // addresses of the first loads are formed from arrives preloaded (above), and the pose (helper wave) and the partials (folding
    // waves) go out together; the pose pointer of source i is selected, never loaded through an index.
    int tl[SFM_MAX_SCALES], ib[SFM_MAX_SCALES];
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) { tl[k] = tlh[k]; ib[k] = ibh[k]; }
    const float* pp = nullptr;
    float* dp = nullptr;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SRC; ++k) {
      float* dk = A.d_pose[k];                 // (needed at the very end: its round trip hides behind everything else)
      asm volatile("" : "+s"(dk));             // (keeps the selection from being folded back into an indexed load)
      dp = (i == k) ? dk : dp;
    }
    if (compact && i < 2) {                    // (uniform) the pose pointers of the first two sources are preloaded
      pp = (i == 0) ? q_pose0 : q_pose1;
    } else {
#pragma unroll
      for (int k = 0; k < SFM_MAX_SRC; ++k) {
        const float* pk = A.pose[k];
        asm volatile("" : "+s"(pk));
        pp = (i == k) ? pk : pp;
      }
    }
    int total = 0;
#pragma unroll
    for (int k = 0; k < SFM_MAX_SCALES; ++k) total += tl[k];
    // The block's waves: the FIRST one is the HELPER -- it builds the Jacobian of the Euler chain, dR/d(theta_k) with the clip mask of
    // transform.py:23 folded in, while the others wait for their partials; the next nfold waves FOLD, one tile per lane (cfg3: 96 tiles,
    // two waves; cfg2 / cfg5: 308 / 376, five / six waves; a sample of more than 960 tiles goes round again), the rest leave at once.
    // Every lane contracts its folded sums with the Jacobian -- d_pose[k] = <dL/dR, dR/d(theta_k)>, d_pose[3 + k] = dL/dt_k are linear
    // in them -- so SIX values go through the wave reduction (not twelve) and nothing is left to do behind it; the folding waves' six sums
    // meet in LDS and are added in wave order: a fixed order, the result does not depend on timing.
    // (Rounds 3-5 had two forms: one wave + helper for samples of up to 128 tiles, sixteen waves, twelve sums each and a 150-instruction
    //  pose_backward by one lane at the very end for the others.)
    const int nfold = min((total + 63) / 64, FINALIZE_WAVES - 1);    // block-uniform, >= 1
    const bool helper = wave == 0;         // (the wave that is launched FIRST: its chain -- pose, sincos, Jacobian -- is the longest)
    const int fw = wave - 1;               // folding wave 0 .. nfold-1
    __shared__ float jac[27];
    if (helper) {
      const float pi = 3.14159265358979323846f;
      float ang[3], sn[3], cs[3], msk[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) ang[k] = pp[b * 6 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        sincos_pi(fminf(fmaxf(ang[k], -pi), pi), &sn[k], &cs[k]);
        msk[k] = (ang[k] > -pi && ang[k] < pi) ? 1.f : 0.f;        // F.clip backward
      }
      // R = (X Y) Z (transform.py:27-39); its derivatives in CLOSED FORM -- the products with the zeros and ones of X, Y, Z and of their
      // derivatives written out (30 instructions instead of the 160 of six general 3x3 products: the helper is what the folding waves
      // wait for at the barrier):
      //   X Y  = [[cy, 0, sy], [sx sy, cx, -sx cy], [-cx sy, sx, cx cy]]
      //   X'Y  = [[0, 0, 0], [cx sy, -sx, -cx cy], [sx sy, cx, -sx cy]]          X Y' = [[-sy, 0, cy], [sx cy, 0, sx sy], [-cx cy, 0, -cx sy]]
      //   M Z  = [M0 cz + M1 sz, -M0 sz + M1 cz, M2] (columns)                   M Z' = [-M0 sz + M1 cz, -M0 cz - M1 sz, 0]
      const float sx = sn[0], cx = cs[0], sy = sn[1], cy = cs[1], sz = sn[2], cz = cs[2];
      const float sxsy = sx * sy, sxcy = sx * cy, cxsy = cx * sy, cxcy = cx * cy;
      float J[27];
      // dR/d(theta_x) = (X'Y) Z
      J[0] = 0.f; J[1] = 0.f; J[2] = 0.f;
      J[3] = fmaf(cxsy, cz, -(sx * sz)); J[4] = -fmaf(cxsy, sz, sx * cz); J[5] = -cxcy;
      J[6] = fmaf(sxsy, cz, cx * sz);    J[7] = fmaf(-sxsy, sz, cx * cz); J[8] = -sxcy;
      // dR/d(theta_y) = (X Y') Z      (the middle column of X Y' is zero)
      J[9] = -(sy * cz);   J[10] = sy * sz;      J[11] = cy;
      J[12] = sxcy * cz;   J[13] = -(sxcy * sz); J[14] = sxsy;
      J[15] = -(cxcy * cz); J[16] = cxcy * sz;   J[17] = -cxsy;
      // dR/d(theta_z) = (X Y) Z'
      J[18] = -(cy * sz);                  J[19] = -(cy * cz);                    J[20] = 0.f;
      J[21] = fmaf(-sxsy, sz, cx * cz);    J[22] = -fmaf(sxsy, cz, cx * sz);      J[23] = 0.f;
      J[24] = fmaf(cxsy, sz, sx * cz);     J[25] = fmaf(cxsy, cz, -(sx * sz));    J[26] = 0.f;
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 27; ++k) jac[k] = J[k] * msk[k / 9];
      }
      __syncthreads();
      return;
    }
    if (fw >= nfold) return;
    if (stamp) SFM_FSTAMP_F(1);
    float gT3[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gT3[k] = 0.f;
    // (macros, not lambdas: arrays captured by reference are not split into registers, the compiler parks them in LDS and fetches
    //  the workgroup size for that from the dispatch packet -- in host memory, 10 us away)
#define SFM_FIN_FETCH(idx_, v0_, v1_, v2_, K_)                                                                                    \
  {                                                                                                                               \
    int s_ = 0, off_ = 0;                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < SFM_MAX_SCALES - 1; ++k)                                                                \
        if ((idx_) >= off_ + tl[k] && s_ == k) { off_ += tl[k]; s_ = k + 1; }                                                     \
    int ibs_ = ib[0], tls_ = tl[0];                                                                                               \
    _Pragma("unroll") for (int k = 1; k < SFM_MAX_SCALES; ++k) { ibs_ = (s_ == k) ? ib[k] : ibs_; tls_ = (s_ == k) ? tl[k] : tls_; } \
    const float4* p_ = reinterpret_cast<const float4*>(h_gpm + ((size_t)(ibs_ + b * tls_ + ((idx_) - off_)) * h_src + i) * 12);  \
    v0_ = p_[0]; v1_ = p_[1]; v2_ = p_[2];                                                                                        \
    const float* Kp_ = h_intr + ((size_t)b * h_scales + s_) * 9;                                                                  \
    _Pragma("unroll") for (int k = 0; k < 9; ++k) K_[k] = Kp_[k];                                                                 \
  }
    // the tile's raw pose sums S_k = (sum x A_k, B_k, A_k, C_k) (pose_sums_raw) -> dL/dPm[k][j] = Kinv[j] . S_k[0:3], [k][3] = S_k[3];
    // then gT3 += K^T . gPm   (K4^T . gPm of the rows that reach R and t)
#define SFM_FIN_FOLD(v0_, v1_, v2_, K_)                                                                                           \
  {                                                                                                                               \
    const float s_[12] = {v0_.x, v0_.y, v0_.z, v0_.w, v1_.x, v1_.y, v1_.z, v1_.w, v2_.x, v2_.y, v2_.z, v2_.w};                    \
    float Ki_[9], g_[12];                                                                                                         \
    inv3_fast(K_, Ki_);                                                                                                           \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                                               \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                               \
          g_[k * 4 + j] = fmaf(Ki_[j * 3 + 2], s_[k * 4 + 2], fmaf(Ki_[j * 3 + 1], s_[k * 4 + 1], Ki_[j * 3 + 0] * s_[k * 4 + 0])); \
      g_[k * 4 + 3] = s_[k * 4 + 3];                                                                                              \
    }                                                                                                                             \
    _Pragma("unroll") for (int r = 0; r < 3; ++r)                                                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                             \
            gT3[r * 4 + c] += K_[0 * 3 + r] * g_[0 * 4 + c] + K_[1 * 3 + r] * g_[1 * 4 + c] + K_[2 * 3 + r] * g_[2 * 4 + c];      \
  }
    const int stride = 64 * nfold;
    const int idx0 = fw * 64 + lane;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    float Ka[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (idx0 < total) SFM_FIN_FETCH(idx0, a0, a1, a2, Ka)
    if (stamp) SFM_FSTAMP_F(2);
    if (idx0 < total) SFM_FIN_FOLD(a0, a1, a2, Ka)
    for (int idx = idx0 + stride; idx < total; idx += stride) {   // (samples of more than 960 tiles)
      SFM_FIN_FETCH(idx, a0, a1, a2, Ka)
      SFM_FIN_FOLD(a0, a1, a2, Ka)
    }
#undef SFM_FIN_FETCH
#undef SFM_FIN_FOLD
    __syncthreads();                       // the helper's Jacobian (the helper and the folding waves)
    float d[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) a = fmaf(gT3[r * 4 + c], jac[k * 9 + r * 3 + c], a);
      d[k] = a;
      d[3 + k] = gT3[k * 4 + 3];
    }
#ifdef SFM_FIN_STAMPS
    if (stamp && d[0] != 77.f) SFM_FSTAMP_F(3);
#endif
    wave_sums_lockstep(d);
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = lane63(d[k]);
#ifdef SFM_FIN_STAMPS
    if (stamp && d[0] != 77.f) SFM_FSTAMP_F(4);
#endif
    if (nfold > 1) {                       // (block-uniform)
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pose_red[fw][k] = d[k];
      }
      __syncthreads();
      if (fw != 0) return;
      if (lane < 6) {                      // lane k adds the folding waves' sums of component k in wave order and stores it
        float a = pose_red[0][lane];
        for (int wv = 1; wv < nfold; ++wv) a += pose_red[wv][lane];
        dp[b * 6 + lane] = a;
      }
    } else if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) dp[b * 6 + k] = d[k];
    }
#ifdef SFM_FIN_STAMPS
    if (stamp && d[0] != 77.f) SFM_FSTAMP_F(5);
#endif
    return;
  }
  if (!(q_bsc & FIN_LOSS)) return;
  // the five reported scalars.  Every lane sums its items in fp64 (sixteen waves: one round of independent loads at cfg3); the 1024
  // lane sums of each scalar meet in LDS, and FOUR waves -- one per scalar, so one per SIMD -- add their scalar's sixteen wave
  // columns lane by lane in wave order and reduce the 64 lane sums in registers ((hi, lo) float pair, DPP): a fixed order, the
  // result does not depend on timing.  (Until round 5 each of the sixteen waves ran a DPP reduction of all four scalars: 150
  // instructions x four waves per SIMD = 0.9 us between the arrival of the partials and the barrier; now 0.4.)
  SFM_FSTAMP(8);
  __shared__ double lane_acc[4][FINALIZE_WAVES][64];
  __shared__ double scalar_red[4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const float4* pl = reinterpret_cast<const float4*>(h_loss);
#pragma unroll 4
  for (int t = threadIdx.x; t < h_items; t += 64 * FINALIZE_WAVES) {
    const float4 v = pl[t];
    acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
  }
#ifdef SFM_FIN_STAMPS
  if (acc[0] != 77.0) SFM_FSTAMP(9);
#endif
#pragma unroll
  for (int k = 0; k < 4; ++k) lane_acc[k][wave][lane] = acc[k];
  __syncthreads();
  SFM_FSTAMP(10);
  if (wave >= 4) return;
  {
    static_assert(FINALIZE_WAVES % 4 == 0, "four chains of wave columns");
    double m4[4] = {0.0, 0.0, 0.0, 0.0};   // (four chains side by side instead of sixteen additions in a row; combined in a fixed order)
#pragma unroll
    for (int wv = 0; wv < FINALIZE_WAVES; ++wv) m4[wv & 3] += lane_acc[wave][wv][lane];
    const double m = (m4[0] + m4[1]) + (m4[2] + m4[3]);
    float hl[2];
    hl[0] = (float)m;
    hl[1] = (float)(m - (double)hl[0]);
    wave_sums_lockstep(hl);
    const double r = (double)lane63(hl[0]) + (double)lane63(hl[1]);
    if (lane == 0) scalar_red[wave] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double pixel = scalar_red[0], ssim = scalar_red[1], smooth = scalar_red[2], expl = scalar_red[3];
    const double a = (double)A.alpha;
    loss5[0] = (float)((1.0 - a) * pixel + a * ssim + smooth + expl);   // base_model.py:117-118
    loss5[1] = (float)pixel;
    loss5[2] = (float)smooth;
    loss5[3] = (float)expl;
    loss5[4] = (float)ssim;
    SFM_FSTAMP(11);
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Plan {
  LossArgs args;
  size_t off_loss, off_gpm, total;
  bool ssim, expl, hwc;
  bool wide;     // the three-waves-per-SIMD build of an L1 gradient kernel (see loss_kernel)
  bool dsrc;     // the launch also produces dL/d(src): the instantiations with the LDS accumulation window (three waves per SIMD)
  bool warped;   // the instantiation that also writes SfmLossDesc.warped
  int smode;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// optional profiling hook (sfm_loss_profile_events): events recorded right before / after the
// main kernel of the NEXT fused-loss call of this thread
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static thread_local unsigned long long* g_trace = nullptr;   // sfm_loss_debug_trace
static thread_local int g_variant = 0;                       // sfm_loss_variant: holds for the next sfm_loss_* call only

// ---- work decomposition -----------------------------------------------------------------
// One wavefront per (scale, sample, strip, chunk of rows).  A wave lives for the whole launch, so the chunk
// height is chosen such that all items fit in as few full "rounds" of resident waves as possible while the
// halo rows (recomputed per chunk) stay a small fraction.  The number of resident waves follows from the
// __launch_bounds__ of the kernel variant (3 waves per SIMD for the SSIM gradient kernels, 4 otherwise; LDS
// and the 32-waves-per-CU cap allow more) and from the CU count of the device -- no occupancy query, nothing
// that differs between a CPU-only host and the GPU box.
template <bool GRAD, bool LOSS>
static const void* kernel_ptr(bool ssim, bool expl, int smode, bool hwc, bool wide, bool warped);

constexpr int MI355X_CUS = 256;   // 8 XCDs x 32 CUs (MI355X_MICROARCH.md); used when no device is visible

static int device_cus() {
  static int cus_of[64];   // 0 = not asked yet; a benign race writes the same value twice
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return MI355X_CUS; }
  if (cus_of[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = MI355X_CUS; }
    cus_of[dev] = cus;
  }
  return cus_of[dev];
}

static int waves_per_simd_of(bool ssim, bool grad, bool wide, bool dsrc) {   // = __launch_bounds__ of the kernel
  if (dsrc) return ssim ? 2 : 3;
  return ((ssim && grad) || wide) ? 3 : 4;
}

// tuning overrides (development only), read from the environment ONCE per process
struct Tuning {
  int chunk_rows = 0;                       // SFM_CHUNK_ROWS: fixed target height
  int rows_list[SFM_MAX_SCALES] = {0};      // SFM_CHUNK_ROWS_LIST: chunk height per scale, "13,13,16,8"
  bool has_prio = false;
  unsigned prio_tab = 0;                    // SFM_PRIO_TABLE: "0123,3210" = levels of ranks 0.. in phase 1, phase 2
  bool no_wide = false;                     // SFM_NO_WIDE: small L1 launches on the four-wave build too
  bool no_fill = false;                     // SFM_NO_FILL: no slot-filling refinement of the chunk heights (plan_chunks)
  int deal_below = 8;                       // SFM_DEAL_ITEMS_BELOW: batches smaller than this have their ITEMS dealt out over the XCDs (8 contiguous
                                            // ranges of the item list) instead of whole samples (b mod 8).  Measured at B = 8 (9): cfg5_2src +9.9 %,
                                            // cfg5 +3 %, cfg2 +1.6 %; items round-robin +22 .. +48 % (profiles/r05_wave_stage_stamps_cfg5_2src.txt)
  Tuning() {
    if (const char* e = getenv("SFM_DEAL_ITEMS_BELOW")) deal_below = atoi(e);
    no_wide = getenv("SFM_NO_WIDE") != nullptr;
    no_fill = getenv("SFM_NO_FILL") != nullptr;
    if (const char* e = getenv("SFM_CHUNK_ROWS")) chunk_rows = atoi(e);
    if (const char* rl = getenv("SFM_CHUNK_ROWS_LIST")) {
      for (int k = 0; *rl && k < SFM_MAX_SCALES; ++k) {
        rows_list[k] = atoi(rl);
        while (*rl && *rl != ',') ++rl;
        if (*rl == ',') ++rl;
      }
    }
    if (const char* pt = getenv("SFM_PRIO_TABLE")) {
      int phase = 0, r = 0;
      for (; *pt; ++pt) {
        if (*pt == ',') { phase = 1; r = 0; }
        else if (*pt >= '0' && *pt <= '3' && r < 4) { prio_tab |= (unsigned)(*pt - '0') << (8 * phase + 2 * r); ++r; }
      }
      has_prio = true;
    }
  }
};
static const Tuning& tuning() { static const Tuning t; return t; }

// chunk height per scale for a target height T: equal chunks, never more than T rows
static void chunk_layout(const SfmLossDesc* d, int sw, int halo2, int T, int* rows, long long* items, long long* work, int* maxcost) {
  *items = 0; *work = 0; *maxcost = 0;
  for (int s = 0; s < d->n_scales; ++s) {
    const int h = d->H[s], strips = (d->W[s] + sw - 1) / sw;
    const int n = (h + T - 1) / T;
    rows[s] = (h + n - 1) / n;
    const int cost = rows[s] + halo2;
    const long long cnt = (long long)d->B * strips * ((h + rows[s] - 1) / rows[s]);
    *items += cnt;
    *work += cnt * cost;
    if (cost > *maxcost) *maxcost = cost;
  }
}

static void plan_chunks(const SfmLossDesc* d, int sw, int halo2, int slots, int* rows) {
  const int forced = tuning().chunk_rows;
  int bestT = MAX_CHUNK_ROWS;
  double best = 1e300;
  for (int T = MIN_CHUNK_ROWS; T <= MAX_CHUNK_ROWS; ++T) {
    if (forced >= MIN_CHUNK_ROWS && forced <= MAX_CHUNK_ROWS && T != forced) continue;
    long long items, work;
    int maxcost, r[SFM_MAX_SCALES];
    chunk_layout(d, sw, halo2, T, r, &items, &work, &maxcost);
    const double rounds = (double)((items + slots - 1) / slots);
    // resident waves run concurrently: a launch lasts about `rounds` times the longest item, and never
    // less than the total work spread over all slots
    double est = rounds * maxcost;
    const double flat = (double)work / slots;
    if (flat > est) est = flat;
    est += 1e-3 * (double)work / slots;   // tie-break: less recomputed halo
    if (est < best) { best = est; bestT = T; }
  }
  long long items, work;
  int maxcost;
  chunk_layout(d, sw, halo2, bestT, rows, &items, &work, &maxcost);
  // Refinement for launches that fit ONE resident round: the equal-height search above leaves slots empty (cfg3: 2912 items on
  // 3072 slots, i.e. 160 SIMDs with two waves instead of three, which finish at 31-48 us of a 55 us launch).  The empty slots are
  // filled by splitting the chunks of the scale with the tallest chunks that still fits, one more chunk at a time: every extra
  // chunk costs its halo rows again (+1 % of the row steps at cfg3).  Measured (profiles/r04_ab_chunk_fill.txt): the main kernel gets
  // 1.5 - 2.2 % shorter where a wave is SHORT -- B = 16 / 24 / 32 at 128x416 with two sources, cfg5 with two sources -- and 3 - 5 %
  // LONGER where it is long (cfg5 with four sources: 17 steps x 4; B = 48: 26 steps x 2), so the refinement is applied while a
  // wave's row steps, tallest chunk x sources, stay within 40.
  if (!tuning().no_fill && items <= slots && !(forced >= MIN_CHUNK_ROWS && forced <= MAX_CHUNK_ROWS) && (long long)maxcost * d->n_src <= 40) {
    for (;;) {
      int pick = -1, pick_rows = 0;
      long long pick_add = 0;
      for (int s = 0; s < d->n_scales; ++s) {
        const int h = d->H[s], strips = (d->W[s] + sw - 1) / sw;
        const int chunks = (h + rows[s] - 1) / rows[s];
        if (rows[s] <= MIN_CHUNK_ROWS) continue;
        const int nr = (h + chunks) / (chunks + 1);                 // ceil(h / (chunks + 1))
        if (nr < MIN_CHUNK_ROWS || nr >= rows[s]) continue;
        const long long add = (long long)d->B * strips * ((h + nr - 1) / nr - chunks);
        if (add <= 0 || items + add > slots) continue;
        if (rows[s] > pick_rows) { pick = s; pick_rows = rows[s]; pick_add = add; }
      }
      if (pick < 0) break;
      const int h = d->H[pick], chunks = (h + rows[pick] - 1) / rows[pick];
      rows[pick] = (h + chunks) / (chunks + 1);
      items += pick_add;
    }
  }
  for (int k = 0; k < d->n_scales; ++k) {
    const int v = tuning().rows_list[k];
    if (v >= MIN_CHUNK_ROWS && v <= MAX_CHUNK_ROWS) rows[k] = v;
  }
}

// upper bound of the item count over every chunking plan_chunks can choose (chunks are never shorter than
// MIN_CHUNK_ROWS unless the image is): what sfm_loss_workspace_bytes sizes for, on any host
static long long max_items(const SfmLossDesc* d, int sw) {
  long long items = 0;
  for (int s = 0; s < d->n_scales; ++s) {
    const int h = d->H[s], strips = (d->W[s] + sw - 1) / sw;
    items += (long long)d->B * strips * ((h + MIN_CHUNK_ROWS - 1) / MIN_CHUNK_ROWS);
  }
  return items;
}

static void set_gy(struct Plan& p, const float gy);

// validates the descriptor and lays out items + workspace for the given mode
static int make_plan(const SfmLossDesc* d, bool grad, bool need_loss, bool need_outputs, float gy, Plan& p) {
  if (!d) return fail(SFM_ERR_NULL, "sfm_loss: NULL descriptor");
  if (d->B < 0 || d->B > (1 << 20)) return fail(SFM_ERR_SHAPE, "sfm_loss: B=%d", d->B);
  if (d->norm_B < d->B || d->norm_B < 1) return fail(SFM_ERR_CONFIG, "sfm_loss: norm_B=%d must be >= max(B,1) (B=%d)", d->norm_B, d->B);
  if (d->n_src < 1 || d->n_src > SFM_MAX_SRC) return fail(SFM_ERR_SHAPE, "sfm_loss: n_src=%d not in [1,%d]", d->n_src, SFM_MAX_SRC);
  if (d->n_scales < 1 || d->n_scales > SFM_MAX_SCALES)
    return fail(SFM_ERR_SHAPE, "sfm_loss: n_scales=%d not in [1,%d]", d->n_scales, SFM_MAX_SCALES);
  if (!(d->ssim_rate >= 0.f && d->ssim_rate <= 1.f)) return fail(SFM_ERR_CONFIG, "sfm_loss: ssim_rate=%g not in [0,1]", d->ssim_rate);
  if (!(d->smooth_reg >= 0.f) || !(d->exp_reg >= 0.f)) return fail(SFM_ERR_CONFIG, "sfm_loss: negative regulariser weight");
  if (d->smooth_mode < SFM_SMOOTH_NONE || d->smooth_mode > SFM_SMOOTH_EDGE_AWARE)
    return fail(SFM_ERR_CONFIG, "sfm_loss: smooth_mode=%d", d->smooth_mode);
  if (!d->intrinsics) return fail(SFM_ERR_NULL, "sfm_loss: intrinsics is NULL");
  if (d->image_layout != SFM_LAYOUT_PLANAR && d->image_layout != SFM_LAYOUT_HWC)
    return fail(SFM_ERR_CONFIG, "sfm_loss: image_layout=%d", d->image_layout);
  p.hwc = d->image_layout == SFM_LAYOUT_HWC;
  p.expl = d->exp_reg > 0.f;                       // base_model.py:86,103
  p.ssim = !p.expl && d->ssim_rate > 0.f;          // base_model.py:110-112
  p.smode = d->smooth_reg > 0.f ? d->smooth_mode : SFM_SMOOTH_NONE;   // base_model.py:75
  LossArgs& A = p.args;
  memset(&A, 0, sizeof(A));
  A.B = d->B;
  A.n_src = d->n_src;
  A.n_scales = d->n_scales;
  A.gy = gy;
  A.alpha = d->ssim_rate;
  A.intrinsics = d->intrinsics;
  for (int i = 0; i < d->n_src; ++i) {
    if (!d->pose[i]) return fail(SFM_ERR_NULL, "sfm_loss: pose[%d] is NULL", i);
    A.pose[i] = d->pose[i];
    if (grad && need_outputs) {
      if (!d->d_pose[i]) return fail(SFM_ERR_NULL, "sfm_loss: d_pose[%d] is NULL", i);
      A.d_pose[i] = d->d_pose[i];
    }
  }
  // the optional warped-image output: an array for every scale or for none
  p.warped = false;
  if (need_loss && need_outputs) {
    int n_w = 0;
    for (int s = 0; s < d->n_scales; ++s) n_w += d->warped[s] != nullptr;
    if (n_w != 0 && n_w != d->n_scales) return fail(SFM_ERR_NULL, "sfm_loss: warped[] is set for %d of %d scales (all or none)", n_w, d->n_scales);
    p.warped = n_w != 0;
  }
  const int sw = strip_width(p.ssim, grad, p.smode);
  const int hs = p.ssim ? (grad ? 2 : 1) : 0;
  const int hm = p.smode == 1 ? 2 : (p.smode == 2 ? 1 : 0);
  int rows[SFM_MAX_SCALES];
  for (int s = 0; s < d->n_scales; ++s)
    if (d->H[s] < 3 || d->W[s] < 3) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, need H,W >= 3", s, d->H[s], d->W[s]);
  const int cus = device_cus();
  // (a launch is "small" when even at the smallest chunk height its waves fit the SIMDs three deep)
  p.dsrc = false;
  if (grad && need_outputs)
    for (int s = 0; s < d->n_scales; ++s) p.dsrc = p.dsrc || d->d_src[s] != nullptr;
  p.wide = grad && !p.dsrc && !p.ssim && !p.expl && !tuning().no_wide && max_items(d, sw) <= (long long)cus * 4 * 3;
  const int waves_per_simd = waves_per_simd_of(p.ssim, grad, p.wide, p.dsrc);
  const int slots = cus * 4 * waves_per_simd;
  A.simds_per_xcd = (cus % 8 == 0) ? cus / 8 * 4 : 128;   // gfx950: 8 XCDs, 4 SIMDs per CU
  A.prio_top = waves_per_simd - 1 < 3 ? waves_per_simd - 1 : 3;
  A.prio_tab = 0;
  for (int r = 0; r <= A.prio_top; ++r)   // youngest preferred in the first half of the sources, oldest in the second
    A.prio_tab |= (unsigned)r << (2 * r) | (unsigned)(A.prio_top - r) << (8 + 2 * r);
  if (tuning().has_prio) A.prio_tab = tuning().prio_tab & 0xffffu;
  if (d->B < (tuning().deal_below > 8 ? tuning().deal_below : 8)) A.prio_tab |= 0x80000000u;   // fewer samples than XCDs (or asked for): deal items
  plan_chunks(d, sw, 2 * (hs > hm ? hs : hm), slots, rows);
  int items = 0;
  for (int s = 0; s < d->n_scales; ++s) {
    const int h = d->H[s], w = d->W[s];
    if (h < 3 || w < 3) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, need H,W >= 3", s, h, w);
    if ((long long)d->B * 3 * d->n_src * h * w >= (1ll << 31)) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d too large", s);
    // (byte offsets inside one pixel-interleaved image are formed exactly in fp32 by the gather of the HWC kernels)
    if (p.hwc && (long long)h * w * 12 >= (1ll << 24))
      return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, SFM_LAYOUT_HWC takes images of fewer than 2^24 / 12 pixels", s, h, w);
    if (!d->tgt[s] || !d->src[s] || !d->disp[s]) return fail(SFM_ERR_NULL, "sfm_loss: tgt/src/disp[%d] is NULL", s);
    if (p.expl && !d->mask_logits[s]) return fail(SFM_ERR_NULL, "sfm_loss: exp_reg > 0 but mask_logits[%d] is NULL", s);
    ScaleArgs& S = A.sc[s];
    S.tgt = d->tgt[s];
    S.src = d->src[s];
    S.disp = d->disp[s];
    S.mlog = p.expl ? d->mask_logits[s] : nullptr;
    S.warped = (need_loss && need_outputs) ? d->warped[s] : nullptr;
    if (grad && need_outputs) {
      if (!d->d_disp[s]) return fail(SFM_ERR_NULL, "sfm_loss: d_disp[%d] is NULL", s);
      if (p.expl && !d->d_mask[s]) return fail(SFM_ERR_NULL, "sfm_loss: exp_reg > 0 but d_mask[%d] is NULL", s);
      S.d_disp = d->d_disp[s];
      S.d_mask = p.expl ? d->d_mask[s] : nullptr;
      S.d_src = d->d_src[s];
    }
    S.h = h;
    S.w = w;
    S.strips = (w + sw - 1) / sw;
    S.chunk_rows = rows[s];
    S.chunks = (h + rows[s] - 1) / rows[s];
    S.tiles = S.strips * S.chunks;
    S.item_begin = items;
    A.tiles_of[s] = S.tiles;
    A.item_begin_of[s] = items;
    items += d->B * S.tiles;
    const double nb = (double)d->norm_B;
    S.inv_cnt = (float)(1.0 / (nb * 3.0 * h * w));
    const double wgt = (double)d->smooth_reg / (double)(1 << s);               // base_model.py:76
    S.c_dx2 = (float)(wgt / (nb * h * (w - 2)));
    S.c_dy2 = (float)(wgt / (nb * (h - 2) * w));
    S.c_dxy = (float)(wgt / (nb * (h - 1) * (w - 1)));
    S.c_ex = (float)(wgt / (nb * h * (w - 1)));
    S.c_ey = (float)(wgt / (nb * (h - 1) * w));
    S.c_exp = (float)((double)d->exp_reg / (nb * h * w));
  }
  A.items = items;
  set_gy(p, gy);
  // The workspace layout does not depend on the chunking (nor on the device): the two partial-sum arrays are placed
  // and sized for the largest item count any chunking can produce.
  const size_t cap = (size_t)max_items(d, sw);
  if ((size_t)items > cap) return fail(SFM_ERR_CONFIG, "sfm_loss: internal error: %d items exceed the bound %zu", items, cap);
  p.off_loss = 0;
  p.off_gpm = align_up(p.off_loss + cap * 4 * sizeof(float), 256);
  p.total = align_up(p.off_gpm + cap * d->n_src * 12 * sizeof(float), 256);
  return SFM_OK;
}

static void set_gy(Plan& p, const float gy) {
  LossArgs& A = p.args;
  A.gy = gy;
  for (int s = 0; s < A.n_scales; ++s) {
    ScaleArgs& S = A.sc[s];
    S.k_pix = gy * (1.0f - A.alpha) * S.inv_cnt;
    S.kq = -gy * A.alpha * S.inv_cnt;
    S.k_exp = gy * S.c_exp;
  }
}

static void bind_workspace(Plan& p, void* ws) {
  char* base = (char*)ws;
  p.args.part_loss = (float*)(base + p.off_loss);
  p.args.part_gpm = (float*)(base + p.off_gpm);
}

template <bool GRAD, bool LOSS>
static const void* kernel_ptr(bool ssim, bool expl, int smode, bool hwc, bool wide, bool warped) {
  // (WARPED only exists for the LOSS entry points: W = LOSS && warped is a constant false elsewhere, and those variants are not built)
#define SFM_KPICK(NAME, ...)                                                                                              \
  do {                                                                                                                    \
    if constexpr (LOSS) {                                                                                                 \
      if (warped) return hwc ? (const void*)&NAME<__VA_ARGS__, true, true> : (const void*)&NAME<__VA_ARGS__, false, true>; \
    }                                                                                                                     \
    return hwc ? (const void*)&NAME<__VA_ARGS__, true, false> : (const void*)&NAME<__VA_ARGS__, false, false>;            \
  } while (0)
  if (GRAD && wide && !ssim && !expl) {
    if (smode == 0) SFM_KPICK(loss_kernel_wide, LOSS, 0);
    else if (smode == 1) SFM_KPICK(loss_kernel_wide, LOSS, 1);
    else SFM_KPICK(loss_kernel_wide, LOSS, 2);
  }
  if (expl) {
    if (smode == 0) SFM_KPICK(loss_kernel, false, GRAD, LOSS, true, 0);
    else if (smode == 1) SFM_KPICK(loss_kernel, false, GRAD, LOSS, true, 1);
    else SFM_KPICK(loss_kernel, false, GRAD, LOSS, true, 2);
  } else if (ssim) {
    if (smode == 0) SFM_KPICK(loss_kernel, true, GRAD, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(loss_kernel, true, GRAD, LOSS, false, 1);
    else SFM_KPICK(loss_kernel, true, GRAD, LOSS, false, 2);
  } else {
    if (smode == 0) SFM_KPICK(loss_kernel, false, GRAD, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(loss_kernel, false, GRAD, LOSS, false, 1);
    else SFM_KPICK(loss_kernel, false, GRAD, LOSS, false, 2);
  }
#undef SFM_KPICK
}

// the same table for the launches that also produce dL/d(src) (loss_kernel_dsrc)
template <bool LOSS>
static const void* kernel_ptr_dsrc(bool ssim, bool expl, int smode, bool hwc, bool warped) {
#define SFM_KPICK(...)                                                                                                                   \
  do {                                                                                                                                   \
    if constexpr (LOSS) {                                                                                                                \
      if (warped) return hwc ? (const void*)&loss_kernel_dsrc<__VA_ARGS__, true, true> : (const void*)&loss_kernel_dsrc<__VA_ARGS__, false, true>; \
    }                                                                                                                                    \
    return hwc ? (const void*)&loss_kernel_dsrc<__VA_ARGS__, true, false> : (const void*)&loss_kernel_dsrc<__VA_ARGS__, false, false>;   \
  } while (0)
  if (expl) {
    if (smode == 0) SFM_KPICK(false, LOSS, true, 0);
    else if (smode == 1) SFM_KPICK(false, LOSS, true, 1);
    else SFM_KPICK(false, LOSS, true, 2);
  } else if (ssim) {
    if (smode == 0) SFM_KPICK(true, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(true, LOSS, false, 1);
    else SFM_KPICK(true, LOSS, false, 2);
  } else {
    if (smode == 0) SFM_KPICK(false, LOSS, false, 0);
    else if (smode == 1) SFM_KPICK(false, LOSS, false, 1);
    else SFM_KPICK(false, LOSS, false, 2);
  }
#undef SFM_KPICK
}

template <bool GRAD, bool LOSS>
static hipError_t launch_main(const Plan& p, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, const int variant) {
  LossArgs args = p.args;
  // the preloaded header (make_hdr): B, n_src, n_scales, prio_top and the tile counts packed
  // (sfm_loss_variant(3), tests: the header read from the struct, the path of counts beyond 16 bits)
  bool compact = variant != 3 && args.B <= 0xffff && args.n_src <= 15 && args.n_scales <= 15 && args.prio_top <= 3;
  for (int k = 0; k < SFM_MAX_SCALES; ++k) compact = compact && args.tiles_of[k] >= 0 && args.tiles_of[k] <= 0xffff;
  unsigned h_bn = 0, h_t[4] = {0, 0, 0, 0}, h_prio = args.prio_tab;
  int h_items = args.items, h_simds = args.simds_per_xcd;
  unsigned long long* h_trace = args.trace;
  if (compact) {
    h_bn = (unsigned)args.B | (unsigned)args.n_src << 16 | (unsigned)args.n_scales << 20 | (unsigned)args.prio_top << 24 | 1u << 31;
    for (int k = 0; k < SFM_MAX_SCALES; ++k) h_t[k >> 1] |= (unsigned)args.tiles_of[k] << (16 * (k & 1));
  }
  void* kargs[] = {&h_bn, &h_items, &h_simds, &h_prio, &h_t[0], &h_t[1], &h_t[2], &h_t[3], &h_trace, &args};
  // 8 x (items of the busiest XCD): see the item mapping at the top of loss_kernel
  int tiles_per_sample = 0;
  for (int s = 0; s < p.args.n_scales; ++s) tiles_per_sample += p.args.sc[s].tiles;
  const int per_xcd = (int)p.args.prio_tab >= 0 ? (p.args.B / 8) * tiles_per_sample + ((p.args.B % 8) * tiles_per_sample + 7) / 8 : (p.args.items + 7) / 8;
  const void* fn = kernel_ptr<GRAD, LOSS>(p.ssim, p.expl, p.smode, p.hwc, p.wide, p.warped);
  if constexpr (GRAD) {
    if (p.dsrc) fn = kernel_ptr_dsrc<LOSS>(p.ssim, p.expl, p.smode, p.hwc, p.warped);
  }
  if constexpr (GRAD && LOSS) {   // the reference-order variants (sfm_loss_variant): fused SSIM launch, pixel-interleaved, with smoothness
    if ((variant == 1 || variant == 2) && p.ssim && !p.expl && p.hwc && p.smode != 0) {
      const int key = (p.smode == 2 ? 4 : 0) | (p.warped ? 2 : 0) | (variant == 2 ? 1 : 0);
      const void* tab[8] = {(const void*)&loss_kernel_ref<1, false, 1>, (const void*)&loss_kernel_ref<1, false, 2>,
                            (const void*)&loss_kernel_ref<1, true, 1>,  (const void*)&loss_kernel_ref<1, true, 2>,
                            (const void*)&loss_kernel_ref<2, false, 1>, (const void*)&loss_kernel_ref<2, false, 2>,
                            (const void*)&loss_kernel_ref<2, true, 1>,  (const void*)&loss_kernel_ref<2, true, 2>};
      fn = tab[key];
    }
  }
  // With profiling events the kernel is launched through hipExtLaunchKernel: the events then carry the begin / end
  // timestamps of THIS dispatch (what rocprofv3's kernel trace reports), and no marker packets are put between the
  // launches of a step (hipEventRecord on either side of the kernel costs the step several microseconds).
  // dynamic LDS: the accumulation window of the optional dL/d(src) output (sfm_ssim_pass.h, dsrc_scatter), only when it is bound
  const size_t smem = (GRAD && p.dsrc) ? dsrc_tile_floats(p.ssim ? 8 : 4) * sizeof(float) : 0;
  if (ev_start && ev_stop) return hipExtLaunchKernel(fn, dim3(8 * per_xcd), dim3(64 * WAVES_PER_BLOCK), kargs, smem, st, ev_start, ev_stop, 0);
  return hipLaunchKernel(fn, dim3(8 * per_xcd), dim3(64 * WAVES_PER_BLOCK), kargs, smem, st);
}

// The plan of a descriptor depends only on the descriptor's bytes and the entry point: the last few are kept per
// thread, so that a training loop that calls with the same buffers every step does the validation and the chunk
// search once.  (Thread-local, like the error string: no state is shared between threads.)
struct CachedPlan {
  SfmLossDesc desc;
  int device;
  bool grad, loss, valid;
  Plan plan;
};
constexpr int PLAN_CACHE = 4;
static thread_local CachedPlan g_plans[PLAN_CACHE];
static thread_local unsigned g_plan_clock = 0;

static int cached_plan(const SfmLossDesc* d, bool grad, bool loss, float gy, Plan& out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  for (int k = 0; k < PLAN_CACHE; ++k) {
    CachedPlan& c = g_plans[k];
    if (c.valid && c.grad == grad && c.loss == loss && c.device == dev && memcmp(&c.desc, d, sizeof(SfmLossDesc)) == 0) {
      out = c.plan;
      set_gy(out, gy);
      return SFM_OK;
    }
  }
  if (int e = make_plan(d, grad, loss, true, gy, out)) return e;
  CachedPlan& c = g_plans[g_plan_clock++ % PLAN_CACHE];
  c.desc = *d; c.device = dev; c.grad = grad; c.loss = loss; c.plan = out; c.valid = true;
  return SFM_OK;
}

static int run(const SfmLossDesc* d, bool grad, bool loss, float gy, float* loss5, void* ws, size_t ws_bytes, void* stream,
               const char* who) {
  hipStream_t st = (hipStream_t)stream;
  // the one-call hooks are taken -- and forgotten -- here, whatever becomes of the call
  const int variant = g_variant;
  g_variant = 0;
  unsigned long long* const trace = g_trace;
  g_trace = nullptr;
  hipEvent_t ev_start = g_ev_start, ev_stop = g_ev_stop;
  g_ev_start = g_ev_stop = nullptr;
  if (d && d->B == 0) {   // empty shard: nothing to launch (input pointers of empty arrays may be NULL)
    if (loss) {
      if (!loss5) return fail(SFM_ERR_NULL, "%s: loss5 is NULL", who);
      hipError_t e = hipMemsetAsync(loss5, 0, 5 * sizeof(float), st);
      if (e != hipSuccess) return fail((int)e, "%s: memset: %s", who, hipGetErrorString(e));
    }
    return SFM_OK;
  }
  if (!d) return fail(SFM_ERR_NULL, "%s: NULL descriptor", who);
  Plan p;
  if (int e = cached_plan(d, grad, loss, gy, p)) return e;
  if (loss && !loss5) return fail(SFM_ERR_NULL, "%s: loss5 is NULL", who);
  if (!ws || ws_bytes < p.total) return fail(SFM_ERR_WORKSPACE, "%s: workspace of %zu bytes needed, got %zu", who, p.total, ws_bytes);
  if (((uintptr_t)ws & 255) != 0) return fail(SFM_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", who);
  bind_workspace(p, ws);
  p.args.trace = trace;
  hipError_t le;
  if (grad && loss) le = launch_main<true, true>(p, st, ev_start, ev_stop, variant);
  else if (grad) le = launch_main<true, false>(p, st, ev_start, ev_stop, variant);
  else le = launch_main<false, true>(p, st, ev_start, ev_stop, variant);
  if (le != hipSuccess) return fail((int)le, "%s: launch of the main kernel: %s", who, hipGetErrorString(le));
  const int n_pose_blocks = grad ? d->B * d->n_src : 0;
  {
    const LossArgs& a = p.args;
    bool compact = variant != 3 && a.B <= 0xffff && a.n_src <= 15 && a.n_scales <= 15;
    for (int k = 0; k < SFM_MAX_SCALES; ++k) compact = compact && a.tiles_of[k] >= 0 && a.tiles_of[k] <= 0xffff;
    unsigned tw[4] = {0, 0, 0, 0};
    if (compact)
      for (int k = 0; k < SFM_MAX_SCALES; ++k) tw[k >> 1] |= (unsigned)a.tiles_of[k] << (16 * (k & 1));
    const unsigned bsc = (compact ? ((unsigned)a.B | (unsigned)a.n_src << 16 | (unsigned)a.n_scales << 20 | FIN_COMPACT) : 0u) |
                         (grad ? FIN_POSE : 0u) | (loss ? FIN_LOSS : 0u);
    hipLaunchKernelGGL(finalize_kernel, dim3(n_pose_blocks + 1), dim3(64 * FINALIZE_WAVES), 0, st, (const float*)a.part_gpm, a.intrinsics,
                       a.pose[0], a.pose[1], bsc, tw[0], tw[1], tw[2], tw[3], (unsigned)(p.off_gpm - p.off_loss), p.args, loss ? loss5 : (float*)nullptr);
  }
  return check_launch(who);
}

}  // namespace sfm

extern "C" {

size_t sfm_loss_workspace_bytes(const SfmLossDesc* desc) {
  // the three entry points lay their work out differently: size for the largest
  size_t total = 0;
  const bool modes[3][2] = {{false, true}, {true, false}, {true, true}};
  for (int m = 0; m < 3; ++m) {
    sfm::Plan p;
    if (sfm::make_plan(desc, modes[m][0], modes[m][1], false, 1.f, p) != SFM_OK) return 0;
    if (p.total > total) total = p.total;
  }
  return total;
}

int sfm_loss_plan_info(const SfmLossDesc* desc, int grad, int loss, int* out, int n_out) {
  sfm::Plan p;
  if (int e = sfm::make_plan(desc, grad != 0, loss != 0, false, 1.f, p)) return e;
  if (!out || n_out < 1 + 4 * desc->n_scales) return sfm::fail(SFM_ERR_NULL, "sfm_loss_plan_info: out needs 1 + 4 * n_scales ints");
  out[0] = p.args.items;
  for (int s = 0; s < desc->n_scales; ++s) {
    const sfm::ScaleArgs& S = p.args.sc[s];
    int* o = out + 1 + 4 * s;
    o[0] = S.strips; o[1] = S.chunks; o[2] = S.chunk_rows; o[3] = S.tiles;
  }
  return SFM_OK;
}

int sfm_loss_variant(int variant) {
  if (variant < 0 || variant > 3) return sfm::fail(SFM_ERR_CONFIG, "sfm_loss_variant: %d not in [0, 3]", variant);
  sfm::g_variant = variant;
  return SFM_OK;
}

int sfm_loss_debug_trace(void* buf) {
  sfm::g_trace = (unsigned long long*)buf;
  return SFM_OK;
}

int sfm_loss_profile_events(void* ev_start, void* ev_stop) {
  sfm::g_ev_start = (hipEvent_t)ev_start;
  sfm::g_ev_stop = (hipEvent_t)ev_stop;
  return SFM_OK;
}

int sfm_loss_fwd(const SfmLossDesc* desc, float* loss5, void* ws, size_t ws_bytes, void* stream) {
  return sfm::run(desc, false, true, 1.f, loss5, ws, ws_bytes, stream, "sfm_loss_fwd");
}

int sfm_loss_bwd(const SfmLossDesc* desc, float gy, void* ws, size_t ws_bytes, void* stream) {
  return sfm::run(desc, true, false, gy, nullptr, ws, ws_bytes, stream, "sfm_loss_bwd");
}

int sfm_loss_fwd_bwd(const SfmLossDesc* desc, float* loss5, void* ws, size_t ws_bytes, void* stream) {
  return sfm::run(desc, true, true, 1.f, loss5, ws, ws_bytes, stream, "sfm_loss_fwd_bwd");
}

}  // extern "C"
