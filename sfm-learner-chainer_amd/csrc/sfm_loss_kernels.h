// The kernels of the fused multi-scale loss (see sfm_loss.hip for the execution model): argument block, smoothness passes, the wave
// body and the __global__ templates.  Included by the translation units that instantiate them -- sfm_loss.hip (the product's
// projection), sfm_loss_ref.hip (SFM_PROJECTION_REFERENCE_ORDER) and sfm_loss_dsrc.hip (launches that also produce dL/d(src)) -- so
// that the ~300 instantiations compile in parallel.
#pragma once
#include "sfm_common.h"
#include "sfm_ssim_pass.h"
#include "sfm_ssim_pair.h"

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
  float* d_src;                // with d_src bound: the RECORD of dL/dI^ (B, n_src, h, w, 3) in the workspace that dsrc_scatter_kernel reads
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
// REF: the projection of the source passes in the reference's own evaluation order per pixel (SfmLossDesc.projection =
// SFM_PROJECTION_REFERENCE_ORDER; sfm_ssim_pass.h, ref_position) -- a test hook of one kernel in round 5, every launch since round 6.
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

// PAIR: two sources per pass (sfm_ssim_pair.h): the SSIM kernels of the pixel-interleaved layout with the product's projection, for
// an even number of sources; two waves per SIMD.
template <bool SSIM, bool GRAD, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED, int REF = 0, bool DSRC = false, bool PAIR = false>
__device__ __forceinline__ void loss_body(const Hdr& H, const LossArgs& A) {
  static_assert(!PAIR || (SSIM && HWC && !EXPL && REF == 0 && !DSRC), "two sources per pass: the SSIM kernels, pixel-interleaved, FAST projection");
  static_assert(GRAD || !DSRC, "dL/d(src) is an output of the backward");
  static_assert(LOSS || !WARPED, "the warped images are an output of the forward and the fused entry points");
  using HH = Halo<SSIM, GRAD, SMODE>;
  constexpr int TILE_ROWS = MAX_CHUNK_ROWS;
  __shared__ float gacc_all[GRAD ? WAVES_PER_BLOCK * TILE_ROWS * 64 : 64];
  const int wave = threadIdx.x >> 6;
  float* gacc = gacc_all + (GRAD ? wave * TILE_ROWS * 64 : 0);

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
  WaveGeom WG;
  WaveGeomRef WGR;
  build_wave_geom_any<REF != 0>(A.pose, H.n_src, b, A.intrinsics + (size_t)(b * H.n_scales + s) * 9, lane, WG, WGR);
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
  const int n_units = PAIR ? (H.n_src >> 1) : H.n_src;        // passes over the sources: one per source, or one per PAIR of sources
  const int n_phases = n_units + (SMODE != 0 ? 1 : 0);
  for (int ph = 0; ph < n_phases; ++ph) {
    const int iu = (SMODE != 0 && !smooth_last) ? ph - 1 : ph;  // unit of this phase; -1 or n_units = the smoothness pass
    const int i = PAIR ? 2 * iu : iu;                           // (first) source of the unit
#ifdef SFM_STAMPS
    unsigned long long tp0 = 0, tp1 = 0;
    SFM_STAMP(tp0);
#endif
    if (SMODE != 0 && (iu < 0 || iu >= n_units)) {
      if (SMODE == 1) smooth2_pass<GRAD, LOSS>(A, S, S.disp + (size_t)b * P, lane, x, xin, outl, y0, y1, gacc, acc_sm, !first);
      else smooth_edge_pass<GRAD, LOSS, HWC>(A, S, S.disp + (size_t)b * P, S.tgt + (size_t)b * 3 * P, lane, x, xin, outl, y0, y1, gacc, acc_sm, !first);
      first = false;
#ifdef SFM_STAMPS
      SFM_STAMP(tp1);
      cyc_smooth += tp1 - tp0;
#endif
      continue;
    }
    if (iu * 2 >= n_units) set_issue_prio((int)((H.prio_tab >> (8 + 2 * prio_rank)) & 3u));
    SsimCtx C;
    const float xf = (float)x;
    C.x0 = x - lane;
    if constexpr (REF != 0) {
      // the reference's chain takes Pm = K4 . T and K^-1 themselves (lane 8 i + k: row k of source i; K^-1 is the same for every source)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        C.Pm[k * 4 + 0] = from_lane(WGR.P0, 8 * i + k);
        C.Pm[k * 4 + 1] = from_lane(WGR.P1, 8 * i + k);
        C.Pm[k * 4 + 2] = from_lane(WGR.P2, 8 * i + k);
        C.Pm[k * 4 + 3] = from_lane(WGR.P3, 8 * i + k);
        C.Ki1[k] = from_lane(WGR.Ki1, k);
        C.Ki2[k] = from_lane(WGR.Ki2, k);
        C.mx[k] = from_lane(WGR.Ki0, k) * xf;      // Kinv[k][0] x: one rounding, as in the reference's K^-1 . pix (transform.py:105)
        C.M1[k] = C.P3[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) C.P3[k] = C.Pm[k * 4 + 3];   // (the backward's dL/d(disp) = (gq . P3) D, geometry_backward)
      C.hw[0] = 0.5f * sc.wm1; C.hw[1] = 0.5f * sc.hm1;       // (W-1) / 2., (H-1) / 2.   transform.py:124-125
      C.rhw[0] = uniform(rcp_refined(C.hw[0])); C.rhw[1] = uniform(rcp_refined(C.hw[1]));
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
    if constexpr (PAIR) {
      // the second source of the pair: the same context with ITS projection rows, image and outputs
      SsimCtx Cb = C;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        Cb.M1[k] = from_lane(WG.M1, 8 * (i + 1) + k);
        Cb.P3[k] = from_lane(WG.P3, 8 * (i + 1) + k);
        Cb.mx[k] = fmaf(from_lane(WG.M0, 8 * (i + 1) + k), xf, from_lane(WG.M2, 8 * (i + 1) + k));
        Cb.sp[k] = C.sp[k] + 3 * P;
      }
      Cb.wp = WARPED ? C.wp + 3 * P : nullptr;
      ssim_pair_pass<GRAD, LOSS, WARPED>(C, Cb, gacc, first, acc_pix, acc_ssim, gpm_out, GRAD ? gpm_out + 12 : nullptr);
    } else if constexpr (SSIM) {
      ssim_source_pass<GRAD, LOSS, HWC, WARPED, REF, DSRC>(C, gacc, first, acc_pix, acc_ssim, gpm_out SFM_STAMPS_PASS);
    } else {
      l1_source_pass<GRAD, LOSS, EXPL, HWC, WARPED, REF, DSRC>(C, gacc, first, acc_pix, acc_exp, gpm_out);
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
// (DSRC, see loss_body: the gradient kernels of a launch that also wants dL/d(src): the same kernels plus three stores per pixel row
//  and source, the record of dL/dI^ that dsrc_scatter_kernel turns into d_src)
template <bool SSIM, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, SSIM ? 3 : 4) loss_kernel_dsrc(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<SSIM, true, LOSS, EXPL, SMODE, HWC, WARPED, 0, true>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (REF, see loss_body: every launch of a descriptor with projection = SFM_PROJECTION_REFERENCE_ORDER; sfm_loss_ref.hip instantiates them)
template <bool SSIM, bool GRAD, bool LOSS, bool EXPL, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, (SSIM && GRAD) ? 3 : 4) loss_kernel_ref(SFM_HDR_PARAMS, const LossArgs A) {
  // (DSRC = GRAD: these kernels record dL/dI^ whenever the descriptor binds d_src -- a run-time test of one pointer and three stores)
  loss_body<SSIM, GRAD, LOSS, EXPL, SMODE, HWC, WARPED, 1, GRAD>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (PAIR, see loss_body: two sources per pass at two waves per SIMD; sfm_loss_pair.hip instantiates them)
template <bool GRAD, bool LOSS, int SMODE>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 2) loss_kernel_pair(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<true, GRAD, LOSS, false, SMODE, true, false, 0, false, true>(make_hdr(A, SFM_HDR_ARGS), A);
}
// (WIDE, see above: L1 gradient kernels only)
template <bool LOSS, int SMODE, bool HWC, bool WARPED = false>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, 3) loss_kernel_wide(SFM_HDR_PARAMS, const LossArgs A) {
  loss_body<false, true, LOSS, false, SMODE, HWC, WARPED>(make_hdr(A, SFM_HDR_ARGS), A);
}


// the kernel tables of the other translation units (sfm_loss_ref.hip, sfm_loss_dsrc.hip); nullptr = not built
const void* kernel_ptr_ref(bool grad, bool loss, bool ssim, bool expl, int smode, bool hwc, bool warped);
const void* kernel_ptr_dsrc(bool loss, bool ssim, bool expl, int smode, bool hwc, bool warped);
const void* kernel_ptr_pair(bool grad, bool loss, int smode);

// ------------------------------------------------------------------------------------------
// The second launch of a call with SfmLossDesc.d_src bound (sfm_loss_dsrc.hip): dL/d(src) from the record of dL/dI^.
// One workgroup per (scale, sample, source, band of columns) walks down ITS band of the target image, DSRC_WAVES pixel rows of 64
// lanes at a time, re-projects the pixels (the projection of the main launch: same inputs, same instructions, same
// cells and fractions) and adds the four taps of every in-view sample into a window of the source image held in LDS as DOUBLES
// (ds_add_f64: 3 - 4 ticks per wave instruction and CU, against 100+ for ds_add_f32: profiles/r06_lds_atomic_cost.txt): win_rows source
// rows x win_cols columns x 3 channels that follow the mean tap row of the band.  A row that leaves the window is added to d_src once
// (float atomics: bands and windows of other workgroups overlap it) and cleared; taps outside the window go to memory directly.
// ------------------------------------------------------------------------------------------
constexpr int DSRC_WAVES = 8;             // wavefronts per workgroup that sample and add: one pixel row of 64 lanes each per step
constexpr int DSRC_FLUSH_WAVES = 4;       // ... and that only flush
constexpr int DSRC_MAX_SEGS = 2;          // a band is at most this many 64-lane segments wide
constexpr int DSRC_MARGIN = 16;           // window columns on either side of the band (on top of the mean horizontal shift, which the window follows)
constexpr int DSRC_LDS_BYTES = 156 * 1024;  // of the CU's 160 KiB: one workgroup per CU
__host__ __device__ constexpr size_t dsrc_lds_bytes(const int rows, const int cols) {      // window, 20 ints of statistics and control, a first column per row
  return (size_t)rows * cols * 3 * sizeof(double) + (20 + (size_t)rows) * sizeof(int); 
}
struct DsrcScale {
  const float* rec;    // (B, n_src, h, w, 3): dL/dI^ of every warped pixel, written by the main launch
  const float* disp;   // (B, 1, h, w)
  float* d_src;        // (B, 3 n_src, h, w), accumulated into
  int h, w;
  int bands, band_w;   // column bands per image, target columns per band
  int seg_shift;       // log2 of the 64-lane segments per band row (1 or 2 of them): DSRC_WAVES >> seg_shift rows per step
  int wg_begin;        // first workgroup of the scale (largest scale first)
};
struct DsrcArgs {
  DsrcScale sc[SFM_MAX_SCALES];
  const float* pose[SFM_MAX_SRC];
  const float* intrinsics;
  int B, n_src, n_scales;
  int win_rows, win_cols;   // the LDS window: source rows (ring) x columns
  int margin;               // window columns on either side of the band
  int nq, nq_inv16;         // 64-column pieces of a window row, and ceil(2^16 / nq)
  int wgs;                  // workgroups
  int ref;                  // SfmLossDesc.projection = SFM_PROJECTION_REFERENCE_ORDER: the samples in the reference's evaluation order
  unsigned long long* counters;   // development (SFM_DSRC_COUNT): samples with taps outside the window; normally nullptr
};
hipError_t launch_dsrc_scatter(const DsrcArgs& a, hipStream_t st);

}  // namespace sfm
