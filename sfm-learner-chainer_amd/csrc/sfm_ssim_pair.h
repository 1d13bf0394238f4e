// Photometric pass with SSIM for one wave and TWO sources at once (round 6; DESIGN.md 8-3b of round 5, round-5 verdict item 2):
// the inner loop of models/base_model.py:88-115 for sources 2p and 2p+1 of a sample in ONE walk down the wave's rows.
//
// ssim_source_pass (sfm_ssim_pass.h) walks a chunk once per source at three waves per SIMD.  Here a wave keeps the rings of two
// sources (two waves per SIMD, <= 256 registers) and shares per row what does not depend on the source:
//   * the disparity load and depth = 1 / disp, the target texel (one 12-byte load instead of two),
//   * stage B's target-side sums: Sy = sum I and sum I^2 (vertical sums once; Sy's horizontal 3-sum once -- seven pooled fields
//     instead of eight),
//   * the d_depth tile: one LDS store (or read-add-write) per row for both sources,
//   * the pass itself: one prologue / pose-sum epilogue instead of two, and -- the larger part -- a launch of two thirds as many
//     wavefronts with chunks 1.5 x taller, i.e. fewer halo rows recomputed (21 % -> 15 % of the row steps at BASELINE cfg3).
// Everything per source -- projection, gathers, bilinear taps, SSIM value and partials, gradient, pose sums -- is the code of the
// one-source pass, instantiated twice per row step: the same statements on the same values, so the results are those of the
// one-source kernels up to the order in which the two sources' shares of d_disp are added.
//
// MEASURED (profiles/r06_pair_kernel.txt): 10.9 % fewer vector instructions per BASELINE cfg3 launch (2.24e7 instead of 2.52e7), as
// designed -- and a launch that is 1 - 5 % LONGER there: the vector ALUs are busy 79 % of the launch instead of 85 %, because two
// waves per SIMD hide one another's latencies less well than three and the 64 wave slots per sample of that launch cannot be filled
// with items of equal length (a SIMD whose shorter wave has finished runs its longer one alone, at three quarters of the pace).
// Where the one-source plan has to cut short chunks to fill the chip (B <= 24 at 128x416) the taller chunks win: -2 ... -12 %.
// make_plan (sfm_loss.hip) therefore picks this form for those launches only.
#pragma once
#include "sfm_ssim_pass.h"

namespace sfm {

// Stage B for both sources at centre row rb (rows rb-1, rb, rb+1 in *2, *1, *0).  The target-side fields come from the FIRST
// source's ring (its RowS.it; the second source's is never written).
template <bool GRAD, bool LOSS>
__device__ __forceinline__ void ssim_stage_b_pair(const SsimCtx& C, const RowS& a2, const RowS& a1, const RowS& a0, const RowS& b2,
                                                  const RowS& b1, const RowS& b0, RowG& ga0, RowG& gb0, const bool count,
                                                  float& acc_pix, float& acc_ssim) {
  float ssum_a = 0.f, ssum_b = 0.f;
  const float kq2_a = C.kq * a1.nm, kq2_b = C.kq * b1.nm;
  SsimSums<f2> pa, pb;
  SsimSums<float> qa, qb;
  // vertical 3-sums (in-lane).  sigma_x + sigma_y only ever appear together (base_model.py:138): E[xx] + E[yy] is one field per
  // source, whose target half is formed once
  const f2 yy_p = vfma(a2.it.p, a2.it.p, vfma(a1.it.p, a1.it.p, a0.it.p * a0.it.p));
  const float yy_s = fmaf(a2.it.s, a2.it.s, fmaf(a1.it.s, a1.it.s, a0.it.s * a0.it.s));
  pa.Sy = a2.it.p + a1.it.p + a0.it.p;
  qa.Sy = a2.it.s + a1.it.s + a0.it.s;
  pa.Sx = a2.ih.p + a1.ih.p + a0.ih.p; qa.Sx = a2.ih.s + a1.ih.s + a0.ih.s;
  pb.Sx = b2.ih.p + b1.ih.p + b0.ih.p; qb.Sx = b2.ih.s + b1.ih.s + b0.ih.s;
  pa.Sqq = vfma(a2.ih.p, a2.ih.p, vfma(a1.ih.p, a1.ih.p, vfma(a0.ih.p, a0.ih.p, yy_p)));
  qa.Sqq = fmaf(a2.ih.s, a2.ih.s, fmaf(a1.ih.s, a1.ih.s, fmaf(a0.ih.s, a0.ih.s, yy_s)));
  pb.Sqq = vfma(b2.ih.p, b2.ih.p, vfma(b1.ih.p, b1.ih.p, vfma(b0.ih.p, b0.ih.p, yy_p)));
  qb.Sqq = fmaf(b2.ih.s, b2.ih.s, fmaf(b1.ih.s, b1.ih.s, fmaf(b0.ih.s, b0.ih.s, yy_s)));
  pa.Sxy = vfma(a2.ih.p, a2.it.p, vfma(a1.ih.p, a1.it.p, a0.ih.p * a0.it.p));
  qa.Sxy = fmaf(a2.ih.s, a2.it.s, fmaf(a1.ih.s, a1.it.s, a0.ih.s * a0.it.s));
  pb.Sxy = vfma(b2.ih.p, a2.it.p, vfma(b1.ih.p, a1.it.p, b0.ih.p * a0.it.p));
  qb.Sxy = fmaf(b2.ih.s, a2.it.s, fmaf(b1.ih.s, a1.it.s, b0.ih.s * a0.it.s));
  // horizontal 3-sums: seven fields x two channel groups in two runs of DPP adds (Sy once)
  hsum3_group(pa.Sx, pa.Sy, pa.Sqq, pa.Sxy, qa.Sx, qa.Sy, qa.Sqq, qa.Sxy);
  hsum3_group(pb.Sx, pb.Sqq, pb.Sxy, qb.Sx, qb.Sqq, qb.Sxy);
  pb.Sy = pa.Sy; qb.Sy = qa.Sy;
  ssim_value_partials<GRAD, LOSS, true>(pa, kq2_a, ga0.a.p, ga0.b.p, ga0.e.p, ssum_a);
  ssim_value_partials<GRAD, LOSS, false>(qa, kq2_a, ga0.a.s, ga0.b.s, ga0.e.s, ssum_a);
  ssim_value_partials<GRAD, LOSS, true>(pb, kq2_b, gb0.a.p, gb0.b.p, gb0.e.p, ssum_b);
  ssim_value_partials<GRAD, LOSS, false>(qb, kq2_b, gb0.a.s, gb0.b.s, gb0.e.s, ssum_b);
  if (GRAD) {
    hsum3_group(ga0.a.p, ga0.b.p, ga0.e.p, ga0.a.s, ga0.b.s, ga0.e.s);
    hsum3_group(gb0.a.p, gb0.b.p, gb0.e.p, gb0.a.s, gb0.b.s, gb0.e.s);
  }
  if (LOSS) {
    const float cf = count ? C.outf : 0.f;
    acc_ssim = fmaf(ssum_a, a1.nm * cf, acc_ssim);                    // base_model.py:114-115
    acc_ssim = fmaf(ssum_b, b1.nm * cf, acc_ssim);
    if (!GRAD) {
      const float ea = vabs_sum(a1.ih.p - a1.it.p) + vabs_sum(a1.ih.s - a1.it.s);
      const float eb = vabs_sum(b1.ih.p - a1.it.p) + vabs_sum(b1.ih.s - a1.it.s);
      acc_pix = fmaf(ea, a1.nm * cf, acc_pix);
      acc_pix = fmaf(eb, b1.nm * cf, acc_pix);
    }
  }
}

// Stage C for both sources at row rc (the row in a2 / b2): dL/dI^ -> dL/dq per source, the two shares of d_depth in ONE tile access.
template <bool LOSS>
__device__ __forceinline__ void ssim_stage_c_pair(const SsimCtx& Ca, const SsimCtx& Cb, const int rc, const RowS& a2, const RowS& b2,
                                                  const RowG& ga2, const RowG& ga1, const RowG& ga0, const RowG& gb2, const RowG& gb1,
                                                  const RowG& gb0, float* gacc, const bool first, PoseAcc& pma, PoseAcc& pmb,
                                                  float& acc_pix) {
  const float kpa = Ca.k_pix * a2.nm, kpb = Ca.k_pix * b2.nm;
  f2 gpa, dpa, gpb, dpb;
  float gsa, dsa, gsb, dsb;
  ssim_stage_c(ga2.a.p, ga1.a.p, ga0.a.p, ga2.b.p, ga1.b.p, ga0.b.p, ga2.e.p, ga1.e.p, ga0.e.p, a2.ih.p, a2.it.p, kpa, gpa, dpa);
  ssim_stage_c(ga2.a.s, ga1.a.s, ga0.a.s, ga2.b.s, ga1.b.s, ga0.b.s, ga2.e.s, ga1.e.s, ga0.e.s, a2.ih.s, a2.it.s, kpa, gsa, dsa);
  ssim_stage_c(gb2.a.p, gb1.a.p, gb0.a.p, gb2.b.p, gb1.b.p, gb0.b.p, gb2.e.p, gb1.e.p, gb0.e.p, b2.ih.p, a2.it.p, kpb, gpb, dpb);
  ssim_stage_c(gb2.a.s, gb1.a.s, gb0.a.s, gb2.b.s, gb1.b.s, gb0.b.s, gb2.e.s, gb1.e.s, gb0.e.s, b2.ih.s, a2.it.s, kpb, gsb, dsb);
  if (LOSS) {                                                         // the L1 terms of this row (base_model.py:95-100,:111)
    acc_pix = fmaf(vabs_sum(dpa) + vabs_sum(dsa), a2.nm * Ca.outf, acc_pix);
    acc_pix = fmaf(vabs_sum(dpb) + vabs_sum(dsb), b2.nm * Ca.outf, acc_pix);
  }
  const float yf = (float)rc;
  const float gd_a = geom_terms(Ca, a2.UV, a2.D, yf, contract_uv(a2, gpa, gsa), pma);
  const float gd_b = geom_terms(Cb, b2.UV, a2.D, yf, contract_uv(b2, gpb, gsb), pmb);
  float* ga = gacc + (rc - Ca.y0) * 64 + Ca.lane;
  if (first) *ga = gd_a + gd_b;
  else *ga = *ga + (gd_a + gd_b);
}

template <bool GRAD, bool LOSS, bool WARPED>
__device__ __forceinline__ void ssim_pair_row_step(const SsimCtx& Ca, const SsimCtx& Cb, const StepMasks& M, const int k, const int r,
                                                   Pipe& psa, Pipe& psb, float& disp_next, RowS& a0, const RowS& a1, const RowS& a2,
                                                   RowS& b0, const RowS& b1, const RowS& b2, RowG& ga0, const RowG& ga1, const RowG& ga2,
                                                   RowG& gb0, const RowG& gb1, const RowG& gb2, float* gacc, const bool first,
                                                   float& acc_pix, float& acc_ssim, PoseAcc& pma, PoseAcc& pmb) {
  // ---------------- A: finish row r of both sources, put row r+1 in flight ----------------
  finish_row<true>(Ca, psa, a0);
  finish_row<true>(Cb, psb, b0);
  if (!step_bit(M.fin, k)) { zero_rare(a0); zero_rare(b0); }
  if constexpr (WARPED) { store_warped_row(Ca, r, a0); store_warped_row(Cb, r, b0); }
  if (step_bit(M.iss, k)) {
    issue_row<true, 0, false>(Ca, r + 1, disp_next, psa);
    psb.D = psa.D;
    issue_row<true, 0, true>(Cb, r + 1, disp_next, psb);
    disp_next = ldf(Ca.dp, (unsigned)min(r + 2, Ca.h - 1) * (unsigned)Ca.w + Ca.xc);
  }
  // ---------------- B: SSIM at row r-1 ----------------
  if (step_bit(M.b, k)) ssim_stage_b_pair<GRAD, LOSS>(Ca, a2, a1, a0, b2, b1, b0, ga0, gb0, step_bit(M.cnt, k), acc_pix, acc_ssim);
  // ---------------- C: gradients at row r-2 ----------------
  if (GRAD) {
    if (step_bit(M.c, k)) ssim_stage_c_pair<LOSS>(Ca, Cb, r - 2, a2, b2, ga2, ga1, ga0, gb2, gb1, gb0, gacc, first, pma, pmb, acc_pix);
  }
}

// Sources (Ca, Cb) of one wave in one walk.  The contexts differ in the projection (M1, P3, mx), the source image (sp) and the
// optional warped output (wp) only.
template <bool GRAD, bool LOSS, bool WARPED>
__device__ __forceinline__ void ssim_pair_pass(const SsimCtx& Ca, const SsimCtx& Cb, float* gacc, const bool first, float& acc_pix,
                                               float& acc_ssim, float* gpm_out_a, float* gpm_out_b) {
  constexpr int HS = GRAD ? 2 : 1;
  const int rbeg = Ca.y0 - HS, rend = Ca.y1 + HS;
  const int rload = min(rend, Ca.h);
  PoseAcc pma, pmb;
  zero(pma); zero(pmb);
  RowS A0, A1, A2, B0, B1, B2;
  RowG GA0, GA1, GA2, GB0, GB1, GB2;
  Pipe psa, psb;
  float disp_next = 1.f;
  const int n = rend - rbeg, R = Ca.y1 - Ca.y0;
  StepMasks M;
  M.fin = step_range(-rbeg, Ca.h - rbeg);
  M.iss = step_range(-rbeg - 1, rload - rbeg - 1);
  M.b = step_range(HS + 1 - (GRAD ? 1 : 0), n);
  M.cnt = step_range(HS + 1, HS + 1 + R);
  M.c = step_range(HS + 2, HS + 2 + R);
  if (rbeg >= 0 && rbeg < Ca.h) {
    issue_row<true, 0, false>(Ca, rbeg, Ca.disp_first, psa);
    psb.D = psa.D;
    issue_row<true, 0, true>(Cb, rbeg, Ca.disp_first, psb);
  }
  disp_next = Ca.disp_second;
  for (int k = 0; k < n; k += 3) {
    const int r = rbeg + k;
    ssim_pair_row_step<GRAD, LOSS, WARPED>(Ca, Cb, M, k, r, psa, psb, disp_next, A0, A2, A1, B0, B2, B1, GA0, GA2, GA1, GB0, GB2, GB1, gacc,
                                           first, acc_pix, acc_ssim, pma, pmb);
    if (k + 1 < n)
      ssim_pair_row_step<GRAD, LOSS, WARPED>(Ca, Cb, M, k + 1, r + 1, psa, psb, disp_next, A1, A0, A2, B1, B0, B2, GA1, GA0, GA2, GB1, GB0, GB2,
                                             gacc, first, acc_pix, acc_ssim, pma, pmb);
    if (k + 2 < n)
      ssim_pair_row_step<GRAD, LOSS, WARPED>(Ca, Cb, M, k + 2, r + 2, psa, psb, disp_next, A2, A1, A0, B2, B1, B0, GA2, GA1, GA0, GB2, GB1, GB0,
                                             gacc, first, acc_pix, acc_ssim, pma, pmb);
  }
  if (GRAD) {
    pose_sums_raw(Ca, pma, gpm_out_a);
    pose_sums_raw(Cb, pmb, gpm_out_b);
  }
}

}  // namespace sfm
