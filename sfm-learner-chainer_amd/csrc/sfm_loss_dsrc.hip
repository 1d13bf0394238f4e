// The kernels of a call that also produces the optional dL/d(src) (SfmLossDesc.d_src): the gradient kernels that record dL/dI^
// (loss_kernel_dsrc) and dsrc_scatter_kernel, the second launch that turns the record into d_src.  A translation unit of its own so
// that it compiles next to sfm_loss.hip (make -j).
#include "sfm_loss_kernels.h"
#include <cstdio>
#include <cstdlib>

namespace sfm {

template <bool LOSS>
static const void* pick_dsrc(bool ssim, bool expl, int smode, bool hwc, bool warped) {
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

const void* kernel_ptr_dsrc(bool loss, bool ssim, bool expl, int smode, bool hwc, bool warped) {
  return loss ? pick_dsrc<true>(ssim, expl, smode, hwc, warped) : pick_dsrc<false>(ssim, expl, smode, hwc, warped);
}


// ------------------------------------------------------------------------------------------
// dsrc_scatter_kernel: see DsrcArgs in sfm_loss_kernels.h.
//
// A step of a workgroup is NW pixel rows of 64 lanes: G = NW >> seg_shift image rows of the band, one (row, segment) per wavefront.
// The window: three planes (one per channel) of win_rows source rows x win_cols columns of doubles, the rows as a ring (slot of row
// v = (sb + v - vb) mod win_rows), of which the first DA = win_rows - SL are ACTIVE (receive taps) and the last SL are slack --
// already flushed, all zero.  Between two steps the window moves by at most SL / 2 rows towards the mean tap row of the step before:
// the rows that leave the active part are flushed DURING the step, by the NF wavefronts that do nothing else, while the taps of the
// step go to the active rows -- disjoint slots, so one barrier per step is all the synchronisation.  Every slot has its own first
// column (cbs[slot], set while the slot is slack and clear, from the mean horizontal shift of the step before): the window follows
// the samples sideways as well, row by row, without a column ever having to move.
//
// What bounds this launch (profiles/r06_d_src.txt): the instruction stream of the sampling wavefronts -- two of them per SIMD, ~300 vector
// and ~140 scalar instructions per pixel row each; the flushing wavefronts wait at the barrier for four fifths of a step.  They are
// wavefronts of their own because loads and atomics share one in-order counter (a wavefront that waits for a load waits for every atomic
// it issued before it), and the flush itself -- 71 MB of dense float atomics per cfg3 step -- needs ~55 us of the chip's atomic rate.
// ------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) DsrcRec {      // dL/dI^ of one warped pixel, as the main launch stores it (geometry_backward)
  float c[3];
};
__device__ __forceinline__ float* at_off(float* base, const unsigned byte_off) {      // (a 32-bit offset from a wave-uniform base)
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off);
}
__device__ __forceinline__ void dsrc_lds_add(double* p, const double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// rows [v_first, v_first + n) of the window, in the slots s_first, s_first + 1, ... (mod DR) -> d_src, and cleared.  The (row, 64
// columns) pieces are dealt out over NF wavefronts (fwave = 0 .. NF-1), U pieces of a wavefront in flight together: their LDS reads
// queue behind the adds of the other wavefronts.
template <int NF>
__device__ __forceinline__ void dsrc_flush_rows(double* win, const int* cbs, const int DR, const int WC, const int nq, const int nq_inv16,
                                                const int s_first, const int v_first, const int n, float* dst, const int h, const int w,
                                                const unsigned P, const int fwave, const int lane) {
  const int PL = DR * WC;
  constexpr int U = 3;
  for (int k0 = fwave; k0 < n * nq; k0 += NF * U) {
    double r[U], g[U], bl[U];
    double* t[U];
    int vv[U], cc[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * NF;
      const int j = (k * nq_inv16) >> 16, q = k - j * nq;       // k / nq, k % nq (k < 4096, nq <= 16: exact)
      vv[u] = v_first + j;
      int slot = s_first + j;
      slot -= slot >= DR ? DR : 0;
      const int col = q * 64 + lane;
      // (a window placed around the taps can reach outside the image: such rows and columns never receive anything)
      on[u] = k < n * nq && (unsigned)vv[u] < (unsigned)h && col < WC;
      t[u] = win + (on[u] ? slot * WC + col : 0);
      cc[u] = (on[u] ? cbs[slot] : 0) + col;
      r[u] = g[u] = bl[u] = 0.0;
      if (on[u]) { r[u] = t[u][0]; g[u] = t[u][PL]; bl[u] = t[u][2 * PL]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (on[u] && (r[u] != 0.0 || g[u] != 0.0 || bl[u] != 0.0)) {
        t[u][0] = 0.0; t[u][PL] = 0.0; t[u][2 * PL] = 0.0;
        if ((unsigned)cc[u] < (unsigned)w) {      // (the test keeps every address inside the plane)
          float* o = dst + (unsigned)(vv[u] * w + cc[u]);
          if (r[u] != 0.0) atomicAdd(o, (float)r[u]);
          if (g[u] != 0.0) atomicAdd(o + P, (float)g[u]);
          if (bl[u] != 0.0) atomicAdd(o + 2 * P, (float)bl[u]);
        }
      }
    }
  }
}

// NW wavefronts that sample and add, NF that flush (and nothing else)
template <int NW, int NF, bool REF>
__global__ void __launch_bounds__(64 * (NW + NF)) dsrc_scatter_kernel(const DsrcArgs A) {
  extern __shared__ __attribute__((aligned(16))) double dsrc_win[];
  const int DR = A.win_rows, WC = A.win_cols, PL = DR * WC;
  // behind the window: stats[3][4] = (sum of v0 - y and of u0 - x over the in-view samples of a step, their number, -), then cbs[DR]
  int* hdr = reinterpret_cast<int*>(dsrc_win + (size_t)PL * 3);
  int* ctl = hdr + 12;      // ctl[3][2]: (vb, sb) of a step, written during the step before it
  int* cbs = hdr + 20;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool flusher = wave >= NW;
  int s = 0;
#pragma unroll
  for (int k = 1; k < SFM_MAX_SCALES; ++k)
    if (k < A.n_scales && (int)blockIdx.x >= A.sc[k].wg_begin && A.sc[k].bands > 0) s = k;
  const DsrcScale& S = A.sc[s];
  int idx = (int)blockIdx.x - S.wg_begin;
  const int band = idx % S.bands;
  idx /= S.bands;
  const int i = idx % A.n_src, b = idx / A.n_src;
  const int h = S.h, w = S.w;
  const unsigned P = (unsigned)h * (unsigned)w;
  const int nseg = 1 << S.seg_shift, G = NW >> S.seg_shift;
  const int seg = wave & (nseg - 1), rsub = wave >> S.seg_shift;
  const int xb = band * S.band_w;                       // first target column of the band
  const int x = xb + seg * 64 + lane;
  const bool xvalid = !flusher && (seg * 64 + lane < S.band_w) && (x < w);
  const float* rec = S.rec + (size_t)(b * A.n_src + i) * 3 * P;
  float* dst = S.d_src + (size_t)(b * A.n_src + i) * 3 * P;
  const float* dpl = S.disp + (size_t)b * P;
  const unsigned xc = (unsigned)min(x, w - 1);

  // the projection rows of (sample, scale, source): what loss_body hands its passes, for either projection
  WaveGeom WG = {0.f, 0.f, 0.f, 0.f};
  WaveGeomRef WGR = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!flusher) build_wave_geom_any<REF>(A.pose, A.n_src, b, A.intrinsics + (size_t)(b * A.n_scales + s) * 9, lane, WG, WGR);
  const float xf = (float)x;
  const ScaleConst sc = make_scale_const(h, w);
  SsimCtx C;       // (only the fields the projection reads are ever set or used)
  C.sc = sc;
  if constexpr (REF) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      C.Pm[k * 4 + 0] = from_lane(WGR.P0, 8 * i + k);
      C.Pm[k * 4 + 1] = from_lane(WGR.P1, 8 * i + k);
      C.Pm[k * 4 + 2] = from_lane(WGR.P2, 8 * i + k);
      C.Pm[k * 4 + 3] = from_lane(WGR.P3, 8 * i + k);
      C.Ki1[k] = from_lane(WGR.Ki1, k);
      C.Ki2[k] = from_lane(WGR.Ki2, k);
      C.mx[k] = from_lane(WGR.Ki0, k) * xf;
    }
    C.hw[0] = 0.5f * sc.wm1; C.hw[1] = 0.5f * sc.hm1;
    C.rhw[0] = uniform(rcp_refined(C.hw[0])); C.rhw[1] = uniform(rcp_refined(C.hw[1]));
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      C.M1[k] = from_lane(WG.M1, 8 * i + k);
      C.P3[k] = from_lane(WG.P3, 8 * i + k);
      C.mx[k] = fmaf(from_lane(WG.M0, 8 * i + k), xf, from_lane(WG.M2, 8 * i + k));
    }
  }

  for (int k = tid; k < PL * 3; k += 64 * (NW + NF)) dsrc_win[k] = 0.0;
  if (tid < 20) hdr[tid] = 0;
  __syncthreads();

  const int SL = max(2, min(2 * G, DR / 3)) & ~1;      // slack rows; the window moves by at most SL / 2 rows between two steps
  const int DA = DR - SL;
  const int steps = (h + G - 1) / G;

  // the sample of (row r, this lane): cell, fractions, in view -- the instructions of the main launch (issue_row / geometry_backward)
  auto sample = [&](const int r, const float disp) -> Proj {
    const float yf = (float)r;
    if constexpr (REF) {
      return ref_proj_cell(ref_position(C, yf, rcp_refined(disp)), h, w);
    } else {
      const float D = rcp(disp);                                                         // base_model.py:60, as issue_row
      const float a0 = fmaf(C.M1[0], yf, C.mx[0]), a1 = fmaf(C.M1[1], yf, C.mx[1]), a2 = fmaf(C.M1[2], yf, C.mx[2]);
      return project(a0, a1, a2, C.P3[0], C.P3[1], C.P3[2], D, sc, h, w);
    }
  };
  auto clampd = [](const int d) -> float { return (float)max(min(d, 2047), -2048); };
  // mean of a step's statistic, rounded down (the same instructions in every wavefront: the same value)
  auto mean_of = [](const int sum, const int n) -> int { return (int)floorf((float)sum * __builtin_amdgcn_rcpf((float)n)); };

  // the loads of a step are issued one step ahead
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, dsp = 1.f;
  if (!flusher) {
    const int r = rsub;
    // (clamped addresses, nothing selected on the loaded values: a select would wait for the load where it is issued; pixels outside
    //  the band or the image are masked where they are used)
    const unsigned o = (unsigned)min(r, h - 1) * (unsigned)w + xc;
    const DsrcRec gv = ld_off<DsrcRec>(rec, 12u * o);      // (32-bit byte offsets from wave-uniform bases: no 64-bit vector arithmetic)
    g0 = gv.c[0]; g1 = gv.c[1]; g2 = gv.c[2];
    dsp = ldf(dpl, o);
  }

  // placement: the first step's own samples (a pass without adds)
  int vb = 0, sb = 0, d_cur = 0;
  {
    const Proj p = sample(rsub, dsp);
    const bool act = xvalid && rsub < h && p.inview;
    const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
    if (cnt > 0) {
      float v2[2] = {act ? clampd(p.v0 - rsub) : 0.f, act ? clampd(p.u0 - x) : 0.f};
      wave_sums_lockstep(v2);
      if (lane == 63) {
        atomicAdd(&hdr[8], (int)v2[0]);
        atomicAdd(&hdr[9], (int)v2[1]);
        atomicAdd(&hdr[10], cnt);
      }
    }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(hdr[10]);
    const int cb0 = xb - A.margin + (n > 0 ? mean_of(__builtin_amdgcn_readfirstlane(hdr[9]), n) : 0);
    if (n > 0) vb = mean_of(__builtin_amdgcn_readfirstlane(hdr[8]), n) + (G + 1) / 2 - DA / 2;      // the window of step 0: around its own taps
    if (tid < DR) cbs[tid] = cb0;
    // (visible to everyone behind the barrier that ends step 0: no slot is read before -- see below -- except by step 0 itself)
    __syncthreads();
  }

  for (int g = 0; g < steps; ++g) {
    const int r = g * G + rsub;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, ndsp = 1.f;
    if (!flusher) {
      const int rn = r + G;
      const unsigned o = (unsigned)min(rn, h - 1) * (unsigned)w + xc;
      const DsrcRec gv = ld_off<DsrcRec>(rec, 12u * o);
      n0 = gv.c[0]; n1 = gv.c[1]; n2 = gv.c[2];
      ndsp = ldf(dpl, o);
    }
    if (flusher) {
      // the rows the move INTO this step's window (d_cur, decided a step ago) pushed out of its active part: to memory, now
      if (d_cur > 0) {                         // rows vb - d .. vb - 1 left at the top
        int sf = sb - d_cur;
        sf += sf < 0 ? DR : 0;
        dsrc_flush_rows<NF>(dsrc_win, cbs, DR, WC, A.nq, A.nq_inv16, sf, vb - d_cur, d_cur, dst, h, w, P, wave - NW, lane);
      } else if (d_cur < 0) {                  // rows vb + DA .. vb + DA - d - 1 left at the bottom
        int sf = sb + DA;
        sf -= sf >= DR ? DR : 0;
        dsrc_flush_rows<NF>(dsrc_win, cbs, DR, WC, A.nq, A.nq_inv16, sf, vb + DA, -d_cur, dst, h, w, P, wave - NW, lane);
      }
      // where the window goes NEXT: towards the mean tap row of the step before this one (complete: the barrier that ended it), two
      // steps of G rows further down.  Decided here, one step ahead and by the wavefronts that flush, so that the others find it ready.
      const int* st = hdr + 4 * ((g + 2) % 3);
      const int n = __builtin_amdgcn_readfirstlane(st[2]);
      int d_next = 0, cbn = 0;
      if (n > 0 && g + 1 < steps) {      // (after the last step the window stays: every wavefront flushes it from where it is)
        const int want = (g + 1) * G + mean_of(__builtin_amdgcn_readfirstlane(st[0]), n) + (G + 1) / 2 - DA / 2;
        d_next = max(min(want - vb, SL / 2), -(SL / 2));
      }
      if (n > 0) cbn = xb - A.margin + mean_of(__builtin_amdgcn_readfirstlane(st[1]), n);
      int sbn = sb + d_next;
      sbn -= sbn >= DR ? DR : 0;
      sbn += sbn < 0 ? DR : 0;
      if (wave == NW + NF - 1) {
        // the first column of the slack slots: the slots of the ring behind the active ones, except those being flushed right now
        // (d_cur > 0: the last d_cur of them; d_cur < 0: the first -d_cur) -- clear, read by nobody during this step; the rows that
        // enter the window with the next move (at most SL / 2 of them, at either end) are among them
        if (n > 0 && lane < SL && (d_cur > 0 ? lane < SL - d_cur : lane >= -d_cur)) {
          int slot = sb + DA + lane;
          slot -= slot >= DR ? DR : 0;
          cbs[slot] = cbn;
        }
        if (lane == 0) {
          int* c = ctl + 2 * ((g + 1) % 3);
          c[0] = vb + d_next; c[1] = sbn;
          int* z = hdr + 4 * ((g + 1) % 3); z[0] = 0; z[1] = 0; z[2] = 0;     // the stats of the NEXT step: nobody reads or adds there now
        }
      }
      vb += d_next; sb = sbn; d_cur = d_next;      // (the window of the next step; this wavefront has nothing else to do in this one)
    } else if (g > 0) {
      const int* c = ctl + 2 * (g % 3);
      vb = __builtin_amdgcn_readfirstlane(c[0]); sb = __builtin_amdgcn_readfirstlane(c[1]);
    }
    const Proj p = sample(r, dsp);
    const bool act = xvalid && r < h && p.inview;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(act);
    if (m != 0) {
      // the four taps: rows v0, v0 + 1 (slots s0, s1, each with its own first column), columns u0, u0 + 1
      const int rel = p.v0 - vb;
      const bool r0 = (unsigned)rel < (unsigned)DA, r1 = (unsigned)(rel + 1) < (unsigned)DA;
      int s0 = sb + (r0 ? rel : 0);
      s0 -= s0 >= DR ? DR : 0;
      int s1 = sb + (r1 ? rel + 1 : 0);
      s1 -= s1 >= DR ? DR : 0;
      const int cu0 = p.u0 - cbs[s0], cu1 = p.u0 - cbs[s1];
      // the statistics that place the window: from the wavefronts of every other image row of the step (all segments: the horizontal
      // shift varies along a row; half the samples of a step are plenty for a mean)
      if ((rsub & 1) == 0) {
        float v2[2] = {act ? (float)(p.v0 - r) : 0.f, act ? (float)(p.u0 - x) : 0.f};      // (in view: within the image, exact in fp32)
        wave_sums_lockstep(v2);                  // (under the latency of the two reads)
        if (lane == 63) {
          int* st = hdr + 4 * (g % 3);
          atomicAdd(&st[0], (int)v2[0]);
          atomicAdd(&st[1], (int)v2[1]);
          atomicAdd(&st[2], (int)__builtin_popcountll(m));
        }
      }
      const bool in0 = r0 && (unsigned)cu0 < (unsigned)(WC - 1), in1 = r1 && (unsigned)cu1 < (unsigned)(WC - 1);   // both columns of the row inside
      const int a0 = s0 * WC + cu0, a1 = s1 * WC + cu1;
      const float fu1 = 1.f - p.fu, fv1 = 1.f - p.fv;
      const float wt[4] = {fu1 * fv1, p.fu * fv1, fu1 * p.fv, p.fu * p.fv};
      const float gI[3] = {g0, g1, g2};
      // the taps inside the window: LDS adds on doubles, predicated tap by tap (in the usual case all four of every sample are inside)
      const bool in4[4] = {act && r0 && (unsigned)cu0 < (unsigned)WC, act && r0 && (unsigned)(cu0 + 1) < (unsigned)WC,
                           act && r1 && (unsigned)cu1 < (unsigned)WC, act && r1 && (unsigned)(cu1 + 1) < (unsigned)WC};
      const int ad4[4] = {a0, a0 + 1, a1, a1 + 1};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (in4[k]) {
          double* t = dsrc_win + ad4[k];
          dsrc_lds_add(t, (double)(gI[0] * wt[k]));
          dsrc_lds_add(t + PL, (double)(gI[1] * wt[k]));
          dsrc_lds_add(t + 2 * PL, (double)(gI[2] * wt[k]));
        }
      }
      // the taps outside it (above, below or beside): straight to memory, all of a sample's in one divergent block
      const bool miss = act && !(in0 && in1);
      if (__builtin_amdgcn_ballot_w64(miss) != 0) {
        if (miss) {
          const unsigned o4 = 4u * (unsigned)(p.v0 * w + p.u0), P4 = 4u * P;      // byte offsets into the three planes
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (!in4[k]) {
              const unsigned ok = o4 + 4u * (unsigned)((k >> 1) * w + (k & 1));
#pragma unroll
              for (int c = 0; c < 3; ++c) atomicAdd(at_off(dst, ok + c * P4), gI[c] * wt[k]);
            }
          }
        }
        if (A.counters) {      // diagnostics (SFM_DSRC_COUNT): samples of this row with a tap outside the window, by rows / by columns only
          const unsigned long long mr = __builtin_amdgcn_ballot_w64(act && !(r0 && r1)), mc = __builtin_amdgcn_ballot_w64(act && r0 && r1 && !(in0 && in1));
          if (lane == 0) { atomicAdd(A.counters + 1, (unsigned long long)__builtin_popcountll(mr)); atomicAdd(A.counters + 2, (unsigned long long)__builtin_popcountll(mc)); }
        }
      }
      if (A.counters && lane == 0) atomicAdd(A.counters, (unsigned long long)__builtin_popcountll(m));      // diagnostics: in-view samples
    }
    __syncthreads();
    g0 = n0; g1 = n1; g2 = n2; dsp = ndsp;
  }
  // what is left in the window (the slack rows are clear)
  dsrc_flush_rows<NW + NF>(dsrc_win, cbs, DR, WC, A.nq, A.nq_inv16, sb, vb, DA, dst, h, w, P, wave, lane);
}

hipError_t launch_dsrc_scatter(const DsrcArgs& a, hipStream_t st) {
  if (a.wgs <= 0) return hipSuccess;
  const size_t smem = dsrc_lds_bytes(a.win_rows, a.win_cols);
  const void* fn = a.ref ? (const void*)&dsrc_scatter_kernel<DSRC_WAVES, DSRC_FLUSH_WAVES, true> : (const void*)&dsrc_scatter_kernel<DSRC_WAVES, DSRC_FLUSH_WAVES, false>;
  static bool attr_set[64][2];      // per device; a benign race sets the same value twice
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = -1; }
  if (dev < 0 || !attr_set[dev][a.ref != 0]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, DSRC_LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0) attr_set[dev][a.ref != 0] = true;
  }
  DsrcArgs args = a;
  void* kargs[] = {&args};
  if (getenv("SFM_DSRC_COUNT")) {      // development: how many samples of this launch had a tap outside their window
    static int calls = 0;
    if (++calls == 5) {
      unsigned long long* dc = nullptr;
      (void)hipMalloc(&dc, 32);
      (void)hipMemsetAsync(dc, 0, 32, st);
      args.counters = dc;
      (void)hipLaunchKernel(fn, dim3(a.wgs), dim3(64 * (DSRC_WAVES + DSRC_FLUSH_WAVES)), kargs, smem, st);
      (void)hipStreamSynchronize(st);
      unsigned long long hc[4] = {0, 0, 0, 0};
      (void)hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost);
      fprintf(stderr, "dsrc counters: %llu in-view samples; with a tap outside the window by rows %llu, by columns only %llu; window %d rows x %d columns, margin %d\n",
              hc[0], hc[1], hc[2], a.win_rows, a.win_cols, a.margin);
      (void)hipFree(dc);
      return hipSuccess;
    }
  }
  return hipLaunchKernel(fn, dim3(a.wgs), dim3(64 * (DSRC_WAVES + DSRC_FLUSH_WAVES)), kargs, smem, st);
}

}  // namespace sfm
