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

#include "sfm_loss_kernels.h"

namespace sfm {

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
  size_t off_rec[SFM_MAX_SCALES];   // with d_src bound: the record of dL/dI^ of the scale (B, 3 n_src, h, w) that the second launch reads
  DsrcArgs dsrc_args;               // ... and that launch (dsrc_scatter_kernel, sfm_loss_dsrc.hip)
  bool ssim, expl, hwc;
  bool wide;     // the three-waves-per-SIMD build of an L1 gradient kernel (see loss_kernel)
  bool pair;     // two sources per pass at two waves per SIMD (loss_kernel_pair, sfm_ssim_pair.h)
  bool ref;      // SfmLossDesc.projection = SFM_PROJECTION_REFERENCE_ORDER: the kernels of sfm_loss_ref.hip
  bool dsrc;     // the call also produces dL/d(src): the instantiations that record dL/dI^, and the second launch
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

static int waves_per_simd_of(bool ssim, bool grad, bool wide, bool dsrc, bool pair) {   // = __launch_bounds__ of the kernel
  if (pair) return 2;
  (void)dsrc;    // (the kernels that record dL/dI^ for dL/d(src) run at the occupancy of the others)
  return ((ssim && grad) || wide) ? 3 : 4;
}

// tuning overrides (development only), read from the environment ONCE per process
struct Tuning {
  int chunk_rows = 0;                       // SFM_CHUNK_ROWS: fixed target height
  int rows_list[SFM_MAX_SCALES] = {0};      // SFM_CHUNK_ROWS_LIST: chunk height per scale, "13,13,16,8"
  bool has_prio = false;
  unsigned prio_tab = 0;                    // SFM_PRIO_TABLE: "0123,3210" = levels of ranks 0.. in phase 1, phase 2
  bool no_wide = false;                     // SFM_NO_WIDE: small L1 launches on the four-wave build too
  int pair = -1;                            // SFM_PAIR=0 / 1: never / whenever possible two sources per pass (default: make_plan decides)
  bool no_fill = false;                     // SFM_NO_FILL: no slot-filling refinement of the chunk heights (plan_chunks)
  int deal_below = 8;                       // SFM_DEAL_ITEMS_BELOW: batches smaller than this have their ITEMS dealt out over the XCDs (8 contiguous
                                            // ranges of the item list) instead of whole samples (b mod 8).  Measured at B = 8 (9): cfg5_2src +9.9 %,
                                            // cfg5 +3 %, cfg2 +1.6 %; items round-robin +22 .. +48 % (profiles/r05_wave_stage_stamps_cfg5_2src.txt)
  Tuning() {
    if (const char* e = getenv("SFM_DEAL_ITEMS_BELOW")) deal_below = atoi(e);
    no_wide = getenv("SFM_NO_WIDE") != nullptr;
    no_fill = getenv("SFM_NO_FILL") != nullptr;
    if (const char* e = getenv("SFM_PAIR")) pair = atoi(e) != 0;
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
  // (T: at most this many rows per chunk)
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

static void plan_chunks(const SfmLossDesc* d, int sw, int halo2, int slots, int passes /* walks of a wave down its chunk */, int* rows,
                        const int max_rows = MAX_CHUNK_ROWS) {
  const int forced = tuning().chunk_rows;
  int bestT = max_rows;
  double best = 1e300;
  for (int T = MIN_CHUNK_ROWS; T <= max_rows; ++T) {
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
  if (!tuning().no_fill && items <= slots && !(forced >= MIN_CHUNK_ROWS && forced <= MAX_CHUNK_ROWS) && (long long)maxcost * passes <= 40) {
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
    if (v >= MIN_CHUNK_ROWS && v <= max_rows) rows[k] = v;
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
static void plan_dsrc(const SfmLossDesc* d, const int cus, struct Plan& p);

constexpr int PAIR_BELOW_ROWS = 12;   // see make_plan

// validates the descriptor and lays out items + workspace for the given mode
static int make_plan(const SfmLossDesc* d, bool grad, bool need_loss, bool need_outputs, float gy, Plan& p, const int pair_hook = -1) {
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
  if (d->projection != SFM_PROJECTION_FAST && d->projection != SFM_PROJECTION_REFERENCE_ORDER)
    return fail(SFM_ERR_CONFIG, "sfm_loss: projection=%d", d->projection);
  p.ref = d->projection == SFM_PROJECTION_REFERENCE_ORDER;
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
  p.wide = grad && !p.dsrc && !p.ref && !p.ssim && !p.expl && !tuning().no_wide && max_items(d, sw) <= (long long)cus * 4 * 3;
  // Two sources per pass (loss_kernel_pair): the SSIM gradient launches of the pixel-interleaved layout with an even number of
  // sources, without the warped output.  By default where it measured faster (profiles/r06_pair_kernel.txt): launches whose
  // one-source plan has to cut the largest scale into chunks of at most PAIR_BELOW_ROWS rows to fill the chip (B <= 24 at 128x416:
  // -2 ... -12 % kernel time; the halo rows of short chunks are what the taller chunks of two thirds as many waves save).  At
  // BASELINE cfg3 / cfg5 (15 / 13 rows) it is 1 - 5 % SLOWER -- 11 % fewer vector instructions, issued 12 % less densely by two
  // waves per SIMD than by three -- and is not used.
  p.pair = false;
  if (grad && p.ssim && p.hwc && !p.dsrc && !p.ref && !p.warped && d->n_src % 2 == 0) {
    int rows1[SFM_MAX_SCALES];
    plan_chunks(d, sw, 2 * (hs > hm ? hs : hm), cus * 4 * 3, d->n_src, rows1);
    p.pair = rows1[0] <= PAIR_BELOW_ROWS;
    if (tuning().pair >= 0) p.pair = tuning().pair != 0;
    if (pair_hook >= 0) p.pair = pair_hook != 0;
  }
  const int waves_per_simd = waves_per_simd_of(p.ssim, grad, p.wide, p.dsrc, p.pair);
  const int slots = cus * 4 * waves_per_simd;
  A.simds_per_xcd = (cus % 8 == 0) ? cus / 8 * 4 : 128;   // gfx950: 8 XCDs, 4 SIMDs per CU
  A.prio_top = waves_per_simd - 1 < 3 ? waves_per_simd - 1 : 3;
  A.prio_tab = 0;
  for (int r = 0; r <= A.prio_top; ++r)   // youngest preferred in the first half of the sources, oldest in the second
    A.prio_tab |= (unsigned)r << (2 * r) | (unsigned)(A.prio_top - r) << (8 + 2 * r);
  if (tuning().has_prio) A.prio_tab = tuning().prio_tab & 0xffffu;
  if (d->B < (tuning().deal_below > 8 ? tuning().deal_below : 8)) A.prio_tab |= 0x80000000u;   // fewer samples than XCDs (or asked for): deal items
  plan_chunks(d, sw, 2 * (hs > hm ? hs : hm), slots, p.pair ? d->n_src / 2 : d->n_src, rows);
  int items = 0;
  for (int s = 0; s < d->n_scales; ++s) {
    const int h = d->H[s], w = d->W[s];
    if (h < 3 || w < 3) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, need H,W >= 3", s, h, w);
    if ((long long)d->B * 3 * d->n_src * h * w >= (1ll << 31)) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d too large", s);
    // (byte offsets inside one pixel-interleaved image are formed exactly in fp32 by the gather of the HWC kernels)
    if (p.hwc && (long long)h * w * 12 >= (1ll << 24))
      return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, SFM_LAYOUT_HWC takes images of fewer than 2^24 / 12 pixels", s, h, w);
    if (!d->tgt[s] || !d->src[s] || !d->disp[s]) return fail(SFM_ERR_NULL, "sfm_loss: tgt/src/disp[%d] is NULL", s);
    // (the second launch of a call with d_src addresses the record of one (sample, source) with 32-bit byte offsets)
    if (d->d_src[s] && (long long)h * w * 12 >= (1ll << 32)) return fail(SFM_ERR_SHAPE, "sfm_loss: scale %d is %dx%d, d_src takes images of fewer than 2^32 / 12 pixels", s, h, w);
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
  // dL/d(src): the record of dL/dI^ per scale that binds d_src (whatever the entry point: the workspace is sized from the descriptor)
  plan_dsrc(d, cus, p);
  return SFM_OK;
}

// The second launch of a call with d_src bound: bands, window and workgroups (DsrcArgs), and the place of the records in the workspace.
static void plan_dsrc(const SfmLossDesc* d, const int cus, Plan& p) {
  DsrcArgs& D = p.dsrc_args;
  memset(&D, 0, sizeof(D));
  D.B = d->B; D.n_src = d->n_src; D.n_scales = d->n_scales;
  D.intrinsics = d->intrinsics;
  D.ref = d->projection == SFM_PROJECTION_REFERENCE_ORDER;
  for (int i = 0; i < d->n_src; ++i) D.pose[i] = d->pose[i];
  // bands of at most max_segs 64-lane segments: as wide as possible (fewer overlapping windows) while the largest scale still gives
  // half the CUs a workgroup
  int max_segs = DSRC_MAX_SEGS;
  int lds_bytes = DSRC_LDS_BYTES;
  D.margin = DSRC_MARGIN;
  if (const char* e = getenv("SFM_DSRC_SEGS")) max_segs = atoi(e) >= 2 ? 2 : 1;        // (development)
  if (const char* e = getenv("SFM_DSRC_LDS_KB")) lds_bytes = atoi(e) * 1024 < lds_bytes ? atoi(e) * 1024 : lds_bytes;
  if (const char* e = getenv("SFM_DSRC_MARGIN")) D.margin = atoi(e);
  int s0 = -1;
  for (int s = 0; s < d->n_scales; ++s)
    if (d->d_src[s] && s0 < 0) s0 = s;
  if (s0 >= 0)
    while (max_segs > 1 && (long long)d->B * d->n_src * ((d->W[s0] + 64 * max_segs - 1) / (64 * max_segs)) < cus / 2) max_segs >>= 1;
  int wgs = 0, widest = 0;
  for (int s = 0; s < d->n_scales; ++s) {
    p.off_rec[s] = 0;
    DsrcScale& S = D.sc[s];
    S.wg_begin = wgs;
    if (!d->d_src[s]) continue;
    const int h = d->H[s], w = d->W[s];
    p.off_rec[s] = p.total;
    p.total = align_up(p.total + (size_t)d->B * d->n_src * 3 * h * w * sizeof(float), 256);
    S.disp = d->disp[s];
    S.d_src = d->d_src[s];
    S.h = h; S.w = w;
    S.bands = (w + 64 * max_segs - 1) / (64 * max_segs);
    S.band_w = (w + S.bands - 1) / S.bands;
    const int segs = (S.band_w + 63) / 64;
    S.seg_shift = segs > 1 ? 1 : 0;
    wgs += d->B * d->n_src * S.bands;
    if (S.band_w > widest) widest = S.band_w;
  }
  D.wgs = wgs;
  D.win_cols = widest + 2 * D.margin;
  if (D.win_cols < 16) D.win_cols = 16;
  // as many rows as the LDS holds, at most 64
  int rows = 64;
  while (rows > 8 && dsrc_lds_bytes(rows, D.win_cols) > (size_t)lds_bytes) --rows;
  D.win_rows = rows;
  D.nq = (D.win_cols + 63) / 64;
  D.nq_inv16 = (65536 + D.nq - 1) / D.nq;
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
  if (p.dsrc)
    for (int s = 0; s < p.args.n_scales; ++s)
      if (p.dsrc_args.sc[s].d_src) {
        p.args.sc[s].d_src = (float*)(base + p.off_rec[s]);       // the main launch writes the record ...
        p.dsrc_args.sc[s].rec = (const float*)(base + p.off_rec[s]);   // ... the second launch reads it
      }
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
  if (p.ref) fn = kernel_ptr_ref(GRAD, LOSS, p.ssim, p.expl, p.smode, p.hwc, p.warped);      // SFM_PROJECTION_REFERENCE_ORDER
  if (p.pair) fn = kernel_ptr_pair(GRAD, LOSS, p.smode);                                     // two sources per pass
  if constexpr (GRAD) {
    if (p.dsrc && !p.ref) fn = kernel_ptr_dsrc(LOSS, p.ssim, p.expl, p.smode, p.hwc, p.warped);      // (the REF kernels record dL/dI^ themselves)
  }
  // With profiling events the kernel is launched through hipExtLaunchKernel: the events then carry the begin / end
  // timestamps of THIS dispatch (what rocprofv3's kernel trace reports), and no marker packets are put between the
  // launches of a step (hipEventRecord on either side of the kernel costs the step several microseconds).
  const size_t smem = 0;
  if (ev_start && ev_stop) return hipExtLaunchKernel(fn, dim3(8 * per_xcd), dim3(64 * WAVES_PER_BLOCK), kargs, smem, st, ev_start, ev_stop, 0);
  return hipLaunchKernel(fn, dim3(8 * per_xcd), dim3(64 * WAVES_PER_BLOCK), kargs, smem, st);
}

// The plan of a descriptor depends only on the descriptor's bytes and the entry point: the last few are kept per
// thread, so that a training loop that calls with the same buffers every step does the validation and the chunk
// search once.  (Thread-local, like the error string: no state is shared between threads.)
struct CachedPlan {
  SfmLossDesc desc;
  int device, pair_hook;
  bool grad, loss, valid;
  Plan plan;
};
constexpr int PLAN_CACHE = 4;
static thread_local CachedPlan g_plans[PLAN_CACHE];
static thread_local unsigned g_plan_clock = 0;

static int cached_plan(const SfmLossDesc* d, bool grad, bool loss, float gy, Plan& out, const int pair_hook) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  for (int k = 0; k < PLAN_CACHE; ++k) {
    CachedPlan& c = g_plans[k];
    if (c.valid && c.grad == grad && c.loss == loss && c.device == dev && c.pair_hook == pair_hook && memcmp(&c.desc, d, sizeof(SfmLossDesc)) == 0) {
      out = c.plan;
      set_gy(out, gy);
      return SFM_OK;
    }
  }
  if (int e = make_plan(d, grad, loss, true, gy, out, pair_hook)) return e;
  CachedPlan& c = g_plans[g_plan_clock++ % PLAN_CACHE];
  c.desc = *d; c.device = dev; c.pair_hook = pair_hook; c.grad = grad; c.loss = loss; c.plan = out; c.valid = true;
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
  // (sfm_loss_variant 4 / 5: one source per pass / two sources per pass where possible, whatever the default -- in-process A/B, tests)
  if (int e = cached_plan(d, grad, loss, gy, p, variant == 4 ? 0 : (variant == 5 ? 1 : -1))) return e;
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
  if (grad && p.dsrc) {      // dL/d(src) from the record the main launch has just written
    le = launch_dsrc_scatter(p.dsrc_args, st);
    if (le != hipSuccess) return fail((int)le, "%s: launch of the d_src kernel: %s", who, hipGetErrorString(le));
  }
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
  if (variant != 0 && (variant < 3 || variant > 5)) return sfm::fail(SFM_ERR_CONFIG, "sfm_loss_variant: %d is not 0, 3, 4 or 5", variant);
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

// One step of SFMLearner.__call__ from the full-resolution frames (models/base_model.py:48-124) in ONE call: the loop head :69-72 --
// both pyramids, pixel-interleaved, into the buffers the descriptor binds as tgt[] / src[] -- then the fused loss.
static int step_from_frames(const float* tgt_full, const float* src_full, const SfmLossDesc* d, bool grad, float* loss5, void* ws,
                            size_t ws_bytes, void* stream, const char* who) {
  if (!d) return sfm::fail(SFM_ERR_NULL, "%s: NULL descriptor", who);
  if (d->image_layout != SFM_LAYOUT_HWC)
    return sfm::fail(SFM_ERR_CONFIG, "%s: the descriptor must bind pixel-interleaved pyramid buffers (SFM_LAYOUT_HWC): they are what this call writes", who);
  if (d->n_scales < 1 || d->n_scales > SFM_MAX_SCALES) return sfm::fail(SFM_ERR_SHAPE, "%s: n_scales=%d", who, d->n_scales);
  for (int s = 1; s < d->n_scales; ++s)
    if (d->H[s] != d->H[0] >> s || d->W[s] != d->W[0] >> s)
      return sfm::fail(SFM_ERR_SHAPE, "%s: scale %d is %dx%d, the pyramid of a %dx%d frame has %dx%d there (base_model.py:70)", who, s, d->H[s],
                       d->W[s], d->H[0], d->W[0], d->H[0] >> s, d->W[0] >> s);
  // (the descriptor's pyramid pointers are inputs of the loss and outputs of this call: the caller owns the buffers either way)
  if (int e = sfm_pyramid_pair_hwc_fwd(tgt_full, src_full, (float* const*)d->tgt, (float* const*)d->src, d->B, d->n_src, d->H[0], d->W[0],
                                       d->n_scales, stream))
    return e;
  return sfm::run(d, grad, true, 1.f, loss5, ws, ws_bytes, stream, who);
}

int sfm_step_fwd(const float* tgt_full, const float* src_full, const SfmLossDesc* desc, float* loss5, void* ws, size_t ws_bytes, void* stream) {
  return step_from_frames(tgt_full, src_full, desc, false, loss5, ws, ws_bytes, stream, "sfm_step_fwd");
}

int sfm_step_fwd_bwd(const float* tgt_full, const float* src_full, const SfmLossDesc* desc, float* loss5, void* ws, size_t ws_bytes,
                     void* stream) {
  return step_from_frames(tgt_full, src_full, desc, true, loss5, ws, ws_bytes, stream, "sfm_step_fwd_bwd");
}

}  // extern "C"
