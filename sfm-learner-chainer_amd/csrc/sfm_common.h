// Shared device helpers for libsfmwarp (gfx950 / CDNA4 only: 64-lane wavefronts, DPP wave shifts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/sfmwarp.h"

namespace sfm {

// ------------------------------------------------------------------------------------------
// error plumbing (host)
// ------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

// ------------------------------------------------------------------------------------------
// Geometry of one (sample, scale, source): everything the per-pixel projection needs,
// 32 floats so that a wave fetches it with scalar loads.
//   P    rows 0..2 of Pm = K4 . T                       (models/transform.py:86-88)
//   Kinv batch_inv(K)                                   (models/transform.py:105)
//   M    P[:, :3] . Kinv, so that q = D * (M . (x,y,1)) + P[:,3]  -- algebraically
//        Pm . (D * Kinv . pix, 1) of transform.py:105-108,122
// ------------------------------------------------------------------------------------------
struct Geom {
  float P[12];
  float Kinv[9];
  float M[9];
  float pad[2];
};
static_assert(sizeof(Geom) == 128, "Geom is 32 floats");

struct Rot {  // euler2mat intermediates, kept for the pose backward
  float c[3], s[3];
  float X[9], Y[9], Z[9], XY[9], R[9];
};

__device__ __forceinline__ void mat3_mul(const float* a, const float* b, float* o) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

__device__ __forceinline__ void mat3_mul_tn(const float* a, const float* b, float* o) {  // a^T . b
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[0 * 3 + i] * b[0 * 3 + j] + a[1 * 3 + i] * b[1 * 3 + j] + a[2 * 3 + i] * b[2 * 3 + j];
}

__device__ __forceinline__ void mat3_mul_nt(const float* a, const float* b, float* o) {  // a . b^T
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
}

// sin and cos of an angle in [-pi, pi] (euler2mat clips to that range, transform.py:23): one Cody-Waite reduction by pi/2 (two
// constants, fused multiply-adds: the reduced argument is exact to 1e-15) and the single-precision minimax kernels on
// [-pi/4, pi/4] (Cephes sinf / cosf; < 1 ulp there).  The library's sincosf spends 500 of its 600 instructions on arguments this
// path never sees (Payne-Hanek reduction), and the geometry and finalize kernels are single chains of dependent instructions:
// 0.6 - 1.6 us of a 64 us step (profiles/r03_ab_small_kernels.txt).
__device__ __forceinline__ void sincos_pi(const float a, float* sn, float* cs) {
  const float kf = rintf(a * 0.636619772367581343f);            // quadrant: -2 .. 2
  float y = fmaf(-kf, 1.57079637050628662109375f, a);
  y = fmaf(-kf, -4.37113900018624283e-8f, y);
  const float z = y * y;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * y, y);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf & 3;                                     // two's complement: -1 -> 3, -2 -> 2
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

// euler2mat, models/transform.py:11-40:  R = (X . Y) . Z, angles clipped to [-pi, pi]
__device__ __forceinline__ void euler2mat(const float* r, Rot& o) {
  const float pi = 3.14159265358979323846f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = fminf(fmaxf(r[k], -pi), pi);
    sincos_pi(a, &o.s[k], &o.c[k]);
  }
  const float Z[9] = {o.c[2], -o.s[2], 0.f, o.s[2], o.c[2], 0.f, 0.f, 0.f, 1.f};
  const float Y[9] = {o.c[1], 0.f, o.s[1], 0.f, 1.f, 0.f, -o.s[1], 0.f, o.c[1]};
  const float X[9] = {1.f, 0.f, 0.f, 0.f, o.c[0], -o.s[0], 0.f, o.s[0], o.c[0]};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    o.X[k] = X[k];
    o.Y[k] = Y[k];
    o.Z[k] = Z[k];
  }
  mat3_mul(o.X, o.Y, o.XY);
  mat3_mul(o.XY, o.Z, o.R);
}

// batch_inv for one 3x3 (models/transform.py:105): adjugate / determinant
__device__ __forceinline__ void inv3(const float* K, float* o) {
#pragma clang fp contract(off)
  const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const float det = a * A + b * B + c * C;
  o[0] = A / det;
  o[1] = -(b * i - c * h) / det;
  o[2] = (b * f - c * e) / det;
  o[3] = B / det;
  o[4] = (a * i - c * g) / det;
  o[5] = -(a * f - c * d) / det;
  o[6] = C / det;
  o[7] = -(a * h - b * g) / det;
  o[8] = (a * e - b * d) / det;
}

// proj_tgt_to_src (models/transform.py:64-91) for one sample; rows 0..2 of K4 . [R|t]
__device__ __forceinline__ void make_geom(const float* pose6, const float* K, Geom& g) {
#pragma clang fp contract(off)
  Rot rot;
  euler2mat(pose6, rot);
  const float t[3] = {pose6[3], pose6[4], pose6[5]};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      g.P[i * 4 + j] = K[i * 3 + 0] * rot.R[0 * 3 + j] + K[i * 3 + 1] * rot.R[1 * 3 + j] + K[i * 3 + 2] * rot.R[2 * 3 + j];
    g.P[i * 4 + 3] = K[i * 3 + 0] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
  }
  inv3(K, g.Kinv);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      g.M[i * 3 + j] = g.P[i * 4 + 0] * g.Kinv[0 * 3 + j] + g.P[i * 4 + 1] * g.Kinv[1 * 3 + j] + g.P[i * 4 + 2] * g.Kinv[2 * 3 + j];
  g.pad[0] = g.pad[1] = 0.f;
}

// Backward of proj_tgt_to_src for one sample.
//   gT3 = (K^T . gPm[0:3, :]) accumulated by the caller over scales: 3x4, row-major
//   rot = euler2mat(pose6): taken as an argument so that a caller can compute it while the sums behind gT3 are still in flight
__device__ __forceinline__ void pose_backward(const float* pose6, const Rot& rot, const float* gT3, float* d_pose6) {
  const float pi = 3.14159265358979323846f;
  float gR[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) gR[i * 3 + j] = gT3[i * 4 + j];
  float gXY[9], gZ[9], gX[9], gY[9];
  mat3_mul_nt(gR, rot.Z, gXY);    // gXY = gR . Z^T
  mat3_mul_tn(rot.XY, gR, gZ);    // gZ  = XY^T . gR
  mat3_mul_nt(gXY, rot.Y, gX);    // gX  = gXY . Y^T
  mat3_mul_tn(rot.X, gXY, gY);    // gY  = X^T . gXY
  const float gcos[3] = {gX[4] + gX[8], gY[0] + gY[8], gZ[0] + gZ[4]};
  const float gsin[3] = {gX[7] - gX[5], gY[2] - gY[6], gZ[3] - gZ[1]};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float g = -rot.s[k] * gcos[k] + rot.c[k] * gsin[k];
    d_pose6[k] = (pose6[k] > -pi && pose6[k] < pi) ? g : 0.f;   // F.clip backward (transform.py:23)
    d_pose6[3 + k] = gT3[k * 4 + 3];
  }
}

__device__ __forceinline__ void pose_backward(const float* pose6, const float* gT3, float* d_pose6) {
  Rot rot;
  euler2mat(pose6, rot);
  pose_backward(pose6, rot, gT3, d_pose6);
}

// ------------------------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// a / b from r ~= 1/b (v_rcp_f32, 1 ulp) with one residual correction: the result is the
// correctly rounded quotient except in rare double-rounding cases, at 3 instructions instead
// of the ~10 of the IEEE division sequence.  Keeps exact quotients exact ((w-1)*D/D == w-1), which
// is what decides the reference's strict `-1 < x < 1` test (models/transform.py:129) on the
// image border.
__device__ __forceinline__ float div_r(float a, float b, float r) {
  const float q = a * r;
  const float e = fmaf(-q, b, a);
  return fmaf(e, r, q);
}
// 1 / b, refined the same way
__device__ __forceinline__ float rcp_refined(float b) {
  const float r = rcp(b);
  const float e = fmaf(-b, r, 1.0f);
  return fmaf(e, r, r);
}

// lane l receives the value of lane l-1; lane 0 receives 0   (DPP wave_shr:1)
__device__ __forceinline__ float from_left(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
// lane l receives the value of lane l+1; lane 63 receives 0  (DPP wave_shl:1)
__device__ __forceinline__ float from_right(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float hsum3(float x) {
  asm("" : "+v"(x));   // keep x a materialised value: an FMA contraction of its producer into the adds would block the DPP operand folding
  return (x + from_left(x)) + from_right(x);
}

// Sum over the 64 lanes, entirely in the VALU (DPP row shifts + row broadcasts); the result is valid
// in lane 63 and returned as a wave-uniform value.  (A __shfl_xor butterfly goes through
// ds_bpermute: ~100+ cycles of LDS latency per step, 6 dependent steps per value.)
__device__ __forceinline__ float wave_sum(float v) {
  int x = __builtin_bit_cast(int, v);
#define SFM_DPP_ADD(ctrl, rmask, bmask)                                                                      \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, bmask, true))
  SFM_DPP_ADD(0x111, 0xf, 0xf);   // row_shr:1
  SFM_DPP_ADD(0x112, 0xf, 0xf);   // row_shr:2
  SFM_DPP_ADD(0x114, 0xf, 0xf);   // row_shr:4   (lanes 15 of each row: sums of 8 = after next: 16)
  SFM_DPP_ADD(0x118, 0xf, 0xf);   // row_shr:8   -> lane 15 of every row of 16 holds the row sum
  SFM_DPP_ADD(0x142, 0xa, 0xf);   // row_bcast:15 into rows 1 and 3 -> lanes 31, 63 hold sums of 32
  SFM_DPP_ADD(0x143, 0xc, 0xf);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef SFM_DPP_ADD
  (void)x;
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// N wave sums in LOCKSTEP (stage by stage over all N values): N independent DPP adds per stage instead of N dependent chains of
// six.  Same adds in the same order per value as wave_sum: bit-identical totals, left in lane 63 of every v[k].
template <int N>
__device__ __forceinline__ void wave_sums_lockstep(float (&v)[N]) {
#define SFM_DPP_STAGE(ctrl, rmask)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < N; ++i)                                                                                 \
      v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), ctrl, rmask, 0xf, true));  \
  __builtin_amdgcn_sched_barrier(0)
  SFM_DPP_STAGE(0x111, 0xf);   // row_shr:1
  SFM_DPP_STAGE(0x112, 0xf);   // row_shr:2
  SFM_DPP_STAGE(0x114, 0xf);   // row_shr:4
  SFM_DPP_STAGE(0x118, 0xf);   // row_shr:8   -> lane 15 of every row of 16 holds the row sum
  SFM_DPP_STAGE(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  SFM_DPP_STAGE(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef SFM_DPP_STAGE
}
// max over the 64 lanes of an integer, wave-uniform (DPP, the stages of wave_sum; a lane without a source lane keeps its own value)
__device__ __forceinline__ int wave_max_i(int v) {
#define SFM_DPP_MAX(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false))
  SFM_DPP_MAX(0x111, 0xf);
  SFM_DPP_MAX(0x112, 0xf);
  SFM_DPP_MAX(0x114, 0xf);
  SFM_DPP_MAX(0x118, 0xf);
  SFM_DPP_MAX(0x142, 0xa);
  SFM_DPP_MAX(0x143, 0xc);
#undef SFM_DPP_MAX
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return -wave_max_i(-v); }

__device__ __forceinline__ float lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }

__device__ __forceinline__ float uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// sign(t) with sign(0) = 0 (F.absolute backward): copysign(1, t) unless t == 0   (v_bfi + v_cmp + v_cndmask)
__device__ __forceinline__ float signf(float t) { return (t != 0.f) ? __builtin_copysignf(1.0f, t) : 0.f; }

// two horizontally adjacent taps with one 8-byte load; only 4-byte alignment is guaranteed
struct __attribute__((packed, aligned(4))) Tap2 {
  float a, b;
};
__device__ __forceinline__ Tap2 load_tap2(const float* p) { return *reinterpret_cast<const Tap2*>(p); }

// ------------------------------------------------------------------------------------------
// Per-lane accesses to a wave-uniform array: base pointer (uniform, SGPR pair) + zero-extended 32-bit BYTE offset (VGPR).
// Written this way the backend selects the `global_load/store ... v_off, s[base:base+1]` addressing form: no 64-bit per-lane
// address is ever formed (a v_lshl_add_u64 per access otherwise) and nothing 64-bit and lane-variant is hoisted out of the row
// loops to be spilled around them.  The arrays indexed like this are single planes / images: far below 4 GiB.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T ld_off(const float* base, const unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float ldf(const float* base, const unsigned idx) { return ld_off<float>(base, idx * 4u); }
__device__ __forceinline__ void stf(float* base, const unsigned idx, const float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + idx * 4u) = v;
}

// The same store written THROUGH the XCD's L2 (sc1): for outputs that nothing in this launch reads again.  A plain store leaves
// its line dirty in L2 until the release at the end of the kernel writes all of them back at once -- time the dispatch is
// charged for and the next kernel waits for (MI355X_MICROARCH.md, 'boundary': + B / 6 TB/s for B dirty bytes); written through,
// the bytes leave while the other waves still compute.
__device__ __forceinline__ void stf_wt(float* base, const unsigned idx, const float v) {
  __hip_atomic_store(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + idx * 4u), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// The per-pixel projection + sampling coordinates shared by every kernel.
//
// Restates models/transform.py:105-108 (pixel2cam), :122-131 (cam2pixel incl. the x2 rule) and
// the coordinate handling of F.spatial_transformer_sampler (:189):
//   xn = U / ((W-1)/2) - 1 ; inside iff -1 < xn < 1 (per component) ; otherwise xn *= 2 ;
//   the sampler maps back u = (xn + 1)(W-1)/2 on an image zero-padded by one pixel.
// Under the x2 rule a component that is not strictly inside (-1,1) lands at least half an image
// outside the padded picture (needs H,W >= 3), where the sampler returns exactly 0 and a zero
// gradient; so the rule collapses to the predicate `inview`, and for in-view pixels the round
// trip U -> xn -> u is the identity up to its own fp32 rounding (<= 3e-5 px at W = 416).  The
// kernels therefore test U and V directly (0 < U < W-1 is -1 < xn < 1 except within one ulp
// of the border) and sample at (U, V).
// ------------------------------------------------------------------------------------------
struct Proj {
  float U, V;    // q0/z, q1/z               (transform.py:124-125 numerators)
  float rz;      // 1/z, z = q2 + 1e-10      (transform.py:123)
  bool inview;   // 0 < U < W-1 and 0 < V < H-1   (transform.py:129)
  int u0, v0;    // top-left tap, in [0,W-2] x [0,H-2]; (0,0) when not in view
  float fu, fv;  // bilinear fractions w.r.t. (u0, v0)
};

struct ScaleConst {  // per-scale constants derived from (H, W)
  float wm1, hm1;    // W-1, H-1
};

__device__ __forceinline__ ScaleConst make_scale_const(int H, int W) {
  ScaleConst s;
  s.wm1 = (float)(W - 1);
  s.hm1 = (float)(H - 1);
  return s;
}

// from q = Pm . (c, 1)   (transform.py:122)
__device__ __forceinline__ Proj project_q(const float q0, const float q1, const float q2, const ScaleConst& sc) {
  Proj o;
  const float z = q2 + 1e-10f;
  o.rz = rcp(z);
  o.U = div_r(q0, z, o.rz);                                     // transform.py:124
  o.V = div_r(q1, z, o.rz);                                     // transform.py:125
  // 0 < U < W-1 and 0 < V < H-1 as two sign tests: U (W-1-U) > 0 holds exactly for the finite U strictly inside (the
  // difference is exact near W-1, the product cannot underflow for U > 1e-36, and NaN / infinities fail) -- two compares and one
  // scalar AND instead of four compares chained through the exec mask
  const float su = o.U * (sc.wm1 - o.U), sv = o.V * (sc.hm1 - o.V);
  o.inview = (su > 0.0f) & (sv > 0.0f);
  // in view => U, V > 0: the fraction is v_fract (= U - floor(U), the same value) and the cell index the truncating conversion
  o.fu = __builtin_amdgcn_fractf(o.U);
  o.fv = __builtin_amdgcn_fractf(o.V);
  o.u0 = o.inview ? (int)o.U : 0;     // in view => in [0, W-2]
  o.v0 = o.inview ? (int)o.V : 0;
  return o;
}

// one depth per pixel (the reference's (N,3,H*W) broadcast, base_model.py:82-84):
// aray = M . (x, y, 1) ;  q = D * aray + P[:,3]
__device__ __forceinline__ Proj project(const float aray0, const float aray1, const float aray2, const float p03,
                                        const float p13, const float p23, const float D, const ScaleConst& sc,
                                        const int H, const int W) {
  return project_q(fmaf(D, aray0, p03), fmaf(D, aray1, p13), fmaf(D, aray2, p23), sc);
}

}  // namespace sfm
