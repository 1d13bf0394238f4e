// Photometric pass with SSIM for one (wave, source): the hot loop of the fused loss.
//
// Per row r the wave runs a 3-stage software pipeline over a register ring:
//   A  warp row r      (models/transform.py:156-193): finish the bilinear taps whose gathers were
//                      issued one step earlier, then issue the gathers of row r+1 and the
//                      disparity load of row r+2, so that their latency is covered by B and C
//   B  SSIM at row r-1 (models/base_model.py:126-142) from the separable 3x3 sums
//                      (horizontal: DPP wave shifts, all fields of both channel groups in one run; vertical:
//                      ring), and the partials a,b,e ~ kappa * dS/d{mu_x, E[xx], E[xy]} (ssim_value_partials)
//   C  gradients at row r-2: transposed 3x3 pool of a,b,e  ->  dL/dI^  ->  (dL/dq0, dL/dq1) as a register
//                      pair (contract_uv)  ->  d_depth (LDS tile, summed over sources) and the nine per-lane
//                      sums behind the 12 of dL/dPm (PoseAcc)
// The ring is rotated statically (the loop body is instantiated three times), so no register
// moves are spent on it.
//
// SSIM in scaled sums (Sx = 9 mu_x etc.), algebraically models/base_model.py:130-140:
//   N1 = 2 Sx Sy + 81 c1        N2 = 18 Sxy - 2 Sx Sy + 81 c2
//   D1 = Sx^2 + Sy^2 + 81 c1    D2 = 9 (Sxx + Syy) - (Sx^2 + Sy^2) + 81 c2      SSIM = N1 N2 / (D1 D2)
#pragma once
#include "sfm_common.h"

namespace sfm {

#ifdef SFM_STAMPS   // diagnostic build only: cycle stamps around the stages of a row step (never in the product build)
#define SFM_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
struct Stamps { unsigned long long a_fin, a_iss, b, c, steps; };   // quad passes: cycles waited for start-S, start-G, finish-S, finish-G; steps = loop cycles
static __device__ Stamps g_dummy_stamps;
#define SFM_STAMPS_ARG , Stamps& st
#define SFM_STAMPS_PASS , st
#else
#define SFM_STAMP(var) do { } while (0)
#define SFM_STAMPS_ARG
#define SFM_STAMPS_PASS
#endif

struct SsimCtx {
  // uniform (SGPR)
  float M1[3], P3[3];
  // REF only (SFM_PROJECTION_REFERENCE_ORDER, see ref_position): columns 1 and 2 of batch_inv(K), rows 0..2 of K4 . T, (W-1)/2 and
  // (H-1)/2 with their reciprocals (all wave-uniform); mx[] then holds Kinv[j][0] x
  float Ki1[3], Ki2[3], Pm[12], hw[2], rhw[2];
  int x0;               // column of lane 0 (uniform); the lane's column is x0 + lane
  float k_pix;   // dL/d(sum |e|)        = gy (1-alpha) / (norm_B 3 h w)   base_model.py:111,117
  float kq;      // -dL/d(sum ssim)      = -gy alpha / (norm_B 3 h w): 2 kappa of App. A.3   base_model.py:115,117,142
  int h, w, y0, y1;
  const float* tp[3];   // target planes of this sample
  const float* sp[3];   // source planes of this (sample, source)
  const float* dp;      // disparity plane
  float* dsp;           // the three planes of this (sample, source) in the RECORD of dL/dI^ a launch with d_src bound keeps, or nullptr
  float* wp;            // planes of the optional warped-image output of this (sample, source), or nullptr   base_model.py:90-94
  const float* mp;      // explainability logits of this (sample, source) or nullptr     base_model.py:104
  float* dmp;           // their gradient plane or nullptr
  float k_exp;          // gy * exp_reg / (norm_B h w)                                  base_model.py:105,167
  size_t P;
  ScaleConst sc;
  // per lane
  float mx[3];          // M[k][0] x + M[k][2]
  float disp_first, disp_second;   // disparity of the first row a pass fetches (row y0 - halo, clamped into the image) and of the next
  unsigned xc;          // column, clamped into the image (address-safe for halo lanes)
  unsigned xc12;        // 12 xc: byte offset of the lane's texel in a pixel-interleaved row
  unsigned x12;         // 12 x as it is (NOT clamped; wraps for the lanes left of the image): lane offset of the range-checked target load
  unsigned img12;       // 12 h w (uniform): bytes of a pixel-interleaved image
  unsigned w12;         // 12 w (uniform): bytes of a pixel-interleaved row
  float w12f;           // ... as a float
  bool xin;             // column inside the image
  bool outb;            // output lane (not halo, inside the image)
  float xinf;           // ... as a factor: 1 or 0 (a multiply issues faster than a select)
  float outf;           // 1 for an output lane, else 0
  int lane;
};

// The three colour channels of a pixel: channels 0 and 1 as one 64-bit register pair, channel 2 alone.  The
// per-channel arithmetic of SSIM and of the gradient is written once, generic over the element type, and
// instantiated for the pair (v_pk_fma/mul/add_f32: two channels per issued instruction) and for the scalar.
// At the 3 waves per SIMD this kernel runs at, instruction ISSUE is the limit, and a packed op costs less than
// two scalar ones (tools/mix_pk.hip: 449 vs 524 cycles per SSIM row step per SIMD).
typedef float f2 __attribute__((ext_vector_type(2)));
struct Ch3 {
  f2 p;      // channels 0, 1
  float s;   // channel 2
};
__device__ __forceinline__ Ch3 ch3(float c0, float c1, float c2) { Ch3 r; r.p.x = c0; r.p.y = c1; r.s = c2; return r; }
__device__ __forceinline__ Ch3 ch3_zero() { return ch3(0.f, 0.f, 0.f); }

template <typename T> __device__ __forceinline__ T T_of(float v);
template <> __device__ __forceinline__ float T_of<float>(float v) { return v; }
template <> __device__ __forceinline__ f2 T_of<f2>(float v) { f2 r; r.x = v; r.y = v; return r; }
__device__ __forceinline__ f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ f2 vrcp(f2 a) { f2 r; r.x = rcp(a.x); r.y = rcp(a.y); return r; }
__device__ __forceinline__ float vrcp(float a) { return rcp(a); }
__device__ __forceinline__ f2 hsum3(f2 v) { f2 r; r.x = hsum3(v.x); r.y = hsum3(v.y); return r; }
// a + (its left neighbour), b + (the right neighbour of a): per component, so that each is ONE v_add_f32_dpp
__device__ __forceinline__ float add_left(float a) { return a + from_left(a); }
__device__ __forceinline__ float add_right(float b, float a) { return b + from_right(a); }
__device__ __forceinline__ f2 add_left(f2 a) { f2 r; r.x = add_left(a.x); r.y = add_left(a.y); return r; }
__device__ __forceinline__ f2 add_right(f2 b, f2 a) { f2 r; r.x = add_right(b.x, a.x); r.y = add_right(b.y, a.y); return r; }
__device__ __forceinline__ void pin(float& x) { asm("" : "+v"(x)); }   // see hsum3(float)
__device__ __forceinline__ void pin(f2& v) { asm("" : "+v"(v)); }
// Horizontal 3-sums of several fields at once, the DPP adds of the fields interleaved: a DPP operand written by the
// instruction just before costs an s_nop (2 wait states), and left to itself the scheduler puts each field's two DPP
// adds right behind its producer (tools/nop_cost.hip: 3.15 -> 2.83 cycles per instruction at 3 waves per SIMD).
template <typename T>
__device__ __forceinline__ void hsum3_group(T& a, T& b, T& c, T& d) {
  pin(a); pin(b); pin(c); pin(d);
  __builtin_amdgcn_sched_barrier(0);
  const T la = add_left(a), lb = add_left(b), lc = add_left(c), ld = add_left(d);
  a = add_right(la, a); b = add_right(lb, b); c = add_right(lc, c); d = add_right(ld, d);
  __builtin_amdgcn_sched_barrier(0);
}
template <typename T>
__device__ __forceinline__ void hsum3_group(T& a, T& b, T& c) {
  pin(a); pin(b); pin(c);
  __builtin_amdgcn_sched_barrier(0);
  const T la = add_left(a), lb = add_left(b), lc = add_left(c);
  a = add_right(la, a); b = add_right(lb, b); c = add_right(lc, c);
  __builtin_amdgcn_sched_barrier(0);
}
// the same for the fields of both channel groups in one run (the scalar group's DPP adds first: its producers are the older ones)
__device__ __forceinline__ void hsum3_group(f2& a, f2& b, f2& c, f2& d, float& e, float& f, float& g, float& h) {
  pin(a); pin(b); pin(c); pin(d); pin(e); pin(f); pin(g); pin(h);
  __builtin_amdgcn_sched_barrier(0);
  const float le = add_left(e), lf = add_left(f), lg = add_left(g), lh = add_left(h);
  const f2 la = add_left(a), lb = add_left(b), lc = add_left(c), ld = add_left(d);
  e = add_right(le, e); f = add_right(lf, f); g = add_right(lg, g); h = add_right(lh, h);
  a = add_right(la, a); b = add_right(lb, b); c = add_right(lc, c); d = add_right(ld, d);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void hsum3_group(f2& a, f2& b, f2& c, float& e, float& f, float& g) {
  pin(a); pin(b); pin(c); pin(e); pin(f); pin(g);
  __builtin_amdgcn_sched_barrier(0);
  const float le = add_left(e), lf = add_left(f), lg = add_left(g);
  const f2 la = add_left(a), lb = add_left(b), lc = add_left(c);
  e = add_right(le, e); f = add_right(lf, f); g = add_right(lg, g);
  a = add_right(la, a); b = add_right(lb, b); c = add_right(lc, c);
  __builtin_amdgcn_sched_barrier(0);
}
// clip((1 - S) / 2, 0, 1) as one multiply-add with the clamp output modifier (the compiler folds the scalar form by itself;
// for the packed pair it splits the clamp off into two v_max, so that one is written out)
__device__ __forceinline__ float half_one_minus_clamped(float S) { return fminf(fmaxf(fmaf(S, -0.5f, 0.5f), 0.f), 1.f); }
__device__ __forceinline__ f2 half_one_minus_clamped(f2 S) {
  f2 r;
  asm("v_pk_fma_f32 %0, %1, 0.5, 0.5 op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(r) : "v"(S));
  return r;
}
// sign(d) with sign(0) = 0 (F.absolute backward) as the difference of two clamped products: clamp(d * 2^127, 0, 1) is 1 for every
// normal d > 0 and 0 for d <= 0 -- the clamp rides on the multiply as its output modifier, and these multiplies and the
// subtraction issue at the full rate, where the compare + select + sign-bit form needs three slow-class instructions per value
// (profiles/r03_op_cost_microbench.txt).  (A difference of two image values is either 0 or far above 2^-126: for denormal d the
// result would be a fraction of 1.)
__device__ __forceinline__ float sign01(float d) {
  const float big = 0x1p127f;
  float sp, sm;
  asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(sp) : "v"(d), "s"(big));
  asm("v_mul_f32_e64 %0, -%1, %2 clamp" : "=v"(sm) : "v"(d), "s"(big));
  return sp - sm;
}
__device__ __forceinline__ f2 sign01(f2 d) {
  f2 big;   // (a packed operand is a 64-bit register pair: the constant sits in one scalar pair, both halves)
  big.x = 0x1p127f; big.y = 0x1p127f;
  f2 sp, sm;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(sp) : "v"(d), "s"(big));
  asm("v_pk_mul_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0] clamp" : "=v"(sm) : "v"(d), "s"(big));
  return sp - sm;
}
// k * sign(d)   (k carries the sign of the upstream gradient)
__device__ __forceinline__ float ksign(float k, float d) { return k * sign01(d); }
__device__ __forceinline__ f2 ksign(float k, f2 d) { return sign01(d) * k; }
// g + k * sign(d)
__device__ __forceinline__ float add_ksign(float g, float k, float d) { return fmaf(sign01(d), k, g); }
__device__ __forceinline__ f2 add_ksign(f2 g, float k, f2 d) { return vfma(sign01(d), T_of<f2>(k), g); }
__device__ __forceinline__ float vabs_sum(f2 d) { return fabsf(d.x) + fabsf(d.y); }
__device__ __forceinline__ float vabs_sum(float d) { return fabsf(d); }
__device__ __forceinline__ float vhadd(f2 v) { return v.x + v.y; }
__device__ __forceinline__ float vhadd(float v) { return v; }

// ------------------------------------------------------------------------------------------
// The geometry of a wave's passes -- proj_tgt_to_src (models/transform.py:64-91: euler2mat :11-40, pose_vec2mat :43-59, K4 . T
// :86-88) and batch_inv(K) (:105) for every source of its (sample, scale) -- built by the wave itself, ONCE, at its start: no
// geometry kernel in front of the launch, no table.  A launch that waits for another launch costs ~2.5 us of a 60 us step
// (profiles/r04_ab_geom_in_wave.txt); the same numbers made redundantly by every wave cost ~100 vector instructions of the
// ~17000 of a wave, because the work is laid out over LANES where it is regular:
//   * lanes 8g .. 8g+7 belong to source g; lane 8g+k (k = 0,1,2) holds angle k of its pose and ROW k of K, lanes 8g+3..5 the
//     translation: ONE evaluation of the short-range sincos (angles are clipped to +-pi, :23) gives all angles of all sources, and
//     the products K . R, K . t and (K R) . K^-1 are 9 + 3 + 9 multiply-adds for all rows of all sources at once;
//   * R = (X . Y) . Z in closed form -- products with the zeros and ones of X, Y, Z are exact in the general product too, so
//     these are its values up to the fused roundings (14 instructions instead of 90);
//   * K^-1 = adj(K) / det with ONE refined reciprocal instead of nine IEEE divisions.
// The rows stay in their lanes (four registers for the life of the wave); a pass takes the twelve numbers its row loop needs
// out of them with v_readlane.  (K^-1 is NOT needed again: the pose sums leave the wave un-multiplied, see pose_sums_raw.)
// ------------------------------------------------------------------------------------------
// three consecutive floats with one 12-byte load; only 4-byte alignment is guaranteed
struct __attribute__((packed, aligned(4))) Rgb {
  float c[3];
};
__device__ __forceinline__ Rgb load_rgb(const float* p) { return *reinterpret_cast<const Rgb*>(p); }
struct __attribute__((packed, aligned(4))) K9 {
  float k[9];
};
__device__ __forceinline__ K9 load_k9(const float* Kp) { return *reinterpret_cast<const K9*>(Kp); }

__device__ __forceinline__ float from_lane(const float v, const int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
// the value of lane `src` (a per-lane index; ds_bpermute: the LDS crossbar, no LDS memory, no vector-ALU slot)
__device__ __forceinline__ float from_lane_v(const float v, const int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}

// batch_inv for one general 3x3 (models/transform.py:105), wave-uniform: cofactors as multiply + fused multiply-add, one reciprocal
__device__ __forceinline__ void inv3_fast(const float* K, float* o) {
  const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  const float A = fmaf(e, i, -(f * h)), B = fmaf(f, g, -(d * i)), Cc = fmaf(d, h, -(e * g));
  const float det = fmaf(a, A, fmaf(b, B, c * Cc));
  const float r = rcp_refined(det);
  o[0] = A * r;
  o[1] = fmaf(c, h, -(b * i)) * r;
  o[2] = fmaf(b, f, -(c * e)) * r;
  o[3] = B * r;
  o[4] = fmaf(a, i, -(c * g)) * r;
  o[5] = fmaf(c, d, -(a * f)) * r;
  o[6] = Cc * r;
  o[7] = fmaf(b, g, -(a * h)) * r;
  o[8] = fmaf(a, e, -(b * d)) * r;
}

// batch_inv for one general 3x3 in the REFERENCE's roundings (models/transform.py:105; oracle: batch_inv3): cofactors and determinant
// as separate multiplies and adds (nothing fused), every element the correctly rounded quotient adj / det -- v_rcp + one Newton
// step for 1 / det, then one residual correction per element (the IEEE result except in rare double-rounding cases).
__device__ __forceinline__ void inv3_exact(const float* K, float* o) {
  float adj[9], det;
  {
#pragma clang fp contract(off)
    const float a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const float A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    det = (a * A + b * B) + c * Cc;
    adj[0] = A; adj[1] = -(b * i - c * h); adj[2] = b * f - c * e;
    adj[3] = B; adj[4] = a * i - c * g; adj[5] = -(a * f - c * d);
    adj[6] = Cc; adj[7] = -(a * h - b * g); adj[8] = a * e - b * d;
  }
  const float r = rcp_refined(det);
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = div_r(adj[k], det, r);
}
// a0 b0 + a1 b1 + a2 b2, left to right, nothing fused: one element of F.batch_matmul as the oracle evaluates it (_bmm)
__device__ __forceinline__ float dot3_unfused(const float a0, const float b0, const float a1, const float b1, const float a2, const float b2) {
#pragma clang fp contract(off)
  return (a0 * b0 + a1 * b1) + a2 * b2;
}

struct WaveGeom {       // lane 8g+k: row k of source g (per-lane registers, alive for the whole wave)
  float M0, M1, M2;     // M = (K R) K^-1:  q = D (M . (x,y,1)) + P3
  float P3;             // K . t
};
// with REF (SfmLossDesc.projection = SFM_PROJECTION_REFERENCE_ORDER) the passes evaluate the reference's own chain per pixel: they
// take the rows of Pm = K4 . T and of K^-1 themselves
struct WaveGeomRef {
  float P0, P1, P2, P3; // row k of Pm[:3, :]                      (transform.py:86-88)
  float Ki0, Ki1, Ki2;  // row k of batch_inv(K)                   (transform.py:105)
};

// pose[g]: pointer to the (B,6) pose array of source g (kernel arguments: wave-uniform, NULL beyond n_src)
// Since round 6 the products are the REFERENCE's (round-5 verdict, item 1a): R = (X . Y) . Z, K . [R | t] and P[:, :3] . K^-1 as
// separate multiplies and adds in the oracle's order (_bmm: left to right, nothing fused), K^-1 as correctly rounded quotients
// (inv3_exact) -- about 50 more vector instructions per WAVE than the fused form of rounds 4-5 (of ~8 500), once, in the shadow of the
// first loads.  With the fused roundings the projection rows differed from the reference's in the last bits, which at U ~ 800 is
// the difference between 20 and 355 warped pixels beyond 1e-4 (profiles/r05_reference_order_variants.txt, variant 1).
template <bool REFG>
__device__ __forceinline__ void build_wave_geom_any(const float* const (&pose)[SFM_MAX_SRC], const int n_src, const int b,
                                                    const float* Kp /* K of this (sample, scale) */, const int lane, WaveGeom& g,
                                                    WaveGeomRef& gr) {
  const float pi = 3.14159265358979323846f;
  const int grp = lane >> 3, sub = lane & 7, base = lane & ~7;
  // One batch of loads: the lane's pose component (sub 0..2 the angles, 3..5 the translation) of ITS source, its row of K, all of K.
  // The lane's pose pointer is SELECTED from the eight kernel arguments (constant indices: scalar registers), never fetched through
  // a run-time index and never under a branch: indexed, the compiler fetched the pointer itself with a vector load from the argument
  // block and waited for it before it could issue the pose load -- two dependent round trips PER SOURCE at the start of every wave,
  // each draining the disparity loads issued before (found in the ISA of the first version of this function).  Lanes of groups
  // beyond n_src take the last source's pointer (the arguments beyond it are NULL); nothing reads their results.
  // (K first: its addresses need nothing but the lane; the fence below keeps all the loads of this function in front of the
  //  arithmetic -- left to itself the scheduler sank the K loads behind the wait for the pose: a second round trip)
  const Rgb Kr = ld_off<Rgb>(Kp, 12u * (unsigned)min(sub, 2));
  const K9 Ku = load_k9(Kp);
  const int gsel = min(grp, n_src - 1);
  const float* pp = nullptr;
#pragma unroll
  for (int k = 0; k < SFM_MAX_SRC; ++k) {
    const float* pk = pose[k];
    asm volatile("" : "+s"(pk));     // (keeps the selection from being folded back into an indexed load of the pointer: it was)
    pp = (gsel == k) ? pk : pp;
  }
  const float pv = ldf(pp + b * 6, (unsigned)min(sub, 5));
  __builtin_amdgcn_sched_barrier(0);
  float sn, cs;
  sincos_pi(fminf(fmaxf(pv, -pi), pi), &sn, &cs);                     // transform.py:23-25
  const float sx = from_lane_v(sn, base), sy = from_lane_v(sn, base + 1), sz = from_lane_v(sn, base + 2);
  const float cx = from_lane_v(cs, base), cy = from_lane_v(cs, base + 1), cz = from_lane_v(cs, base + 2);
  const float tx = from_lane_v(pv, base + 3), ty = from_lane_v(pv, base + 4), tz = from_lane_v(pv, base + 5);
  // X . Y = [[cy, 0, sy], [sx sy, cx, -sx cy], [-cx sy, sx, cx cy]] ;  R = (X . Y) . Z   (transform.py:27-39).  The products with
  // the zeros and ones of X, Y, Z are exact and adding their zeros changes nothing, so these closed forms ARE the general products'
  // values when the remaining two-term sums are evaluated unfused, as here.
  float R[9];
  {
#pragma clang fp contract(off)
    const float xy10 = sx * sy, xy20 = -(cx * sy), xy12 = -(sx * cy), xy22 = cx * cy;
    R[0] = cy * cz;               R[1] = cy * (-sz);             R[2] = sy;
    R[3] = xy10 * cz + cx * sz;   R[4] = xy10 * (-sz) + cx * cz; R[5] = xy12;
    R[6] = xy20 * cz + sx * sz;   R[7] = xy20 * (-sz) + sx * cz; R[8] = xy22;
  }
  float Kinv[9];
  inv3_exact(Ku.k, Kinv);
  // lane 8g+k: row k of P = K . [R | t] (transform.py:56-58,86-88; the fourth term of K4 . T is a product with 0 or the lone 1 . t)
  float P[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) P[j] = dot3_unfused(Kr.c[0], R[j], Kr.c[1], R[3 + j], Kr.c[2], R[6 + j]);
  const float P3 = dot3_unfused(Kr.c[0], tx, Kr.c[1], ty, Kr.c[2], tz);
  if constexpr (REFG) {
    gr.P0 = P[0]; gr.P1 = P[1]; gr.P2 = P[2]; gr.P3 = P3;
    const int k = min(sub, 2);
    gr.Ki0 = (k == 0) ? Kinv[0] : (k == 1) ? Kinv[3] : Kinv[6];
    gr.Ki1 = (k == 0) ? Kinv[1] : (k == 1) ? Kinv[4] : Kinv[7];
    gr.Ki2 = (k == 0) ? Kinv[2] : (k == 1) ? Kinv[5] : Kinv[8];
  } else {
    // ... and of M = P[:, :3] . K^-1
    g.P3 = P3;
    g.M0 = dot3_unfused(P[0], Kinv[0], P[1], Kinv[3], P[2], Kinv[6]);
    g.M1 = dot3_unfused(P[0], Kinv[1], P[1], Kinv[4], P[2], Kinv[7]);
    g.M2 = dot3_unfused(P[0], Kinv[2], P[1], Kinv[5], P[2], Kinv[8]);
  }
}
__device__ __forceinline__ WaveGeom build_wave_geom(const float* const (&pose)[SFM_MAX_SRC], const int n_src, const int b,
                                                    const float* Kp, const int lane) {
  WaveGeom g;
  WaveGeomRef unused;
  build_wave_geom_any<false>(pose, n_src, b, Kp, lane, g, unused);
  return g;
}
__device__ __forceinline__ WaveGeomRef build_wave_geom_ref(const float* const (&pose)[SFM_MAX_SRC], const int n_src, const int b,
                                                           const float* Kp, const int lane) {
  WaveGeom unused;
  WaveGeomRef g;
  build_wave_geom_any<true>(pose, n_src, b, Kp, lane, unused, g);
  return g;
}

struct RowS {            // a warped row as the later stages need it (per lane = per pixel)
  Ch3 ih, it;            // I^ (0 where not in view / outside the image), I (0 outside the image)
  f2 duv0, duv1, duv_s;  // (dI^/du, dI^/dv) of channels 0, 1, 2 (of whatever the tap registers held where the sample is not in view: rzi
                         // masks them): one register pair per channel, so that the contraction to dL/d(u,v) is three packed
                         // multiply-adds with the channel's dL/dI^ broadcast -- no horizontal adds, no register shuffles
  f2 UV;                 // q0/z, q1/z: one register pair, so that everything per-coordinate is one packed instruction
  float D;               // depth
  float rzi;             // 1/z where the sample is in view AND the lane is an output lane, else 0: the factor that takes
                         // dL/d(u,v) to dL/d(q0,q1), and the only mask the gradient needs downstream
  float nm;              // 1 - mask, mask = all three channels of I^ exactly 0   base_model.py:96
};
struct RowG { Ch3 a, b, e; };                             // horizontal 3-sums of the SSIM partials

struct Pipe {            // a row whose gathers are in flight
  float ta[3], tb[3];    // taps (u0, v0), (u0+1, v0) of the three channels
  float ba[3], bb[3];    // taps (u0, v0+1), (u0+1, v0+1)
  float it[3];
  f2 UV, f;              // sampling position (q0/z, q1/z) and its bilinear fractions
  float rz, D;
  float lg;              // explainability logit (only loaded when C.mp != nullptr)
  bool inview;           // the sample is taken (in view, column inside the image)
  bool inview_o;         // ... and the lane is an output lane: only there do dI^/du, dI^/dv exist (a halo lane has no gradient of its own)
};

__device__ __forceinline__ void zero(RowS& s) {
  s.ih = s.it = ch3_zero(); s.duv0 = s.duv1 = s.duv_s = T_of<f2>(0.f);
  s.UV = T_of<f2>(0.f); s.D = s.rzi = s.nm = 0.f;
}
__device__ __forceinline__ void zero(RowG& s) { s.a = s.b = s.e = ch3_zero(); }
// The same zeros, but produced by instructions the compiler must leave where they are written.  The ring slot of a row
// outside the image is zero; written as plain assignments, LLVM materialises the 16 zeros in FRONT of the (wave-uniform)
// branch that separates such rows from ordinary ones, i.e. 16 v_mov per row step of every row (5 % of the vector
// instructions of a step), for a case that only the top and bottom chunks of an image ever see.
__device__ __forceinline__ float opaque_zero() {
  float z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}
__device__ __forceinline__ Ch3 ch3_opaque_zero() {
  const float a = opaque_zero(), b = opaque_zero(), c = opaque_zero();
  return ch3(a, b, c);
}
__device__ __forceinline__ void zero_rare(RowS& s) {
  s.ih = ch3_opaque_zero(); s.it = ch3_opaque_zero();
  s.duv0.x = opaque_zero(); s.duv0.y = opaque_zero(); s.duv1.x = opaque_zero(); s.duv1.y = opaque_zero(); s.duv_s.x = opaque_zero(); s.duv_s.y = opaque_zero();
  s.UV.x = opaque_zero(); s.UV.y = opaque_zero(); s.D = opaque_zero(); s.rzi = opaque_zero(); s.nm = opaque_zero();
}

// stage A, first half: project row r and issue its loads (row r is inside the image)
struct __attribute__((packed, aligned(4))) Rgb2 {   // two horizontally adjacent pixel-interleaved texels: one 24-byte access
  float c[6];
};

// The sampling position of one pixel in the REFERENCE's own evaluation order (SfmLossDesc.projection =
// SFM_PROJECTION_REFERENCE_ORDER), statement for statement what the oracle -- and the stand-alone warp operator's ref_project,
// sfm_ops.hip -- evaluate, nothing fused (models/transform.py:105-108, 122-131, and the sampler's own position, :189):
//   ray = K^-1 . (x, y, 1) ;  c = D ray ;  q = Pm . (c, 1) ;  z = q2 + 1e-10 ;  U = q0 / z ;  xn = U / ((W-1)/2.) - 1 ;
//   in view iff -1 < xn < 1 and -1 < yn < 1 ;  the sampler samples at (xn + 1) (W-1) / 2 on the image padded by one pixel.
// Every DIVISION (1 / disp, q / z, U / half) is v_rcp + residual correction(s): the correctly rounded quotient except in rare
// double-rounding cases, at 3-4 instructions instead of the ~10 of the IEEE sequence; the x and y components of every step share
// one packed instruction.  45 vector instructions per row step against the 16 of the product's chain (FAST).
struct RefPos {
  f2 UV;        // (q0, q1) / z: what the backward's dL/dq2 = -(gU U + gV V) / z takes
  f2 fr, cell;  // bilinear fractions and top-left tap (as floats, un-padded) of the SAMPLER's position
  float rz, D;
  bool inview;  // transform.py:129, on xn and yn
};
// ... as the integer cell the planar gather and the dL/d(src) scatter take.  In view the sampler's position lies in (1, W] -- W itself
// when (xn + 1) rounds up to 2 -- so its cell can be the LAST column / row with a zero fraction: folded onto the cell before it with
// fraction 1 (the same value; no tap beyond the image is ever addressed).
__device__ __forceinline__ Proj ref_proj_cell(const RefPos& rp, const int h, const int w) {
  Proj p;
  p.U = rp.UV.x; p.V = rp.UV.y; p.rz = rp.rz; p.inview = rp.inview;
  int u0 = (int)rp.cell.x, v0 = (int)rp.cell.y;
  p.fu = rp.fr.x; p.fv = rp.fr.y;
  if (u0 > w - 2) { u0 = w - 2; p.fu = 1.f; }
  if (v0 > h - 2) { v0 = h - 2; p.fv = 1.f; }
  p.u0 = rp.inview ? u0 : 0;
  p.v0 = rp.inview ? v0 : 0;
  return p;
}
__device__ __forceinline__ RefPos ref_position(const SsimCtx& C, const float yf, const float D /* 1 / disp, correctly rounded */) {
  RefPos o;
  o.D = D;
  f2 ray, q, kxp, Ki1p, Ki2p, Pc0, Pc1, Pc2, Pc3, hwp, rhwp, whm1;
  kxp.x = C.mx[0]; kxp.y = C.mx[1]; Ki1p.x = C.Ki1[0]; Ki1p.y = C.Ki1[1]; Ki2p.x = C.Ki2[0]; Ki2p.y = C.Ki2[1];
  Pc0.x = C.Pm[0]; Pc0.y = C.Pm[4]; Pc1.x = C.Pm[1]; Pc1.y = C.Pm[5]; Pc2.x = C.Pm[2]; Pc2.y = C.Pm[6]; Pc3.x = C.Pm[3]; Pc3.y = C.Pm[7];
  hwp.x = C.hw[0]; hwp.y = C.hw[1]; rhwp.x = C.rhw[0]; rhwp.y = C.rhw[1]; whm1.x = C.sc.wm1; whm1.y = C.sc.hm1;
  float q2;
  {
#pragma clang fp contract(off)
    ray = (kxp + Ki1p * yf) + Ki2p;                                   // transform.py:105-106 (the third pixel coordinate is 1)
    const float ray2 = (C.mx[2] + C.Ki1[2] * yf) + C.Ki2[2];
    const f2 c = ray * o.D;                                           // :107
    const float c2 = ray2 * o.D;
    q = ((Pc0 * c.x + Pc1 * c.y) + Pc2 * c2) + Pc3;                   // :122 (the fourth camera coordinate is 1, :108)
    q2 = ((C.Pm[8] * c.x + C.Pm[9] * c.y) + C.Pm[10] * c2) + C.Pm[11];
  }
  const float z = q2 + 1e-10f;                                        // :123
  o.rz = rcp(z);
  const f2 qr = q * o.rz;
  o.UV = vfma(vfma(-qr, T_of<f2>(z), q), T_of<f2>(o.rz), qr);         // :124-125, the quotients
  const f2 xr = o.UV * rhwp;
  const f2 xn = vfma(vfma(-xr, hwp, o.UV), rhwp, xr) - 1.0f;          // ... / ((W-1)/2.) - 1
  // -1 < xn < 1 exactly when 1 - xn^2 > 0 (one rounding of the exact value: the sign is right to the last bit; a NaN fails)
  const f2 t = vfma(-xn, xn, T_of<f2>(1.0f));
  o.inview = (t.x > 0.0f) & (t.y > 0.0f);                             // :129
  f2 up;
  {
#pragma clang fp contract(off)
    up = ((xn + 1.0f) * whm1) * 0.5f + 1.0f;                          // the sampler's position on the zero-padded image: in (1, W) in view
  }
  o.fr.x = __builtin_amdgcn_fractf(up.x); o.fr.y = __builtin_amdgcn_fractf(up.y);
  o.cell = (up - o.fr) - 1.0f;                                        // floor(up) - 1: the top-left tap, un-padded
  return o;
}

// HWC: the images are pixel-interleaved (SFM_LAYOUT_HWC): C.tp[0] / C.sp[0] are the (h,w,3) images of this sample /
// (sample, source), and the three channels of a tap come with one load.
// REF: 0 the product's chain (SFM_PROJECTION_FAST): q = D (M . pix) + P3, the in-view test on U, V directly, the sample at (U, V);
//      1 the reference's evaluation order per pixel (SFM_PROJECTION_REFERENCE_ORDER, ref_position above).
// SECOND (ssim_pair_pass, sfm_ssim_pair.h): the second source of a pass that handles two sources per row -- the depth (ps.D, set by
//      the caller) and the target texel were fetched with the first one and are not fetched again.
template <bool HWC, int REF = 0, bool SECOND = false>
__device__ __forceinline__ void issue_row(const SsimCtx& C, const int r, const float disp, Pipe& ps) {
  static_assert(!SECOND || (HWC && REF == 0), "two sources per pass: pixel-interleaved layout, the product's projection");
  const float yf = (float)r;
  Proj p;
  f2 UV, fr, cell;
  float rz;
  if constexpr (REF != 0) {
    const RefPos rp = ref_position(C, yf, rcp_refined(disp));         // base_model.py:60: depth = 1 / disp
    ps.D = rp.D; UV = rp.UV; fr = rp.fr; cell = rp.cell; rz = rp.rz; p.inview = rp.inview;
    p.u0 = p.v0 = 0;
    if constexpr (!HWC) {
      p = ref_proj_cell(rp, C.h, C.w);
      fr.x = p.fu; fr.y = p.fv;
    }
  } else if constexpr (HWC) {
    // depth = 1 / disp (base_model.py:60) from v_rcp_f32 alone (1 ulp): the quotients U = q0/z, V = q1/z below keep their residual
    // correction (they decide the strict in-view test), the depth does not need one -- a last-bit change of D moves the sample by
    // 1e-7 of its parallax
    if constexpr (!SECOND) ps.D = rcp(disp);
    // The projection of sfm_common.h's project(), with the x and y components of every step in ONE packed instruction (the same
    // IEEE operations per component: bit-identical values, two thirds of the instructions):
    //   a = M (x,y,1) ; q = D a + P[:,3] ; z = q2 + 1e-10 ; (U,V) = (q0,q1) / z ; in view iff U (W-1-U) > 0 and V (H-1-V) > 0
    f2 M1p, mxp, P3p, whm1;
    M1p.x = C.M1[0]; M1p.y = C.M1[1]; mxp.x = C.mx[0]; mxp.y = C.mx[1]; P3p.x = C.P3[0]; P3p.y = C.P3[1];
    whm1.x = C.sc.wm1; whm1.y = C.sc.hm1;
    const f2 a = vfma(M1p, T_of<f2>(yf), mxp);
    const float a2 = fmaf(C.M1[2], yf, C.mx[2]);
    const f2 q = vfma(T_of<f2>(ps.D), a, P3p);
    const float z = fmaf(ps.D, a2, C.P3[2]) + 1e-10f;                 // transform.py:123
    rz = rcp(z);
    const f2 qr = q * rz;                                             // div_r of both quotients (transform.py:124-125)
    UV = vfma(vfma(-qr, T_of<f2>(z), q), T_of<f2>(rz), qr);
    const f2 sg = UV * (whm1 - UV);
    p.inview = (sg.x > 0.0f) & (sg.y > 0.0f);
    fr.x = __builtin_amdgcn_fractf(UV.x); fr.y = __builtin_amdgcn_fractf(UV.y);
    cell = UV - fr;
    p.u0 = p.v0 = 0;   // (the pixel-interleaved gather forms its offset from the cell = UV - fr)
  } else {
    // (the planar gather needs the integer cell as well, and measured 4 % slower with the packed chain in front of it)
    ps.D = rcp(disp);
    const float a0 = fmaf(C.M1[0], yf, C.mx[0]), a1 = fmaf(C.M1[1], yf, C.mx[1]), a2 = fmaf(C.M1[2], yf, C.mx[2]);
    p = project(a0, a1, a2, C.P3[0], C.P3[1], C.P3[2], ps.D, C.sc, C.h, C.w);
    UV.x = p.U; UV.y = p.V; fr.x = p.fu; fr.y = p.fv; rz = p.rz;
    cell = UV;   // (unused: the planar gather takes the integer cell of project())
  }
  ps.UV = UV; ps.f = fr;
  if constexpr (HWC) {
    // (nothing of the masks crosses the step: out-of-view taps arrive as zeros, and 1/z is masked here)
    ps.rz = (p.inview && C.outb) ? rz : 0.f;
    ps.inview = ps.inview_o = true;
  } else {
    ps.rz = rz;
    ps.inview = p.inview && C.xin;
    ps.inview_o = p.inview && C.outb;
  }
  const unsigned off = (unsigned)(p.v0 * C.w + p.u0);
  const unsigned offt = (unsigned)r * (unsigned)C.w + C.xc;
  if constexpr (HWC) {
    // Byte offset of the top-left tap, 12 (v0 w + u0), formed in FLOAT from the integer parts U - fu, V - fv (exact: an image has
    // fewer than 2^24 bytes, make_plan checks) and converted once: two full-rate subtractions, a multiply and a multiply-add
    // instead of two conversions, two selects and a 64-bit multiply-add.  The two horizontally adjacent taps of a row are ONE
    // 24-byte access (dwordx4 + dwordx2: four separate 12-byte loads cost the gather path 40 % more time per step at 256x832), the
    // row below is the same address + 12 w.
    const float bof = fmaf(cell.y, C.w12f, cell.x * 12.f);
    // Range-checked buffer loads (MUBUF, raw descriptor): a lane whose offset is not below num_records gets ZEROS.
    //  * source taps: num_records = the image; a lane that is not in view is given an offset outside it, so all its taps are 0 and
    //    with them the value and both derivatives -- no select per channel in finish_row (base_model.py:96's mask wants exactly 0);
    //    the row below is the same lane offset against a descriptor that starts one row further down (the instruction's scalar
    //    offset would do the same, but it takes part in the range check)
    //  * target texel: the descriptor is ONE row (base = that row, num_records = its bytes: scalar arithmetic per step); the lane
    //    offset is 12 x, which is out of range exactly for the columns outside the image (x < 0 wraps) -- no clamped column, no
    //    multiply by the column mask
    const char* sp8 = reinterpret_cast<const char*>(C.sp[0]);
    const char* tp8 = reinterpret_cast<const char*>(C.tp[0]) + (size_t)((unsigned)r * C.w12);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sp8), 0, (int)C.img12, 0x00027000);
    const __amdgpu_buffer_rsrc_t srs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sp8 + C.w12), 0, (int)(C.img12 - C.w12), 0x00027000);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tp8), 0, (int)C.w12, 0x00027000);
    const unsigned o12 = (p.inview && C.xin) ? (unsigned)bof : 0x80000000u;   // (a column outside the image is not a pixel: its I^ is 0)
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    typedef unsigned u3v __attribute__((ext_vector_type(3)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u4v t0 = __builtin_amdgcn_raw_buffer_load_b128(srs, o12, 0, 0);
    const u2v t1 = __builtin_amdgcn_raw_buffer_load_b64(srs, o12 + 16u, 0, 0);
    const u4v b0 = __builtin_amdgcn_raw_buffer_load_b128(srs2, o12, 0, 0);
    const u2v b1 = __builtin_amdgcn_raw_buffer_load_b64(srs2, o12 + 16u, 0, 0);
    u3v i0 = {0u, 0u, 0u};
    if constexpr (!SECOND) i0 = __builtin_amdgcn_raw_buffer_load_b96(trs, C.x12, 0, 0);
    Rgb2 T, Bt;
    Rgb I;
    T.c[0] = __uint_as_float(t0.x); T.c[1] = __uint_as_float(t0.y); T.c[2] = __uint_as_float(t0.z); T.c[3] = __uint_as_float(t0.w);
    T.c[4] = __uint_as_float(t1.x); T.c[5] = __uint_as_float(t1.y);
    Bt.c[0] = __uint_as_float(b0.x); Bt.c[1] = __uint_as_float(b0.y); Bt.c[2] = __uint_as_float(b0.z); Bt.c[3] = __uint_as_float(b0.w);
    Bt.c[4] = __uint_as_float(b1.x); Bt.c[5] = __uint_as_float(b1.y);
    I.c[0] = __uint_as_float(i0.x); I.c[1] = __uint_as_float(i0.y); I.c[2] = __uint_as_float(i0.z);
    (void)off; (void)offt;
#pragma unroll
    for (int c = 0; c < 3; ++c) { ps.ta[c] = T.c[c]; ps.tb[c] = T.c[3 + c]; ps.ba[c] = Bt.c[c]; ps.bb[c] = Bt.c[3 + c]; ps.it[c] = I.c[c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const Tap2 t = ld_off<Tap2>(C.sp[c], 4u * off), b = ld_off<Tap2>(C.sp[c], 4u * (off + (unsigned)C.w));
      ps.ta[c] = t.a; ps.tb[c] = t.b; ps.ba[c] = b.a; ps.bb[c] = b.b;
      ps.it[c] = ldf(C.tp[c], offt);
    }
  }
  if constexpr (!SECOND) {
    if (C.mp != nullptr) ps.lg = ldf(C.mp, offt);
  }
}

// stage A, second half: bilinear value and derivatives from the gathered taps
// (PREMASKED: the taps of a sample that is not in view and the target texel of a column outside the image arrived as zeros, and
//  ps.rz is already masked: the pixel-interleaved path's range-checked loads, see issue_row)
template <bool PREMASKED>
__device__ __forceinline__ void finish_row(const SsimCtx& C, const Pipe& ps, RowS& s) {
  float ih[3], it[3], du[3], dv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dxt = ps.tb[c] - ps.ta[c], dxb = ps.bb[c] - ps.ba[c];
    const float top = fmaf(ps.f.x, dxt, ps.ta[c]);
    const float bot = fmaf(ps.f.x, dxb, ps.ba[c]);
    const float dvv = bot - top;
    const float val = (PREMASKED || ps.inview) ? fmaf(ps.f.y, dvv, top) : 0.f;
    ih[c] = val;
    dv[c] = dvv;
    du[c] = fmaf(ps.f.y, dxb - dxt, dxt);
    it[c] = PREMASKED ? ps.it[c] : ps.it[c] * C.xinf;       // 0 outside the image (planar: the load came from the clamped column)
  }
  s.ih = ch3(ih[0], ih[1], ih[2]);
  s.it = ch3(it[0], it[1], it[2]);
  s.duv0.x = du[0]; s.duv0.y = dv[0]; s.duv1.x = du[1]; s.duv1.y = dv[1];
  s.duv_s.x = du[2]; s.duv_s.y = dv[2];
  s.UV = ps.UV; s.D = ps.D;
  s.rzi = (PREMASKED || ps.inview_o) ? ps.rz : 0.f;   // halo lanes get no gradient of their own
  // base_model.py:96: mask = all three channels exactly 0 (+-0 both count).  The largest magnitude among the channels (ONE
  // v_max3_f32 with |.| operand modifiers) is 0 or a normal number, for images of ANY range: the clamped product with 2^127 is the
  // 0 / 1 indicator.  (Rounds 1-3 OR-ed the channels' bit patterns instead: exact for the reference's [-1, 1] images, where no
  // exponent field can fill up, but a 200.0 next to a 1.5 made the pattern of a NaN and the pixel counted as masked.)
  {
    const float big = 0x1p127f;
    const float mag = __builtin_fmaxf(__builtin_fabsf(ih[0]), __builtin_fmaxf(__builtin_fabsf(ih[1]), __builtin_fabsf(ih[2])));
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(s.nm) : "v"(mag), "s"(big));
  }
}

// Optional output (SfmLossDesc.warped): the warped image I^ of row r, as the loss sees it (curr_proj_img, base_model.py:90-94;
// exactly 0 where the sample is not in view) -- the output lanes of the rows of this wave's chunk, written through the L2.
// Compiled into the WARPED instantiations only (picked by the host when the descriptor asks for the output): the kernels of a
// launch without it are instruction for instruction what they were -- this kernel sits at its register limit, and two more live
// scalar registers in the row loop showed up as 14 more scalar spills into vector-register lanes.
__device__ __forceinline__ void store_warped_row(const SsimCtx& C, const int r, const RowS& s) {
  if ((unsigned)(r - C.y0) < (unsigned)(C.y1 - C.y0) && C.outb) {
    const unsigned o = (unsigned)r * (unsigned)C.w + C.xc;
    stf_wt(C.wp, o, s.ih.p.x);
    stf_wt(C.wp + C.P, o, s.ih.p.y);
    stf_wt(C.wp + 2 * C.P, o, s.ih.s);
  }
}

// the nine per-lane sums of geometry_backward, the x / y components of each triple as one register pair
struct PoseAcc {
  f2 A, B, Cq;         // sum gq D, sum y gq D, sum gq   (components 0, 1)
  float A2, B2, C2;    // ... component 2
};
__device__ __forceinline__ void zero(PoseAcc& a) {
  a.A = a.B = a.Cq = T_of<f2>(0.f); a.A2 = a.B2 = a.C2 = 0.f;
}

// ------------------------------------------------------------------------------------------
// Optional dL/d(src) (SfmLossDesc.d_src; the reference discards it in training, models/base_model.py:71-72 `.data`): the scatter of
// dL/dI^ over the four taps of every sample, SURVEY.md App. A.3.  The main launch does NOT scatter: it RECORDS dL/dI^ of every warped
// pixel (three coalesced stores per pixel row into the workspace, geometry_backward below) and a second launch, dsrc_scatter_kernel
// (sfm_loss_dsrc.hip), re-projects the pixels and sums their taps in an LDS window per workgroup.  History of the scatter inside this
// kernel -- straight to memory 1.26 ms per cfg3 step (round 4), an LDS window per wave 0.51 (round 5) and 0.42 (round 6, first
// form: the window costs the kernel a wave per SIMD and half its chunk height) -- in profiles/r06_d_src.txt.
// ------------------------------------------------------------------------------------------

// From dL/dI^ of one pixel (already contracted with dI^/du, dI^/dv and 1/z into gq = (gq0, gq1)) to its
// share of d_depth (LDS tile), of the 12 sums of dL/dPm and, optionally, of dL/d(src).
// the arithmetic of geometry_backward for one (pixel, source): its share of dL/d(disp) (returned) and of the nine pose sums
__device__ __forceinline__ float geom_terms(const SsimCtx& C, const f2 UV, const float D, const float yf, const f2 gq, PoseAcc& gpm) {
  const f2 guv = gq * UV;
  const float gq2 = -(guv.x + guv.y);
  // (see geometry_backward for the derivation)
  const f2 t = gq * D;
  const float t2 = gq2 * D;
  const float gdisp = fmaf(t.x, C.P3[0], fmaf(t.y, C.P3[1], t2 * C.P3[2]));
  gpm.A += t; gpm.A2 += t2;
  gpm.B = vfma(T_of<f2>(yf), t, gpm.B); gpm.B2 = fmaf(yf, t2, gpm.B2);
  gpm.Cq += gq; gpm.C2 += gq2;
  return gdisp;
}

template <bool DSRC /* the launch also produces dL/d(src) */, int REF = 0>
__device__ __forceinline__ void geometry_backward(const SsimCtx& C, const RowS& s2, const int rc, const f2 gq,
                                                  const float* gI, float* gacc, const bool first, PoseAcc& gpm) {
  const int h = C.h, w = C.w;
  const float yf = (float)rc;
  // dL/d(depth) = gq . a with a = M (x,y,1) the ray of q = D a + P3.  D a = q - P3, and gq . q = 0 because U and
  // V are homogeneous of degree 0 in q (with z = q2 + 1e-10 it is (gq0 U + gq1 V) 1e-10: ten orders below gq . P3), so
  // gD = -(gq . P3) / D and dL/d(disp) = -gD D^2 (depth = 1/disp, base_model.py:60) = (gq . P3) D: three products with
  // wave-uniform factors instead of rebuilding the ray -- and without the cancellation of a0 - U a2 for small translations.
  // (with t = gq D, which the pose sums below need anyway: three products, the factor D is already inside)
  // dL/dPm[k][j] = sum over pixels of gq_k * c_j with c = D * (K1 y + kx) (the back-projected point), c_3 = 1.  The ray
  // is linear in the row, so a lane only accumulates  A_k = sum gq_k D,  B_k = sum y gq_k D,  C_k = sum gq_k  (9 values
  // instead of 12); pose_sums_raw reduces them over the wave once per pass and finalize_kernel multiplies K^-1 in
  const float gdisp = geom_terms(C, s2.UV, s2.D, yf, gq, gpm);
  float* ga = gacc + (rc - C.y0) * 64 + C.lane;
  // the tile is private to this wave: a plain store for the first contribution, then read-add-write through a register
  // (an LDS add without return value, ds_add_f32, has nothing to wait for but costs the launch 3 % at cfg3: it is a slow LDS op)
  if (first) *ga = gdisp;
  else *ga = *ga + gdisp;
  // (DSRC: the instantiations a launch with SfmLossDesc.d_src bound runs: they record dL/dI^ of the pixel for dsrc_scatter_kernel,
  //  which applies the in-view test itself.  Instantiations of their own so that no other launch carries the branch.)
  if (DSRC && C.dsp != nullptr && C.outb) {
    // (pixel-interleaved: one 12-byte store here, one 12-byte load there)
    struct __attribute__((packed, aligned(4))) Rec { float c[3]; };
    const unsigned o = (unsigned)rc * (unsigned)w + (unsigned)(C.x0 + C.lane);
    Rec g; g.c[0] = gI[0]; g.c[1] = gI[1]; g.c[2] = gI[2];
    reinterpret_cast<Rec*>(C.dsp)[o] = g;
  }
  (void)h;
}

// The pose sums of this (wave, source) from the 9 per-lane accumulators of geometry_backward.  dL/dPm[k][j] = sum over pixels of
// gq_k c_j with c = D K^-1 (x,y,1) the back-projected point (c_3 = 1), i.e. with A_k = sum gq_k D, B_k = sum y gq_k D, C_k = sum gq_k:
//   dL/dPm[k][j] = Kinv[j][0] (sum x A_k) + Kinv[j][1] B_k + Kinv[j][2] A_k ,   dL/dPm[k][3] = C_k .
// The wave writes the RAW sums (sum x A_k, B_k, A_k, C_k), k = 0..2; finalize_kernel, which holds K of the scale anyway for
// K^T . gPm, multiplies K^-1 in once per tile: K^-1 never enters the main kernel's passes.
// The twelve wave reductions run in LOCKSTEP (stage by stage over all twelve values): twelve independent DPP adds per stage
// instead of twelve dependent chains of six (each link of a chain waits for the previous one; 2.7-3.9k cycles per pass in
// profiles/r02_wave_stage_stamps.txt).  Same adds in the same order per value: the sums are bit-identical to wave_sum's.
__device__ __forceinline__ void pose_sums_raw(const SsimCtx& C, const PoseAcc& pa, float* gpm_out) {
  const float acc[9] = {pa.A.x, pa.A.y, pa.A2, pa.B.x, pa.B.y, pa.B2, pa.Cq.x, pa.Cq.y, pa.C2};
  const float xf = (float)(C.x0 + C.lane);
  float v[12];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v[k * 4 + 0] = xf * acc[k];
    v[k * 4 + 1] = acc[3 + k];
    v[k * 4 + 2] = acc[k];
    v[k * 4 + 3] = acc[6 + k];
  }
  wave_sums_lockstep(v);
  // lane 63 holds the twelve totals: it writes them (three 16-byte stores)
  if (C.lane == 63) {
    float4* o = reinterpret_cast<float4*>(gpm_out);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
    o[2] = make_float4(v[8], v[9], v[10], v[11]);
  }
}

// Stage B, split in the three parts that the two channel groups (T = f2: channels 0,1; T = float: channel 2) run in step:
//   x2,x1,x0 / y2,y1,y0: I^ and I of rows r-2, r-1, r;  out: horizontal 3-sums of the three SSIM partials of row r-1
// SSIM in scaled sums (Sx = 9 mu_x ...), see the header of this file; models/base_model.py:130-142.
template <typename T>
struct SsimSums { T Sx, Sy, Sqq, Sxy; };

// (1) vertical 3-sums over the ring rows r-2..r (in-lane); the horizontal ones follow for both groups together (ssim_stage_b_row)
template <typename T>
__device__ __forceinline__ void ssim_vsums(const T x2, const T x1, const T x0, const T y2, const T y1, const T y0, SsimSums<T>& o) {
  o.Sx = x2 + x1 + x0;
  o.Sy = y2 + y1 + y0;
  // sigma_x + sigma_y only ever appear together (base_model.py:138), so E[xx] and E[yy] are pooled as one field
  o.Sqq = vfma(x2, x2, vfma(x1, x1, vfma(x0, x0, vfma(y2, y2, vfma(y1, y1, y0 * y0)))));
  o.Sxy = vfma(x2, y2, vfma(x1, y1, x0 * y0));
}

// v where |S| < 1, else 0: F.clip backward of clip((1 - S) / 2, 0, 1) -- strictly inside (0, 1) exactly when -1 < S < 1
// (a NaN fails the test, like the clamped value it would produce)
__device__ __forceinline__ float vsel_abs_lt1(float S, float v) { return (fabsf(S) < 1.f) ? v : 0.f; }
// the packed pair: v * [1 - S^2 > 0] with the indicator as a clamped product (1 - S^2 is exact to the last bit next to |S| = 1: fma);
// three packed instructions instead of two compares and two selects
__device__ __forceinline__ f2 vsel_abs_lt1(f2 S, f2 v) {
  f2 big;
  big.x = 0x1p127f; big.y = 0x1p127f;
  const f2 t = vfma(-S, S, T_of<f2>(1.f));
  f2 m;
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(m) : "v"(t), "s"(big));
  return v * m;
}

// (2) SSIM value from the pooled sums and, with GRAD, its three partials BEFORE their horizontal 3-sums, in the scaling
//   ga = (1/9) kappa dS/dmu_x,   gb = (-2/81) kappa dS/dE[xx],   ge = (1/81) kappa dS/dE[xy]
// (kq2_nm carries kappa's factor 2: with these scalings no constant multiplies are left here; stage C applies the 9)
template <bool GRAD, bool LOSS, bool FIRST, typename T>
__device__ __forceinline__ void ssim_value_partials(const SsimSums<T>& p, const float kq2_nm, T& ga, T& gb, T& ge, float& ssum) {
  const float C1 = 81.f * 0.0001f, C2 = 81.f * 0.0009f;             // 81 c1, 81 c2   base_model.py:127-128
  const T Sx = p.Sx, Sy = p.Sy;
  const T pxy = Sx * Sy;
  const T sq = vfma(Sx, Sx, Sy * Sy);
  const T N1 = pxy * 2.f + C1;
  const T N2 = vfma(p.Sxy, T_of<T>(18.f), vfma(pxy, T_of<T>(-2.f), T_of<T>(C2)));   // (one non-inline constant per instruction: a packed
                                                                                     //  op takes a single scalar operand, a second one costs a register copy)
  const T D1 = sq + C1;
  const T D2 = vfma(p.Sqq, T_of<T>(9.f), C2 - sq);
  const T rD = vrcp(D1 * D2);
  const T Sv = N1 * N2 * rD;                                        // base_model.py:140
  // (1 - SSIM) / 2 clipped to [0, 1] (base_model.py:142) in ONE instruction: the clamp rides on the multiply-add as its output modifier
  if (LOSS) ssum = FIRST ? vhadd(half_one_minus_clamped(Sv)) : ssum + vhadd(half_one_minus_clamped(Sv));
  if (GRAD) {
    const T kap = vsel_abs_lt1(Sv, rD * kq2_nm);                    // 2 kappa / (D1 D2); F.clip backward
    const T u3 = vfma(-(Sv * Sx), D2 - D1, Sy * (N2 - N1));
    ga = kap * u3;
    gb = kap * Sv * D1;
    ge = kap * N1;
  }
}

// Stage C for one channel group: dL/dI^ of row r-2 from the transposed 3x3 pool of the SSIM partials plus the L1
// term (the contraction with dI^/du, dI^/dv and 1/z follows for all channels together: contract_uv)
template <typename T>
__device__ __forceinline__ void ssim_stage_c(const T a2, const T a1, const T a0, const T b2, const T b1, const T b0, const T e2,
                                             const T e1, const T e0, const T ih, const T it, const float kpn, T& g, T& diff) {
  const T Aq = a2 + a1 + a0, Bq = b2 + b1 + b0, Eq = e2 + e1 + e0;
  diff = ih - it;
  // dL/dI^ = A + 2 I^ B + I E in the partials' own scaling (ssim_value_partials): A + 9 (I E' - I^ B')
  g = add_ksign(vfma(vfma(it, Eq, -(ih * Bq)), T_of<T>(9.f), Aq), kpn, diff);
}

// Stage B at centre row rb (rows rb-1, rb, rb+1 in s2, s1, s0): SSIM value and, with GRAD, the horizontal 3-sums of its
// partials into g0.  `count` = whether the row's loss terms belong to this wave (a halo row gets weight 0 -- branch-free: a
// branch here splits the block and un-folds the DPP adds into moves + adds: 232 -> 249 vector instructions per step).
template <bool GRAD, bool LOSS>
__device__ __forceinline__ void ssim_stage_b_row(const SsimCtx& C, const RowS& s2, const RowS& s1, const RowS& s0, RowG& g0,
                                                 const bool count, float& acc_pix, float& acc_ssim) {
  float ssum = 0.f;
  const float kq2_nm = C.kq * s1.nm;   // 2 kappa before the clip test; s1.nm is 0 outside the image and on masked pixels (:114)
  // Both channel groups go through each part together: ONE run of DPP adds per horizontal 3-sum of all their fields (eight
  // pooled sums, then six partials) instead of one per group -- every run starts with the wait states of the DPP hazard.
  SsimSums<f2> pp;
  SsimSums<float> ps;
  ssim_vsums(s2.ih.p, s1.ih.p, s0.ih.p, s2.it.p, s1.it.p, s0.it.p, pp);
  ssim_vsums(s2.ih.s, s1.ih.s, s0.ih.s, s2.it.s, s1.it.s, s0.it.s, ps);
  hsum3_group(pp.Sx, pp.Sy, pp.Sqq, pp.Sxy, ps.Sx, ps.Sy, ps.Sqq, ps.Sxy);
  ssim_value_partials<GRAD, LOSS, true>(pp, kq2_nm, g0.a.p, g0.b.p, g0.e.p, ssum);
  ssim_value_partials<GRAD, LOSS, false>(ps, kq2_nm, g0.a.s, g0.b.s, g0.e.s, ssum);
  if (GRAD) hsum3_group(g0.a.p, g0.b.p, g0.e.p, g0.a.s, g0.b.s, g0.e.s);
  if (LOSS) {
    const float wgt = count ? s1.nm * C.outf : 0.f;
    acc_ssim = fmaf(ssum, wgt, acc_ssim);                            // base_model.py:114-115
    if (!GRAD) {                                                     // (with gradients stage C adds this term: it forms I^ - I anyway)
      const float e1 = vabs_sum(s1.ih.p - s1.it.p) + vabs_sum(s1.ih.s - s1.it.s);   // :95
      acc_pix = fmaf(e1, wgt, acc_pix);                              // :98-100,:111
    }
  }
}

// dL/dI^ of the three channels (gp: channels 0, 1; gs: channel 2) contracted with dI^/du, dI^/dv and 1/z: (dL/dq0, dL/dq1) as a pair
__device__ __forceinline__ f2 contract_uv(const RowS& s, const f2 gp, const float gs) {
  f2 hq = T_of<f2>(gp.x) * s.duv0;
  hq = vfma(T_of<f2>(gp.y), s.duv1, hq);
  hq = vfma(T_of<f2>(gs), s.duv_s, hq);
  return hq * s.rzi;
}

// Stage C at row rc (the row in s2; partials of the rows rc-1, rc, rc+1 in g2, g1, g0): dL/dI^ -> dL/d(u,v) -> dL/dq -> d_depth
// tile and the sums of dL/dPm.
template <bool LOSS, bool DSRC, int REF = 0>
__device__ __forceinline__ void ssim_stage_c_row(const SsimCtx& C, const int rc, const RowS& s2, const RowG& g2, const RowG& g1,
                                                 const RowG& g0, float* gacc, const bool first, PoseAcc& gpm, float& acc_pix) {
  const float kpn = C.k_pix * s2.nm;
  f2 gp, dp;
  float gs, ds;
  ssim_stage_c(g2.a.p, g1.a.p, g0.a.p, g2.b.p, g1.b.p, g0.b.p, g2.e.p, g1.e.p, g0.e.p, s2.ih.p, s2.it.p, kpn, gp, dp);
  ssim_stage_c(g2.a.s, g1.a.s, g0.a.s, g2.b.s, g1.b.s, g0.b.s, g2.e.s, g1.e.s, g0.e.s, s2.ih.s, s2.it.s, kpn, gs, ds);
  // the L1 term of this row (base_model.py:95-100,:111): stage C runs on exactly the rows whose loss terms belong to this wave,
  // and I^ - I is at hand (with the fused kernel stage B would form it a second time, one row earlier)
  if (LOSS) acc_pix = fmaf(vabs_sum(dp) + vabs_sum(ds), s2.nm * C.outf, acc_pix);
  const float gI[3] = {gp.x, gp.y, gs};
  geometry_backward<DSRC, REF>(C, s2, rc, contract_uv(s2, gp, gs), gI, gacc, first, gpm);
}

// Which stages run on which step of a pass, as bit k of one 32-bit word per question (a pass has at most 32 steps): every
// question is an interval of steps, the words are built once per pass, and a step asks with one s_bitcmp + branch.  (Asked
// with row arithmetic -- add, compare, select, branch per question -- the six questions were a quarter of the scalar
// instructions of a step, and a scalar instruction costs a wave as much issue time as a DPP add: profiles/r02_op_cost_microbench.txt.)
struct StepMasks {
  unsigned fin;    // row r of the step is inside the image: its gathers are in flight and are finished now
  unsigned iss;    // row r+1 is fetched by this pass: put it in flight
  unsigned b;      // stage B runs (centre row r-1)
  unsigned cnt;    // ... and its loss terms belong to this wave (centre row inside the chunk)
  unsigned c;      // stage C runs (row r-2 inside the chunk)
};
__device__ __forceinline__ unsigned step_range(int lo, int hi) {   // bits lo .. hi-1, clamped to 0 .. 32
  lo = min(max(lo, 0), 32);
  hi = min(max(hi, lo), 32);
  const unsigned long long one = 1ull;
  return (unsigned)(((one << hi) - 1ull) & ~((one << lo) - 1ull));
}
__device__ __forceinline__ bool step_bit(const unsigned m, const int k) { return ((m >> k) & 1u) != 0u; }

template <bool GRAD, bool LOSS, bool HWC, bool WARPED, int REF, bool DSRC>
__device__ __forceinline__ void ssim_row_step(const SsimCtx& C, const StepMasks& M, const int k, const int r, Pipe& ps, float& disp_next,
                                              RowS& s0, const RowS& s1, const RowS& s2,
                                              RowG& g0, const RowG& g1, const RowG& g2, float* gacc, const bool first,
                                              float& acc_pix, float& acc_ssim, PoseAcc& gpm SFM_STAMPS_ARG) {
  const int w = C.w;
#ifdef SFM_STAMPS
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#endif
  SFM_STAMP(t0);
  // ---------------- A: finish row r, put row r+1 in flight ----------------
  // (the arithmetic always runs -- on whatever the tap registers hold when the row lies outside the image -- and the rare case
  // overwrites the slot: an if / else costs five more scalar instructions per step than an if)
  finish_row<HWC>(C, ps, s0);
  if (!step_bit(M.fin, k)) zero_rare(s0);
  if constexpr (WARPED) store_warped_row(C, r, s0);
  SFM_STAMP(t1);
  if (step_bit(M.iss, k)) {
    issue_row<HWC, REF>(C, r + 1, disp_next, ps);
    // the disparity of the row after that; on the last fetched row of the pass the prefetch reads the image's last row again
    disp_next = ldf(C.dp, (unsigned)min(r + 2, C.h - 1) * (unsigned)w + C.xc);
  }
  SFM_STAMP(t2);

  // ---------------- B: SSIM at row r-1 ----------------
  // The first two steps of a pass only fill the ring: their centre rows lie above every row whose SSIM value
  // (forward) or SSIM partials (gradient, one more row) anything will read, so the whole stage is skipped
  // (wave-uniform branch).
  if (step_bit(M.b, k)) {
    ssim_stage_b_row<GRAD, LOSS>(C, s2, s1, s0, g0, step_bit(M.cnt, k), acc_pix, acc_ssim);
  }

  SFM_STAMP(t3);
  // ---------------- C: gradients at row r-2 ----------------
  if (GRAD) {
    if (step_bit(M.c, k)) {
      ssim_stage_c_row<LOSS, DSRC, REF>(C, r - 2, s2, g2, g1, g0, gacc, first, gpm, acc_pix);
    }
  }
  SFM_STAMP(t4);
#ifdef SFM_STAMPS
  st.a_fin += t1 - t0; st.a_iss += t2 - t1; st.b += t3 - t2; st.c += t4 - t3; st.steps += 1;
#endif
}

// One source of one wave.  HS = halo of this pass (2 with gradients, 1 forward only).
template <bool GRAD, bool LOSS, bool HWC, bool WARPED, int REF = 0, bool DSRC = false>
__device__ __forceinline__ void ssim_source_pass(const SsimCtx& C, float* gacc, const bool first, float& acc_pix, float& acc_ssim,
                                                 float* gpm_out /* 12 floats in global memory, or nullptr */ SFM_STAMPS_ARG) {
  constexpr int HS = GRAD ? 2 : 1;
  const int rbeg = C.y0 - HS, rend = C.y1 + HS;
  const int rload = min(rend, C.h);   // rows below are neither inside the image nor part of this pass: never fetched
  PoseAcc gpm;    // A_k, B_k, C_k of geometry_backward
  zero(gpm);
  // The rings need no initial value: stage B first runs on the third row of the pass, when all three RowS slots have
  // been written, and stage C two rows later, when all three RowG slots have (see ssim_row_step).
  RowS S0, S1, S2;
  RowG G0, G1, G2;
  Pipe ps;
  float disp_next = 1.f;
  // step k handles row r = rbeg + k: which stages it runs (see StepMasks)
  const int n = rend - rbeg, R = C.y1 - C.y0;
  StepMasks M;
  M.fin = step_range(-rbeg, C.h - rbeg);
  M.iss = step_range(-rbeg - 1, rload - rbeg - 1);
  M.b = step_range(HS + 1 - (GRAD ? 1 : 0), n);
  M.cnt = step_range(HS + 1, HS + 1 + R);
  M.c = step_range(HS + 2, HS + 2 + R);
  // prologue: row rbeg in flight, and the disparity of the first row the loop will put in flight (row rbeg + 1, or row 0 for
  // the chunk at the top of the image: the steps before it fetch nothing and leave disp_next alone)
  // (the two disparities a pass starts from are the same for every source: the wave loaded them once, at its start, so that no
  // pass waits for a disparity before it can even form the addresses of its first gathers)
  if (rbeg >= 0 && rbeg < C.h) issue_row<HWC, REF>(C, rbeg, C.disp_first, ps);
  disp_next = C.disp_second;
  for (int k = 0; k < n; k += 3) {
    const int r = rbeg + k;
    ssim_row_step<GRAD, LOSS, HWC, WARPED, REF, DSRC>(C, M, k, r, ps, disp_next, S0, S2, S1, G0, G2, G1, gacc, first, acc_pix, acc_ssim, gpm SFM_STAMPS_PASS);
    if (k + 1 < n)
      ssim_row_step<GRAD, LOSS, HWC, WARPED, REF, DSRC>(C, M, k + 1, r + 1, ps, disp_next, S1, S0, S2, G1, G0, G2, gacc, first, acc_pix, acc_ssim, gpm SFM_STAMPS_PASS);
    if (k + 2 < n)
      ssim_row_step<GRAD, LOSS, HWC, WARPED, REF, DSRC>(C, M, k + 2, r + 2, ps, disp_next, S2, S1, S0, G2, G1, G0, gacc, first, acc_pix, acc_ssim, gpm SFM_STAMPS_PASS);
  }
  if (GRAD) {
    pose_sums_raw(C, gpm, gpm_out);
  }
}

// Photometric pass WITHOUT SSIM for one (wave, source): L1 (+ explainability weighting, base_model.py:103-109).
// Everything is per pixel, so there is no ring; the loads of row r+1 are in flight while row r is finished.
template <bool GRAD, bool LOSS, bool EXPL, bool HWC, bool WARPED, int REF = 0, bool DSRC = false>
__device__ __forceinline__ void l1_source_pass(const SsimCtx& C, float* gacc, const bool first, float& acc_pix, float& acc_exp,
                                               float* gpm_out) {
  PoseAcc gpm;    // A_k, B_k, C_k of geometry_backward
  zero(gpm);
  Pipe ps;
  ps.lg = 0.f;
  float disp_next = 1.f;
  const int rbeg = C.y0, rend = C.y1;    // rows of a chunk are always inside the image
  issue_row<HWC, REF>(C, rbeg, C.disp_first, ps);
  if (rbeg + 1 < rend) disp_next = C.disp_second;
  for (int r = rbeg; r < rend; ++r) {
    RowS s0;
    finish_row<HWC>(C, ps, s0);
    if constexpr (WARPED) store_warped_row(C, r, s0);
    const float lg = ps.lg;
    if (r + 1 < rend) issue_row<HWC, REF>(C, r + 1, disp_next, ps);
    disp_next = ldf(C.dp, (unsigned)min(r + 2, C.h - 1) * (unsigned)C.w + C.xc);
    float sgm = 1.f;
    if (EXPL) {
      sgm = rcp(1.0f + __expf(-lg));                                  // F.sigmoid, base_model.py:107
      if (LOSS) acc_exp += C.outf * (fmaxf(-lg, 0.f) + __logf(1.0f + __expf(-fabsf(lg))));   // softplus(-x), :165-167
    }
    const float e1 = (vabs_sum(s0.ih.p - s0.it.p) + vabs_sum(s0.ih.s - s0.it.s)) * s0.nm;   // :95-100
    if (LOSS) acc_pix = fmaf(e1 * sgm, C.outf, acc_pix);              // :109 / :111
    if (GRAD) {
      const float kpn = C.k_pix * s0.nm * sgm;
      const f2 gp = ksign(kpn, s0.ih.p - s0.it.p);
      const float gs = ksign(kpn, s0.ih.s - s0.it.s);
      const float gI[3] = {gp.x, gp.y, gs};
      if (EXPL) {
        // d/dlogit of (1-alpha) mean(err sigmoid) + exp_reg mean(softplus(-logit))
        if (C.outf != 0.f) stf_wt(C.dmp, (unsigned)r * (unsigned)C.w + C.xc, C.k_pix * e1 * sgm * (1.f - sgm) + C.k_exp * (sgm - 1.f));
      }
      geometry_backward<DSRC, REF>(C, s0, r, contract_uv(s0, gp, gs), gI, gacc, first, gpm);
    }
  }
  if (GRAD) {
    pose_sums_raw(C, gpm, gpm_out);
  }
}

}  // namespace sfm
