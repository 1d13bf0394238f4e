// Operator-level kernels of libsfmwarp: the individually callable pieces of the view-synthesis
// path (pose -> projection, projective inverse warp, the two samplers, pyramid resize).
// The multi-scale fused loss lives in sfm_loss.hip.  gfx950 only.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "sfm_common.h"

#define SFM_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return fail(code, __VA_ARGS__); \
  } while (0)

namespace sfm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return SFM_OK;
}

// ------------------------------------------------------------------------------------------
// pose -> projection  (models/transform.py:64-91)
// ------------------------------------------------------------------------------------------
__global__ void pose_proj_fwd_kernel(const float* __restrict__ pose6, const float* __restrict__ K,
                                     float* __restrict__ proj, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Geom g;
  make_geom(pose6 + n * 6, K + n * 9, g);
  float* o = proj + n * 16;
#pragma unroll
  for (int k = 0; k < 12; ++k) o[k] = g.P[k];
  o[12] = 0.f;
  o[13] = 0.f;
  o[14] = 0.f;
  o[15] = 1.f;
}

// gT3 (3x4) = K^T . gPm[0:3, :]   (K4^T . gPm restricted to the rows that reach R and t)
__device__ __forceinline__ void kt_times_gpm(const float* K, const float* gPm3x4, float* gT3, bool accumulate) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = K[0 * 3 + i] * gPm3x4[0 * 4 + j] + K[1 * 3 + i] * gPm3x4[1 * 4 + j] + K[2 * 3 + i] * gPm3x4[2 * 4 + j];
      gT3[i * 4 + j] = accumulate ? gT3[i * 4 + j] + v : v;
    }
}

__global__ void pose_proj_bwd_kernel(const float* __restrict__ pose6, const float* __restrict__ K,
                                     const float* __restrict__ g_proj, float* __restrict__ d_pose6, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float gT3[12];
  kt_times_gpm(K + n * 9, g_proj + n * 16, gT3, false);
  float d[6];
  pose_backward(pose6 + n * 6, gT3, d);
#pragma unroll
  for (int k = 0; k < 6; ++k) d_pose6[n * 6 + k] = d[k];
}

// ------------------------------------------------------------------------------------------
// F.spatial_transformer_sampler (call site models/transform.py:189): general semantics on the
// zero-padded image, for arbitrary grids
// ------------------------------------------------------------------------------------------
struct PadTap {
  int u0, v0;             // top-left tap in PADDED coordinates, u0 in [0,W], v0 in [0,H]
  float wx0, wx1, wy0, wy1;
  bool ok_u, ok_v;        // coordinate inside the padded image (gradient mask)
};

__device__ __forceinline__ PadTap pad_taps(float gx, float gy, int H, int W) {
#pragma clang fp contract(off)
  PadTap t;
  const float up = (gx + 1.0f) * (float)(W - 1) * 0.5f + 1.0f;
  const float vp = (gy + 1.0f) * (float)(H - 1) * 0.5f + 1.0f;
  const float uc = fminf(fmaxf(up, 0.0f), (float)(W + 1));
  const float vc = fminf(fmaxf(vp, 0.0f), (float)(H + 1));
  t.u0 = min(max((int)floorf(uc), 0), W);
  t.v0 = min(max((int)floorf(vc), 0), H);
  t.wx0 = (float)(t.u0 + 1) - uc;
  t.wx1 = uc - (float)t.u0;
  t.wy0 = (float)(t.v0 + 1) - vc;
  t.wy1 = vc - (float)t.v0;
  t.ok_u = (up >= 0.0f) && (up <= (float)(W + 1));
  t.ok_v = (vp >= 0.0f) && (vp <= (float)(H + 1));
  return t;
}

__device__ __forceinline__ float pad_read(const float* img, int v, int u, int H, int W) {  // padded coords
  return (u >= 1 && u <= W && v >= 1 && v <= H) ? img[(v - 1) * W + (u - 1)] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// projective_inverse_warp  (models/transform.py:156-193) -- the API-parity operator.
//
// Unlike the fused loss kernels (which pre-multiply the geometry), this operator keeps the REFERENCE'S
// evaluation order, step by step and without fused multiply-adds:
//   ray = K^-1 . (x, y, 1)                         transform.py:105-106   (batch_matmul: left to right over k)
//   c   = D (.) ray ; c4 = (c, 1)                  :107-108
//   q   = Pm . c4 ; z = q2 + 1e-10                 :122-123
//   xn  = (q0 / z) / ((W-1)/2.) - 1 ; yn likewise  :124-125
//   each component not strictly inside (-1, 1) is doubled   :128-131
//   F.spatial_transformer_sampler on the zero-padded image  :189  (pad_taps / pad_read above)
// so that the set of exactly-zero output pixels and the sampling positions are the reference's own.
// one block = 256 consecutive pixels of one sample; the block's geometry is built once in LDS
// ------------------------------------------------------------------------------------------
constexpr int WARP_BLOCK = 256;

struct RefProj {
  float ray[3], c[3];   // K^-1 . pix ; D (.) ray
  float z, U, V;        // q2 + 1e-10 ; q0 / z ; q1 / z
  float mx, my;         // 1 inside (-1, 1), else 2     (transform.py:128-130)
  float gx, gy;         // the grid coordinates handed to the sampler (xn * mx, yn * my)
};

__device__ __forceinline__ RefProj ref_project(const Geom& g, const float xf, const float yf, const float* D, const int H, const int W) {
#pragma clang fp contract(off)
  RefProj r;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.ray[j] = (g.Kinv[j * 3 + 0] * xf + g.Kinv[j * 3 + 1] * yf) + g.Kinv[j * 3 + 2];   // the third coordinate of pix is 1
    r.c[j] = D[j] * r.ray[j];
  }
  float q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = ((g.P[k * 4 + 0] * r.c[0] + g.P[k * 4 + 1] * r.c[1]) + g.P[k * 4 + 2] * r.c[2]) + g.P[k * 4 + 3];
  r.z = q[2] + 1e-10f;
  r.U = q[0] / r.z;
  r.V = q[1] / r.z;
  const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
  const float xn = r.U / half_w - 1.0f, yn = r.V / half_h - 1.0f;
  r.mx = (xn > -1.0f && xn < 1.0f) ? 1.0f : 2.0f;     // NaN compares false: doubled, stays NaN
  r.my = (yn > -1.0f && yn < 1.0f) ? 1.0f : 2.0f;
  r.gx = xn * r.mx;
  r.gy = yn * r.my;
  return r;
}

__device__ __forceinline__ void load_depth3(const float* depth, const int n, const int drows, const int P, const int j, float* D) {
  if (drows == 1) {   // one row of the reference's (N,3,H*W) broadcast (base_model.py:82-84)
    D[0] = D[1] = D[2] = depth[(size_t)n * P + j];
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) D[r] = depth[((size_t)n * 3 + r) * P + j];
  }
}

__global__ void __launch_bounds__(WARP_BLOCK) warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                              const float* __restrict__ pose6, const float* __restrict__ K,
                                                              float* __restrict__ warped, int C, int H, int W, int drows) {
#pragma clang fp contract(off)
  __shared__ Geom g;
  const int n = blockIdx.y;
  if (threadIdx.x == 0) make_geom(pose6 + n * 6, K + n * 9, g);
  __syncthreads();
  const int P = H * W;
  const int j = blockIdx.x * WARP_BLOCK + threadIdx.x;
  if (j >= P) return;
  const int y = j / W, x = j - y * W;
  float D[3];
  load_depth3(depth, n, drows, P, j, D);
  const RefProj r = ref_project(g, (float)x, (float)y, D, H, W);
  const PadTap t = pad_taps(r.gx, r.gy, H, W);
  const float w1 = t.wx0 * t.wy0, w2 = t.wx1 * t.wy0, w3 = t.wx0 * t.wy1, w4 = t.wx1 * t.wy1;
  for (int c = 0; c < C; ++c) {
    const float* img = src + ((size_t)n * C + c) * P;
    float v = w1 * pad_read(img, t.v0, t.u0, H, W);
    v += w2 * pad_read(img, t.v0, t.u0 + 1, H, W);
    v += w3 * pad_read(img, t.v0 + 1, t.u0, H, W);
    v += w4 * pad_read(img, t.v0 + 1, t.u0 + 1, H, W);
    warped[((size_t)n * C + c) * P + j] = v;
  }
}

// per pixel: the reference's backward chain (sampler -> x mask -> normalisation -> perspective division -> Pm . c4 -> D (.) ray),
// d_depth, and the 12 sums of gPm (block-reduced into ws)
__global__ void __launch_bounds__(WARP_BLOCK) warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                              const float* __restrict__ pose6, const float* __restrict__ K,
                                                              const float* __restrict__ g_warped, float* __restrict__ d_depth,
                                                              float* __restrict__ d_src, float* __restrict__ part, int C, int H,
                                                              int W, int drows) {
#pragma clang fp contract(off)
  __shared__ Geom g;
  __shared__ float red[WARP_BLOCK / 64][12];
  const int n = blockIdx.y;
  if (threadIdx.x == 0) make_geom(pose6 + n * 6, K + n * 9, g);
  __syncthreads();
  const int P = H * W;
  const int j = blockIdx.x * WARP_BLOCK + threadIdx.x;
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  if (j < P) {
    const int y = j / W, x = j - y * W;
    float D[3];
    load_depth3(depth, n, drows, P, j, D);
    const RefProj r = ref_project(g, (float)x, (float)y, D, H, W);
    const PadTap t = pad_taps(r.gx, r.gy, H, W);
    float gu = 0.f, gv = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* img = src + ((size_t)n * C + c) * P;
      const float gc = g_warped[((size_t)n * C + c) * P + j];
      const float x1 = pad_read(img, t.v0, t.u0, H, W), x2 = pad_read(img, t.v0, t.u0 + 1, H, W);
      const float x3 = pad_read(img, t.v0 + 1, t.u0, H, W), x4 = pad_read(img, t.v0 + 1, t.u0 + 1, H, W);
      gu += gc * (-t.wy0 * x1 + t.wy0 * x2 - t.wy1 * x3 + t.wy1 * x4);
      gv += gc * (-t.wx0 * x1 - t.wx1 * x2 + t.wx0 * x3 + t.wx1 * x4);
      if (d_src) {
        float* o = d_src + ((size_t)n * C + c) * P;
        const int u = t.u0, v = t.v0;   // padded coordinates: taps on the zero frame receive nothing
        if (u >= 1 && u <= W && v >= 1 && v <= H) atomicAdd(o + (v - 1) * W + (u - 1), gc * t.wx0 * t.wy0);
        if (u + 1 >= 1 && u + 1 <= W && v >= 1 && v <= H) atomicAdd(o + (v - 1) * W + u, gc * t.wx1 * t.wy0);
        if (u >= 1 && u <= W && v + 1 >= 1 && v + 1 <= H) atomicAdd(o + v * W + (u - 1), gc * t.wx0 * t.wy1);
        if (u + 1 >= 1 && u + 1 <= W && v + 1 >= 1 && v + 1 <= H) atomicAdd(o + v * W + u, gc * t.wx1 * t.wy1);
      }
    }
    // sampler backward to the grid, then p_s_xy *= mask (transform.py:131)
    const float ggx = t.ok_u ? gu * ((float)(W - 1) * 0.5f) : 0.f;
    const float ggy = t.ok_v ? gv * ((float)(H - 1) * 0.5f) : 0.f;
    const float half_w = (float)((double)(W - 1) / 2.0), half_h = (float)((double)(H - 1) / 2.0);
    const float gU = (ggx * r.mx) / half_w, gV = (ggy * r.my) / half_h;
    const float gq0 = gU / r.z, gq1 = gV / r.z;
    const float gq2 = -(gU * r.U + gV * r.V) / r.z;
    // g_c = Pm^T . gq ; g_depthes[j] = g_c[j] * ray[j]   (transform.py:107,122 backward)
    const float gd0 = ((g.P[0] * gq0 + g.P[4] * gq1) + g.P[8] * gq2) * r.ray[0];
    const float gd1 = ((g.P[1] * gq0 + g.P[5] * gq1) + g.P[9] * gq2) * r.ray[1];
    const float gd2 = ((g.P[2] * gq0 + g.P[6] * gq1) + g.P[10] * gq2) * r.ray[2];
    if (drows == 1) {
      d_depth[(size_t)n * P + j] = (gd0 + gd1) + gd2;     // broadcast_to backward: sum of the three rows
    } else {
      d_depth[((size_t)n * 3 + 0) * P + j] = gd0;
      d_depth[((size_t)n * 3 + 1) * P + j] = gd1;
      d_depth[((size_t)n * 3 + 2) * P + j] = gd2;
    }
    acc[0] = gq0 * r.c[0]; acc[1] = gq0 * r.c[1]; acc[2] = gq0 * r.c[2];  acc[3] = gq0;
    acc[4] = gq1 * r.c[0]; acc[5] = gq1 * r.c[1]; acc[6] = gq1 * r.c[2];  acc[7] = gq1;
    acc[8] = gq2 * r.c[0]; acc[9] = gq2 * r.c[1]; acc[10] = gq2 * r.c[2]; acc[11] = gq2;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WARP_BLOCK / 64; ++w) s += red[w][threadIdx.x];
    part[((size_t)n * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = s;
  }
}

// one wave per sample: fixed-order sum of the block partials, then the pose backward
__global__ void __launch_bounds__(64) warp_bwd_pose_kernel(const float* __restrict__ pose6, const float* __restrict__ K,
                                                           const float* __restrict__ part, float* __restrict__ d_pose6,
                                                           int nblk) {
  const int n = blockIdx.x, lane = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.0;
  for (int b = lane; b < nblk; b += 64)
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] += (double)part[((size_t)n * nblk + b) * 12 + k];
  float gPm[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    double v = acc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    gPm[k] = (float)v;
  }
  if (lane == 0) {
    float gT3[12], d[6];
    kt_times_gpm(K + n * 9, gPm, gT3, false);
    pose_backward(pose6 + n * 6, gT3, d);
#pragma unroll
    for (int k = 0; k < 6; ++k) d_pose6[n * 6 + k] = d[k];
  }
}

// ------------------------------------------------------------------------------------------
// F.spatial_transformer_sampler kernels (helpers above)
// ------------------------------------------------------------------------------------------
__global__ void sampler_fwd_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ y, int C,
                                   int H, int W, int oP) {
  const int n = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= oP) return;
  const PadTap t = pad_taps(grid[((size_t)n * 2 + 0) * oP + j], grid[((size_t)n * 2 + 1) * oP + j], H, W);
  const float w1 = t.wx0 * t.wy0, w2 = t.wx1 * t.wy0, w3 = t.wx0 * t.wy1, w4 = t.wx1 * t.wy1;
  for (int c = 0; c < C; ++c) {
    const float* img = x + ((size_t)n * C + c) * H * W;
    float v = w1 * pad_read(img, t.v0, t.u0, H, W);
    v += w2 * pad_read(img, t.v0, t.u0 + 1, H, W);
    v += w3 * pad_read(img, t.v0 + 1, t.u0, H, W);
    v += w4 * pad_read(img, t.v0 + 1, t.u0 + 1, H, W);
    y[((size_t)n * C + c) * oP + j] = v;
  }
}

// gx is a scatter: every output pixel adds to its four taps (float atomics: 12 per pixel at C = 3).  A float atomic is priced
// per 64-byte request at the memory side, i.e. per wave-instruction and row it touches (MI355X_MICROARCH.md), so what counts is
// the number of atomic INSTRUCTIONS with many active lanes.  A wave owns 64 consecutive output columns and walks DOWN a segment
// of SAMPLER_BWD_ROWS output rows; in a warp field neighbouring output pixels sample neighbouring cells:
//   * the pixel to the RIGHT usually has its cell one column further (same tap rows): its left-hand taps are my right-hand
//     taps, so I take its two left-hand shares along (DPP shift) and it skips those two atomics;
//   * the pixel BELOW usually has its cell one row further (same tap columns): its top taps are my bottom taps, so the bottom
//     shares of a row are not written but CARRIED into the next row's top shares.
// In the interior of a smooth field one atomic instruction per row and channel is left (the top-right taps) instead of four;
// where cells do not line up the carried shares are written at once (a sparse instruction).  On a random grid nothing merges
// and nothing is lost.  Which shares travel depends only on the grid, not on timing.
__device__ __forceinline__ int int_from_left(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int int_from_right(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true); }

constexpr int SAMPLER_BWD_ROWS = 8;        // output rows a wave walks down
constexpr int SAMPLER_BWD_MAXC = 4;        // channels whose carried shares live in registers (more channels: no vertical carry)

template <bool CARRY>
__global__ void __launch_bounds__(256) sampler_bwd_kernel(const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ gy,
                                                          float* __restrict__ ggrid, float* __restrict__ gx, int C, int H, int W, int oH, int oW) {
  const int n = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ox = blockIdx.x * 64 + lane;
  const int oy0 = (blockIdx.y * 4 + wave) * SAMPLER_BWD_ROWS;
  if (oy0 >= oH) return;                           // whole wave
  const bool col_ok = ox < oW;                     // no early exit of single lanes: the lanes of a wave exchange values below
  const int oP = oH * oW;
  const bool scatter = gx != nullptr;
  // the bottom shares of the previous row that have not been written yet (per channel), and the cell they belong to
  float pend_l[SAMPLER_BWD_MAXC], pend_r[SAMPLER_BWD_MAXC];
  bool pend = false, pend_l_owned = false;
  int pu = 0, pv = 0;
#pragma unroll
  for (int c = 0; c < SAMPLER_BWD_MAXC; ++c) pend_l[c] = pend_r[c] = 0.f;
  const int rows = min(SAMPLER_BWD_ROWS, oH - oy0);
  for (int r = 0; r < rows; ++r) {
    const int oy = oy0 + r;
    const bool valid = col_ok;
    const int j = oy * oW + (valid ? ox : oW - 1);
    const PadTap t = pad_taps(grid[((size_t)n * 2 + 0) * oP + j], grid[((size_t)n * 2 + 1) * oP + j], H, W);
    const int u = t.u0, v = t.v0;
    // horizontal pairing: my right-hand column is the right neighbour's left-hand column, in the same pair of rows
    const int nu = int_from_right(u), nv = int_from_right(v), nvalid = int_from_right(valid ? 1 : 0);
    const bool take_right = scatter && valid && lane != 63 && nvalid != 0 && nv == v && nu == u + 1;
    const bool skip_left = int_from_left(take_right ? 1 : 0) != 0;     // lane 0 receives 0
    // vertical pairing: the cell of the previous row sits right above this one
    const bool match = CARRY && pend && valid && pv + 1 == v && pu == u;
    const bool in_u0 = u >= 1 && u <= W, in_u1 = u + 1 >= 1 && u + 1 <= W;
    const bool in_v0 = v >= 1 && v <= H, in_v1 = v + 1 >= 1 && v + 1 <= H;
    const bool pin_u0 = pu >= 1 && pu <= W, pin_u1 = pu + 1 >= 1 && pu + 1 <= W, pin_v1 = pv + 1 >= 1 && pv + 1 <= H;
    float gu = 0.f, gv = 0.f;
    auto channel = [&](const int c) {
      {
        const float* img = x + ((size_t)n * C + c) * H * W;
        const float g = valid ? gy[((size_t)n * C + c) * oP + j] : 0.f;
        const float x1 = pad_read(img, v, u, H, W), x2 = pad_read(img, v, u + 1, H, W);
        const float x3 = pad_read(img, v + 1, u, H, W), x4 = pad_read(img, v + 1, u + 1, H, W);
        gu += g * (-t.wy0 * x1 + t.wy0 * x2 - t.wy1 * x3 + t.wy1 * x4);
        gv += g * (-t.wx0 * x1 - t.wx1 * x2 + t.wx0 * x3 + t.wx1 * x4);
        if (scatter) {   // uniform
          float* o = gx + ((size_t)n * C + c) * H * W;
          float a00 = g * t.wx0 * t.wy0, a10 = g * t.wx0 * t.wy1;
          float a01 = g * t.wx1 * t.wy0, a11 = g * t.wx1 * t.wy1;
          const float n00 = from_right(a00), n10 = from_right(a10);   // the right neighbour's left-hand shares
          if (take_right) { a01 += n00; a11 += n10; }
          if (CARRY) {
            // what the previous row left pending either joins this row's top shares (same addresses) or is written now
            if (pend) {
              if (match) a01 += pend_r[c];
              else if (pin_u1 && pin_v1) atomicAdd(o + pv * W + pu, pend_r[c]);
              if (pend_l_owned) {
                if (match && !skip_left) a00 += pend_l[c];
                else if (pin_u0 && pin_v1) atomicAdd(o + pv * W + (pu - 1), pend_l[c]);
              }
            }
            if (valid) {
              if (!skip_left && in_u0 && in_v0) atomicAdd(o + (v - 1) * W + (u - 1), a00);
              if (in_u1 && in_v0) atomicAdd(o + (v - 1) * W + u, a01);
            }
            pend_l[c] = a10; pend_r[c] = a11;
          } else if (valid) {
            if (!skip_left) {
              if (in_u0 && in_v0) atomicAdd(o + (v - 1) * W + (u - 1), a00);
              if (in_u0 && in_v1) atomicAdd(o + v * W + (u - 1), a10);
            }
            if (in_u1 && in_v0) atomicAdd(o + (v - 1) * W + u, a01);
            if (in_u1 && in_v1) atomicAdd(o + v * W + u, a11);
          }
        }
      }
    };
    if constexpr (CARRY) {   // a compile-time bound: the carried shares are register arrays
#pragma unroll
      for (int c = 0; c < SAMPLER_BWD_MAXC; ++c)
        if (c < C) channel(c);
    } else {
      for (int c = 0; c < C; ++c) channel(c);
    }
    if (CARRY) { pend = scatter && valid; pend_l_owned = !skip_left; pu = u; pv = v; }
    if (valid) {
      ggrid[((size_t)n * 2 + 0) * oP + j] = t.ok_u ? gu * ((float)(W - 1) * 0.5f) : 0.f;
      ggrid[((size_t)n * 2 + 1) * oP + j] = t.ok_v ? gv * ((float)(H - 1) * 0.5f) : 0.f;
    }
  }
  if (CARRY && pend) {   // the bottom shares of the segment's last row
    const bool pin_u0 = pu >= 1 && pu <= W, pin_u1 = pu + 1 >= 1 && pu + 1 <= W, pin_v1 = pv + 1 >= 1 && pv + 1 <= H;
#pragma unroll
    for (int c = 0; c < SAMPLER_BWD_MAXC; ++c) {
      if (c < C) {
        float* o = gx + ((size_t)n * C + c) * H * W;
        if (pin_u1 && pin_v1) atomicAdd(o + pv * W + pu, pend_r[c]);
        if (pend_l_owned && pin_u0 && pin_v1) atomicAdd(o + pv * W + (pu - 1), pend_l[c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// SpatialTransformerSamplerInterp (models/spational_transformer_sampler_interp.py:32-149)
// ------------------------------------------------------------------------------------------
struct InterpTap {
  int u0, v0, u1, v1;
  float wx0, wx1, wy0, wy1;
};

__device__ __forceinline__ InterpTap interp_taps(float u, float v, int H, int W) {
#pragma clang fp contract(off)
  InterpTap t;
  float u0 = floorf(u), v0 = floorf(v);                                    // :41-44
  float u1 = u0 + 1.0f, v1 = v0 + 1.0f;
  u0 = fminf(fmaxf(u0, 0.0f), (float)(W - 1));                             // :46-49
  v0 = fminf(fmaxf(v0, 0.0f), (float)(H - 1));
  u1 = fminf(fmaxf(u1, 0.0f), (float)(W - 1));
  v1 = fminf(fmaxf(v1, 0.0f), (float)(H - 1));
  t.wx0 = u1 - u;                                                          // :52-55
  t.wx1 = u - u0;
  t.wy0 = v1 - v;
  t.wy1 = v - v0;
  t.u0 = (int)u0; t.v0 = (int)v0; t.u1 = (int)u1; t.v1 = (int)v1;         // :66-69
  return t;
}

__global__ void interp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ y, int C,
                                  int H, int W, int oP) {
#pragma clang fp contract(off)
  const int n = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= oP) return;
  const InterpTap t = interp_taps(grid[((size_t)n * 2 + 0) * oP + j], grid[((size_t)n * 2 + 1) * oP + j], H, W);
  const float w1 = t.wx0 * t.wy0, w2 = t.wx1 * t.wy0, w3 = t.wx0 * t.wy1, w4 = t.wx1 * t.wy1;   // :57-60
  for (int c = 0; c < C; ++c) {
    const float* img = x + ((size_t)n * C + c) * H * W;
    float v = w1 * img[t.v0 * W + t.u0];                                   // :72-75
    v += w2 * img[t.v0 * W + t.u1];
    v += w3 * img[t.v1 * W + t.u0];
    v += w4 * img[t.v1 * W + t.u1];
    y[((size_t)n * C + c) * oP + j] = v;
  }
}

__global__ void interp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ grid, const float* __restrict__ gy,
                                  float* __restrict__ ggrid, int C, int H, int W, int oP) {
#pragma clang fp contract(off)
  const int n = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= oP) return;
  const InterpTap t = interp_taps(grid[((size_t)n * 2 + 0) * oP + j], grid[((size_t)n * 2 + 1) * oP + j], H, W);
  float su = 0.f, sv = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* img = x + ((size_t)n * C + c) * H * W;
    const float g = gy[((size_t)n * C + c) * oP + j];
    const float x1 = img[t.v0 * W + t.u0], x2 = img[t.v0 * W + t.u1];
    const float x3 = img[t.v1 * W + t.u0], x4 = img[t.v1 * W + t.u1];
    float gu = -t.wy0 * x1;                                                // :129-132
    gu += t.wy0 * x2;
    gu -= t.wy1 * x3;
    gu += t.wy1 * x4;
    float gv = -t.wx0 * x1;                                                // :134-137
    gv -= t.wx1 * x2;
    gv += t.wx0 * x3;
    gv += t.wx1 * x4;
    su += gu * g;                                                          // :142-145
    sv += gv * g;
  }
  ggrid[((size_t)n * 2 + 0) * oP + j] = su;
  ggrid[((size_t)n * 2 + 1) * oP + j] = sv;
}

// ------------------------------------------------------------------------------------------
// F.resize_images, align-corners bilinear (models/base_model.py:70-72)
// ------------------------------------------------------------------------------------------
__global__ void resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int oH, int oW) {
#pragma clang fp contract(off)
  const int nc = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= oH * oW) return;
  const int oy = j / oW, ox = j - oy * oW;
  // numpy.linspace(0, W-1, oW)[ox] = ox * step with step = (W-1)/(oW-1), evaluated in double
  const float u = oW > 1 ? (float)((double)ox * ((double)(W - 1) / (double)(oW - 1))) : 0.f;
  const float v = oH > 1 ? (float)((double)oy * ((double)(H - 1) / (double)(oH - 1))) : 0.f;
  const int u0 = min(max((int)floorf(u), 0), max(W - 2, 0)), v0 = min(max((int)floorf(v), 0), max(H - 2, 0));
  const int u1 = min(u0 + 1, W - 1), v1 = min(v0 + 1, H - 1);
  const float wu1 = u - (float)u0, wv1 = v - (float)v0;
  const float wu0 = 1.0f - wu1, wv0 = 1.0f - wv1;
  const float* img = x + (size_t)nc * H * W;
  const float top = img[v0 * W + u0] * wu0 + img[v0 * W + u1] * wu1;
  const float bot = img[v1 * W + u0] * wu0 + img[v1 * W + u1] * wu1;
  y[(size_t)nc * oH * oW + j] = top * wv0 + bot * wv1;
}

// ------------------------------------------------------------------------------------------
// The whole image pyramid of a step in one launch (models/base_model.py:69-72): scales 1..S-1 of
// F.resize_images(x, (H >> s, W >> s)), each resampled from the full-resolution input as the
// reference does.  One thread per output pixel of any scale.
// ------------------------------------------------------------------------------------------
struct PyramidArgs {
  const float* x;
  const float* x2;               // pair form only: second input tensor (N,3*G2,H,W) and its outputs
  float* y2[SFM_MAX_SCALES];
  int n_first, G2;               // pair form only: blockIdx.y < n_first -> first tensor (one image per sample)
  float* y[SFM_MAX_SCALES];      // y[s] for s = 1..n_scales-1 (y[0] unused: scale 0 is the input itself)
  int oH[SFM_MAX_SCALES], oW[SFM_MAX_SCALES];
  int begin[SFM_MAX_SCALES + 1]; // prefix sums of oH*oW over scales 1..   (HWC forms: of the THREAD counts, scale 0 first)
  int H, W, n_scales;
  // numpy.linspace(0, W-1, oW)[ox] = ox * step with step = (W-1)/(oW-1) in double (F.resize_images): the quotient is the
  // same for every pixel of a scale, so the host computes it (a double division is ~40 instructions per thread)
  double step_u[SFM_MAX_SCALES], step_v[SFM_MAX_SCALES];
  float inv_oW[SFM_MAX_SCALES];  // 1 / oW, for the row of a flat index (corrected to the exact quotient in the kernel)
  int quads0;                    // HWC forms: threads of scale 0, four pixels each (ceil(H W / 4))
};

static void pyramid_steps(PyramidArgs& A) {
  for (int s = 0; s < A.n_scales; ++s) {
    A.step_u[s] = A.oW[s] > 1 ? (double)(A.W - 1) / (double)(A.oW[s] - 1) : 0.0;
    A.step_v[s] = A.oH[s] > 1 ? (double)(A.H - 1) / (double)(A.oH[s] - 1) : 0.0;
    A.inv_oW[s] = 1.0f / (float)A.oW[s];
  }
}

__global__ void pyramid_fwd_kernel(const PyramidArgs A) {
#pragma clang fp contract(off)
  const int nc = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.begin[A.n_scales]) return;
  int s = 1;
#pragma unroll
  for (int k = 2; k < SFM_MAX_SCALES; ++k)
    if (k < A.n_scales && j >= A.begin[k]) s = k;
  const int jj = j - A.begin[s];
  const int oW = A.oW[s], oH = A.oH[s], H = A.H, W = A.W;
  const int oy = jj / oW, ox = jj - oy * oW;
  const float u = (float)((double)ox * A.step_u[s]);
  const float v = (float)((double)oy * A.step_v[s]);
  const int u0 = min(max((int)floorf(u), 0), max(W - 2, 0)), v0 = min(max((int)floorf(v), 0), max(H - 2, 0));
  const int u1 = min(u0 + 1, W - 1), v1 = min(v0 + 1, H - 1);
  const float wu1 = u - (float)u0, wv1 = v - (float)v0;
  const float wu0 = 1.0f - wu1, wv0 = 1.0f - wv1;
  const float* img = A.x + (size_t)nc * H * W;
  const float top = img[v0 * W + u0] * wu0 + img[v0 * W + u1] * wu1;
  const float bot = img[v1 * W + u0] * wu0 + img[v1 * W + u1] * wu1;
  A.y[s][(size_t)nc * oH * oW + jj] = top * wv0 + bot * wv1;
}

// The same pyramid, pixel-interleaved (SFM_LAYOUT_HWC): x (N,3G,H,W) -> y[s] (N,G,h_s,w_s,3), s = 0..S-1.  One thread
// per output pixel computes the three channels with one set of weights; per channel the arithmetic is that of
// pyramid_fwd_kernel, so the values agree bit for bit.  Scale 0 is a copy.
struct __attribute__((packed, aligned(4))) Pair2 {    // two horizontally adjacent pixels of a plane
  float a, b;
};
struct __attribute__((packed, aligned(4))) Float3 {   // one pixel-interleaved texel: written with one 12-byte store
  float c[3];
};

template <bool PAIR>
__global__ void pyramid_hwc_fwd_kernel(const PyramidArgs A) {
#pragma clang fp contract(off)
  int ng = blockIdx.y;   // n * G + g
  const float* xin = A.x;
  float* const* yout = A.y;
  if (PAIR && ng >= A.n_first) { ng -= A.n_first; xin = A.x2; yout = A.y2; }   // block-uniform
  const int H = A.H, W = A.W;
  const size_t P = (size_t)H * W;
  const float* img = xin + (size_t)ng * 3 * P;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.begin[A.n_scales]) return;
  // scale 0 is a transposition (planar -> pixel-interleaved): four consecutive pixels per thread.  A block that lies wholly
  // inside scale 0 (uniform test) goes through LDS: 16-byte loads from the three planes, and 16-byte stores whose addresses
  // are consecutive ACROSS the lanes (a lane writing its own 48 bytes makes every store instruction touch each cache line
  // of the span for a third of its bytes).  Blocks at the end of the range and unaligned tensors go pixel by pixel.
  __shared__ float4 stage[3 * 256];
  const int jb = blockIdx.x * blockDim.x;   // first thread of the block
  const bool block_vec = (jb + (int)blockDim.x <= A.quads0) && ((size_t)(jb + (int)blockDim.x) * 4 <= P) && (P % 4 == 0) &&
                         blockDim.x == 256 && ((((uintptr_t)xin) | ((uintptr_t)yout[0])) % 16 == 0);
  if (block_vec) {
    const size_t p0 = (size_t)j * 4;
    const float4 r = *reinterpret_cast<const float4*>(img + p0), g = *reinterpret_cast<const float4*>(img + P + p0),
                 bl = *reinterpret_cast<const float4*>(img + 2 * P + p0);
    const int t = threadIdx.x;
    stage[3 * t + 0] = make_float4(r.x, g.x, bl.x, r.y);
    stage[3 * t + 1] = make_float4(g.y, bl.y, r.z, g.z);
    stage[3 * t + 2] = make_float4(bl.z, r.w, g.w, bl.w);
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(yout[0] + ((size_t)ng * P + (size_t)jb * 4) * 3);   // 768 float4 per block
    o[t] = stage[t];
    o[t + 256] = stage[t + 256];
    o[t + 512] = stage[t + 512];
    return;
  }
  if (j < A.quads0) {
    const size_t p0 = (size_t)j * 4;
    {
      for (size_t q = p0; q < p0 + 4 && q < P; ++q) {
        Float3 t;
#pragma unroll
        for (int c = 0; c < 3; ++c) t.c[c] = img[c * P + q];
        *reinterpret_cast<Float3*>(yout[0] + ((size_t)ng * P + q) * 3) = t;
      }
    }
    return;
  }
  int s = 1;
#pragma unroll
  for (int k = 2; k < SFM_MAX_SCALES; ++k)
    if (k < A.n_scales && j >= A.begin[k]) s = k;
  const int jj = j - A.begin[s];
  const int oW = A.oW[s], oH = A.oH[s];
  // row and column of the flat index: estimate with the reciprocal, then make it exact
  int oy, ox;
  if (jj < (1 << 22)) {   // the estimate is within one of the quotient while the index is exact in a float
    oy = (int)((float)jj * A.inv_oW[s]);
    ox = jj - oy * oW;
    if (ox < 0) { --oy; ox += oW; }
    else if (ox >= oW) { ++oy; ox -= oW; }
  } else {
    oy = jj / oW;
    ox = jj - oy * oW;
  }
  const float u = (float)((double)ox * A.step_u[s]);
  const float v = (float)((double)oy * A.step_v[s]);
  const int u0 = min(max((int)floorf(u), 0), max(W - 2, 0)), v0 = min(max((int)floorf(v), 0), max(H - 2, 0));
  const int u1 = min(u0 + 1, W - 1), v1 = min(v0 + 1, H - 1);
  const float wu1 = u - (float)u0, wv1 = v - (float)v0;
  const float wu0 = 1.0f - wu1, wv0 = 1.0f - wv1;
  Float3 t;
  if (W >= 2) {   // u1 = u0 + 1: the two taps of a row with one 8-byte load
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pl = img + c * P;
      const Pair2 a = *reinterpret_cast<const Pair2*>(pl + v0 * W + u0), b = *reinterpret_cast<const Pair2*>(pl + v1 * W + u0);
      const float top = a.a * wu0 + a.b * wu1;
      const float bot = b.a * wu0 + b.b * wu1;
      t.c[c] = top * wv0 + bot * wv1;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pl = img + c * P;
      const float top = pl[v0 * W + u0] * wu0 + pl[v0 * W + u1] * wu1;
      const float bot = pl[v1 * W + u0] * wu0 + pl[v1 * W + u1] * wu1;
      t.c[c] = top * wv0 + bot * wv1;
    }
  }
  *reinterpret_cast<Float3*>(yout[s] + ((size_t)ng * oH * oW + jj) * 3) = t;
}

// The same pyramid with every input pixel read ONCE: a workgroup owns a band of input rows of one image, brings it (plus
// one row below) into LDS with 16-byte loads, writes the band's own rows of scale 0 (the planar -> pixel-interleaved
// transposition) and every output row of the smaller scales whose upper tap row v0 lies in the band (then v0 + 1 is in LDS too).
// pyramid_hwc_fwd_kernel fetches the input once per scale -- 2.75x the image in cache lines at four scales, from other XCDs' L2.
// Per output pixel the arithmetic is that kernel's, statement by statement: the values agree bit for bit.
struct BandArgs {
  PyramidArgs P;
  int band_rows;      // input rows per band (the LDS holds band_rows + 1 rows of three planes)
  int n_bands;
};

__device__ __forceinline__ int pyr_v0(const PyramidArgs& A, const int s, const int oy) {
#pragma clang fp contract(off)
  const float v = (float)((double)oy * A.step_v[s]);
  return min(max((int)floorf(v), 0), max(A.H - 2, 0));
}

template <bool PAIR, int NT>
__global__ void __launch_bounds__(NT) pyramid_band_hwc_kernel(const BandArgs B) {
#pragma clang fp contract(off)
  extern __shared__ float band[];   // [3][rows_here][W]
  const PyramidArgs& A = B.P;
  int ng = blockIdx.y;   // n * G + g
  const float* xin = A.x;
  float* const* yout = A.y;
  if (PAIR && ng >= A.n_first) { ng -= A.n_first; xin = A.x2; yout = A.y2; }   // block-uniform
  const int H = A.H, W = A.W;
  const size_t P = (size_t)H * W;
  const float* img = xin + (size_t)ng * 3 * P;
  const int r0 = blockIdx.x * B.band_rows;
  const int own = min(B.band_rows, H - r0);             // rows of scale 0 this band writes
  const int rows = min(B.band_rows + 1, H - r0);        // rows in LDS (one more than owned, unless the image ends)
  const int span = rows * W;                            // floats per plane, contiguous in memory
  const int tid = threadIdx.x;
  // ---- load: three contiguous spans (host guarantees W % 4 == 0 and 16-byte aligned tensors for this kernel)
  {
    const int q4 = span >> 2;
    const float4* g[3];
    float4* l[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      g[c] = reinterpret_cast<const float4*>(img + c * P + (size_t)r0 * W);
      l[c] = reinterpret_cast<float4*>(band + c * span);
    }
    // (a plain copy loop: each 16 bytes are waited for before the next load goes out.  With 1, 2, 4 float4 per plane in flight per
    //  thread the pair at B=32 takes 31.8 / 43.1 / 48.7 us against 29.4 for this loop -- a workgroup that loads its band in one burst
    //  and then stores in one burst does worse than twelve waves per CU trickling; 256 threads: 128 / 512 / 1024 give 34.4 / 30.5 / 37.8)
    for (int q = tid; q < q4; q += NT) {
#pragma unroll
      for (int c = 0; c < 3; ++c) l[c][q] = g[c][q];
    }
  }
  __syncthreads();
  // ---- scale 0: own * W texels of 12 bytes, contiguous in the output; 16-byte stores whose addresses run across the lanes
  {
    float4* o = reinterpret_cast<float4*>(yout[0] + ((size_t)ng * P + (size_t)r0 * W) * 3);
    const int n4 = own * W * 3 / 4;                     // W % 4 == 0
    for (int q = tid; q < n4; q += NT) {
      const int e = q * 4;                              // element e = 3 * pixel + channel
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = (e + k) / 3, c = (e + k) - 3 * px;
        v[k] = band[c * span + px];
      }
      o[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  // ---- smaller scales: the output rows whose upper tap row lies in [r0, r0 + band_rows)
  for (int s = 1; s < A.n_scales; ++s) {
    const int oW = A.oW[s], oH = A.oH[s];
    // v0(oy) is monotone in oy: first row with v0 >= r0 and first row with v0 >= r0 + band_rows, found from an estimate
    int lo, hi;
    {
      const double sv = A.step_v[s];
      int e = sv > 0.0 ? (int)((double)r0 / sv) : 0;
      e = min(max(e - 2, 0), oH);
      while (e > 0 && pyr_v0(A, s, e - 1) >= r0) --e;    // (never taken for a down-scaling step; keeps the search exact for any step)
      while (e < oH && pyr_v0(A, s, e) < r0) ++e;
      lo = e;
      e = sv > 0.0 ? (int)((double)(r0 + B.band_rows) / sv) : oH;
      e = min(max(e - 2, lo), oH);
      while (e > lo && pyr_v0(A, s, e - 1) >= r0 + B.band_rows) --e;
      while (e < oH && pyr_v0(A, s, e) < r0 + B.band_rows) ++e;
      hi = e;
    }
    const int cnt = (hi - lo) * oW;
    for (int idx = tid; idx < cnt; idx += NT) {
      const int dy = idx / oW, ox = idx - dy * oW, oy = lo + dy;
      const float u = (float)((double)ox * A.step_u[s]);
      const float v = (float)((double)oy * A.step_v[s]);
      const int u0 = min(max((int)floorf(u), 0), max(W - 2, 0)), v0 = min(max((int)floorf(v), 0), max(H - 2, 0));
      const int v1 = min(v0 + 1, H - 1);
      const float wu1 = u - (float)u0, wv1 = v - (float)v0;
      const float wu0 = 1.0f - wu1, wv0 = 1.0f - wv1;
      Float3 t;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pl = band + c * span;
        const float a0 = pl[(v0 - r0) * W + u0], a1 = pl[(v0 - r0) * W + u0 + 1];     // (W >= 2 for this kernel: u1 = u0 + 1)
        const float b0 = pl[(v1 - r0) * W + u0], b1 = pl[(v1 - r0) * W + u0 + 1];
        const float top = a0 * wu0 + a1 * wu1;
        const float bot = b0 * wu0 + b1 * wu1;
        t.c[c] = top * wv0 + bot * wv1;
      }
      *reinterpret_cast<Float3*>(yout[s] + ((size_t)ng * oH * oW + (size_t)oy * oW + ox) * 3) = t;
    }
  }
}

// tuning overrides of the pyramid launch (development only), read from the environment ONCE per process
struct PyramidTuning {
  int band_rows = 0;     // SFM_PYRAMID_BAND_ROWS: input rows per band (LDS permitting)
  int threads = 256;     // SFM_PYRAMID_THREADS: 256 or 512
  PyramidTuning() {
    if (const char* e = getenv("SFM_PYRAMID_BAND_ROWS")) band_rows = atoi(e);
    if (const char* e = getenv("SFM_PYRAMID_THREADS")) threads = atoi(e);
  }
};
static const PyramidTuning& pyramid_tuning() { static const PyramidTuning t; return t; }

// sfm_pyramid_variant(): which kernel the NEXT pixel-interleaved pyramid call of this thread runs (then back to 0 = automatic)
static thread_local int g_pyramid_variant = 0;

// Launches the band kernel when its preconditions hold (returns false otherwise: the caller uses the per-pixel kernel).
// the hook is taken -- and forgotten -- at the TOP of the entry point it applies to, whatever becomes of the call (an empty batch, a
// rejected argument): left armed it would pick the kernel of some later, unrelated call of the thread (round-4 advisor finding)
static int take_pyramid_variant() {
  const int v = g_pyramid_variant;
  g_pyramid_variant = 0;
  return v;
}

template <bool PAIR>
static bool launch_pyramid_band(const PyramidArgs& A, int images, hipStream_t st, const int variant) {
  const int H = A.H, W = A.W;
  if (variant == 1) return false;                                          // the per-pixel kernel was asked for
  if (H < 2 || W < 4 || (W & 3)) return false;
  uintptr_t al = (uintptr_t)A.x | (uintptr_t)A.y[0];
  if (PAIR) al |= (uintptr_t)A.x2 | (uintptr_t)A.y2[0];
  if (al & 15) return false;
  const int lds_budget = 48 * 1024;                                        // three workgroups per CU
  int band_rows = lds_budget / (12 * W) - 1;
  if (band_rows < 1) return false;
  if (band_rows > 8) band_rows = 8;
  const PyramidTuning& T = pyramid_tuning();
  if (T.band_rows >= 1 && (size_t)3 * (T.band_rows + 1) * W * sizeof(float) <= 64 * 1024) band_rows = T.band_rows;
  BandArgs B;
  B.P = A;
  B.band_rows = band_rows;
  B.n_bands = (H + band_rows - 1) / band_rows;
  const size_t lds = (size_t)3 * (band_rows + 1) * W * sizeof(float);
  if (T.threads == 512) hipLaunchKernelGGL((pyramid_band_hwc_kernel<PAIR, 512>), dim3(B.n_bands, images), dim3(512), lds, st, B);
  else hipLaunchKernelGGL((pyramid_band_hwc_kernel<PAIR, 256>), dim3(B.n_bands, images), dim3(256), lds, st, B);
  return true;
}

// ------------------------------------------------------------------------------------------
// data_augmentation (datasets/kitti/kitti_raw_transformed.py:23-74) as one gather: random scaling
// (F.resize_images to (int(H*ys), int(W*xs)), :32-45), random crop back to (H, W) at (oy, ox) (:48-59)
// and horizontal flip (:62-67), for all frames of a sample with the sample's parameters.
//   params (B,5): scaled_h, scaled_w, offset_y, offset_x, flip   (integers stored as float)
// ------------------------------------------------------------------------------------------
__global__ void augment_fwd_kernel(const float* __restrict__ x, const float* __restrict__ params, float* __restrict__ y,
                                   int planes_per_sample, int H, int W) {
#pragma clang fp contract(off)
  const int plane = blockIdx.y;                       // (sample, frame, channel)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= H * W) return;
  const float* p = params + (plane / planes_per_sample) * 5;
  const int H2 = (int)p[0], W2 = (int)p[1], oy = (int)p[2], ox = (int)p[3];
  const bool flip = p[4] != 0.f;
  const int yo = j / W, xo = j - yo * W;
  const int Y = yo + oy, X = (flip ? (W - 1 - xo) : xo) + ox;          // position in the scaled image
  const float u = W2 > 1 ? (float)((double)X * ((double)(W - 1) / (double)(W2 - 1))) : 0.f;
  const float v = H2 > 1 ? (float)((double)Y * ((double)(H - 1) / (double)(H2 - 1))) : 0.f;
  const int u0 = min(max((int)floorf(u), 0), max(W - 2, 0)), v0 = min(max((int)floorf(v), 0), max(H - 2, 0));
  const int u1 = min(u0 + 1, W - 1), v1 = min(v0 + 1, H - 1);
  const float wu1 = u - (float)u0, wv1 = v - (float)v0;
  const float wu0 = 1.0f - wu1, wv0 = 1.0f - wv1;
  const float* img = x + (size_t)plane * H * W;
  const float top = img[v0 * W + u0] * wu0 + img[v0 * W + u1] * wu1;
  const float bot = img[v1 * W + u0] * wu0 + img[v1 * W + u1] * wu1;
  y[(size_t)plane * H * W + j] = top * wv0 + bot * wv1;
}

// ------------------------------------------------------------------------------------------
// DispNet's output activation at all scales in one launch (models/disp_net.py:7-8,104,110,116,122):
//   disp = DISP_SCALING * sigmoid(x) + MIN_DISP ;  g_x = g_disp * DISP_SCALING * s (1 - s), s = (disp - MIN_DISP) / DISP_SCALING
// ------------------------------------------------------------------------------------------
struct ActArgs {
  const float* a[SFM_MAX_SCALES];   // fwd: x        bwd: disp
  const float* g[SFM_MAX_SCALES];   //               bwd: g_disp
  float* o[SFM_MAX_SCALES];         // fwd: disp     bwd: g_x
  long long begin[SFM_MAX_SCALES + 1];
  int n_scales;
};
constexpr float DISP_SCALING = 10.f, MIN_DISP = 0.01f;

template <bool BWD>
__global__ void disp_act_kernel(const ActArgs A) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.begin[A.n_scales]) return;
  int s = 0;
#pragma unroll
  for (int k = 1; k < SFM_MAX_SCALES; ++k)
    if (k < A.n_scales && j >= A.begin[k]) s = k;
  const long long jj = j - A.begin[s];
  if (!BWD) {
    const float x = A.a[s][jj];
    A.o[s][jj] = DISP_SCALING * (1.0f / (1.0f + expf(-x))) + MIN_DISP;
  } else {
    const float sg = (A.a[s][jj] - MIN_DISP) * (1.0f / DISP_SCALING);
    A.o[s][jj] = A.g[s][jj] * DISP_SCALING * sg * (1.0f - sg);
  }
}

static int disp_act_launch(bool bwd, const float* const* a, const float* const* g, float* const* o, const long long* numel,
                           int n_scales, void* stream, const char* who) {
  SFM_REQUIRE(a && o && numel && (!bwd || g), SFM_ERR_NULL, "%s: NULL pointer", who);
  SFM_REQUIRE(n_scales >= 1 && n_scales <= SFM_MAX_SCALES, SFM_ERR_SHAPE, "%s: n_scales=%d", who, n_scales);
  ActArgs A;
  A.n_scales = n_scales;
  A.begin[0] = 0;
  for (int s = 0; s < n_scales; ++s) {
    SFM_REQUIRE(numel[s] >= 0, SFM_ERR_SHAPE, "%s: numel[%d] < 0", who, s);
    SFM_REQUIRE(numel[s] == 0 || (a[s] && o[s] && (!bwd || g[s])), SFM_ERR_NULL, "%s: NULL array at scale %d", who, s);
    A.a[s] = a[s]; A.g[s] = bwd ? g[s] : nullptr; A.o[s] = o[s];
    A.begin[s + 1] = A.begin[s] + numel[s];
  }
  const long long total = A.begin[n_scales];
  if (total == 0) return SFM_OK;
  SFM_REQUIRE(total < (1ll << 40), SFM_ERR_SHAPE, "%s: too large", who);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (bwd) hipLaunchKernelGGL(disp_act_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
  else hipLaunchKernelGGL(disp_act_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, A);
  return check_launch(who);
}

}  // namespace sfm

using namespace sfm;

extern "C" {

int sfm_abi_version(void) { return SFM_ABI_VERSION; }
const char* sfm_last_error(void) { return sfm::g_err; }


int sfm_pose_proj_fwd(const float* pose6, const float* K, float* proj, int N, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(pose6 && K && proj, SFM_ERR_NULL, "sfm_pose_proj_fwd: NULL pointer");
  SFM_REQUIRE(N >= 0, SFM_ERR_SHAPE, "sfm_pose_proj_fwd: N=%d", N);
  if (N == 0) return SFM_OK;
  hipLaunchKernelGGL(pose_proj_fwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose6, K, proj, N);
  return check_launch("sfm_pose_proj_fwd");
}

int sfm_pose_proj_bwd(const float* pose6, const float* K, const float* g_proj, float* d_pose6, int N, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(pose6 && K && g_proj && d_pose6, SFM_ERR_NULL, "sfm_pose_proj_bwd: NULL pointer");
  SFM_REQUIRE(N >= 0, SFM_ERR_SHAPE, "sfm_pose_proj_bwd: N=%d", N);
  if (N == 0) return SFM_OK;
  hipLaunchKernelGGL(pose_proj_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose6, K, g_proj, d_pose6, N);
  return check_launch("sfm_pose_proj_bwd");
}

static int check_warp_shape(const char* who, int N, int C, int H, int W) {
  SFM_REQUIRE(N >= 0 && N <= 65535, SFM_ERR_SHAPE, "%s: N=%d out of range [0,65535]", who, N);
  SFM_REQUIRE(C >= 1, SFM_ERR_SHAPE, "%s: C=%d", who, C);
  SFM_REQUIRE(H >= 3 && W >= 3, SFM_ERR_SHAPE, "%s: H=%d W=%d, need H,W >= 3", who, H, W);
  SFM_REQUIRE((long long)C * H * W < (1ll << 31), SFM_ERR_SHAPE, "%s: C*H*W too large", who);
  return SFM_OK;
}

int sfm_warp_fwd(const float* src, const float* depth, int depth_rows, const float* pose6, const float* K, float* warped,
                 int N, int C, int H, int W, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(src && depth && pose6 && K && warped, SFM_ERR_NULL, "sfm_warp_fwd: NULL pointer");
  if (int e = check_warp_shape("sfm_warp_fwd", N, C, H, W)) return e;
  SFM_REQUIRE(depth_rows == 1 || depth_rows == 3, SFM_ERR_SHAPE, "sfm_warp_fwd: depth_rows=%d, must be 1 or 3", depth_rows);
  dim3 grid((H * W + WARP_BLOCK - 1) / WARP_BLOCK, N);
  hipLaunchKernelGGL(warp_fwd_kernel, grid, dim3(WARP_BLOCK), 0, (hipStream_t)stream, src, depth, pose6, K, warped, C, H, W,
                     depth_rows);
  return check_launch("sfm_warp_fwd");
}

size_t sfm_warp_bwd_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const size_t nblk = ((size_t)H * W + WARP_BLOCK - 1) / WARP_BLOCK;
  return (size_t)N * nblk * 12 * sizeof(float);
}

int sfm_warp_bwd(const float* src, const float* depth, int depth_rows, const float* pose6, const float* K,
                 const float* g_warped, float* d_depth, float* d_pose6, float* d_src, void* ws, size_t ws_bytes, int N, int C,
                 int H, int W, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(src && depth && pose6 && K && g_warped && d_depth && d_pose6, SFM_ERR_NULL, "sfm_warp_bwd: NULL pointer");
  if (int e = check_warp_shape("sfm_warp_bwd", N, C, H, W)) return e;
  SFM_REQUIRE(depth_rows == 1 || depth_rows == 3, SFM_ERR_SHAPE, "sfm_warp_bwd: depth_rows=%d, must be 1 or 3", depth_rows);
  SFM_REQUIRE(ws && ws_bytes >= sfm_warp_bwd_workspace_bytes(N, H, W), SFM_ERR_WORKSPACE,
              "sfm_warp_bwd: workspace of %zu bytes needed, got %zu", sfm_warp_bwd_workspace_bytes(N, H, W), ws_bytes);
  const int nblk = (H * W + WARP_BLOCK - 1) / WARP_BLOCK;
  hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk, N), dim3(WARP_BLOCK), 0, (hipStream_t)stream, src, depth, pose6, K, g_warped,
                     d_depth, d_src, (float*)ws, C, H, W, depth_rows);
  hipLaunchKernelGGL(warp_bwd_pose_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, pose6, K, (const float*)ws, d_pose6, nblk);
  return check_launch("sfm_warp_bwd");
}

static int check_sampler_shape(const char* who, int N, int C, int H, int W, int oH, int oW) {
  SFM_REQUIRE(N >= 0 && N <= 65535, SFM_ERR_SHAPE, "%s: N=%d out of range [0,65535]", who, N);
  SFM_REQUIRE(C >= 1 && H >= 1 && W >= 1 && oH >= 0 && oW >= 0, SFM_ERR_SHAPE, "%s: bad shape C=%d H=%d W=%d oH=%d oW=%d", who,
              C, H, W, oH, oW);
  SFM_REQUIRE((long long)C * H * W < (1ll << 31) && (long long)C * oH * oW < (1ll << 31), SFM_ERR_SHAPE, "%s: too large", who);
  return SFM_OK;
}

int sfm_sampler_fwd(const float* x, const float* grid, float* y, int N, int C, int H, int W, int oH, int oW, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && grid && y, SFM_ERR_NULL, "sfm_sampler_fwd: NULL pointer");
  if (int e = check_sampler_shape("sfm_sampler_fwd", N, C, H, W, oH, oW)) return e;
  if (N == 0 || oH * oW == 0) return SFM_OK;
  hipLaunchKernelGGL(sampler_fwd_kernel, dim3((oH * oW + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, x, grid, y, C, H, W,
                     oH * oW);
  return check_launch("sfm_sampler_fwd");
}

int sfm_sampler_bwd(const float* x, const float* grid, const float* gy, float* ggrid, float* gx, int N, int C, int H, int W,
                    int oH, int oW, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && grid && gy && ggrid, SFM_ERR_NULL, "sfm_sampler_bwd: NULL pointer");
  if (int e = check_sampler_shape("sfm_sampler_bwd", N, C, H, W, oH, oW)) return e;
  if (N == 0 || oH * oW == 0) return SFM_OK;
  const int row_blocks = (oH + 4 * SAMPLER_BWD_ROWS - 1) / (4 * SAMPLER_BWD_ROWS);
  SFM_REQUIRE(N <= 65535 && row_blocks <= 65535, SFM_ERR_SHAPE, "sfm_sampler_bwd: N=%d / oH=%d exceed the grid", N, oH);
  const dim3 g((oW + 63) / 64, row_blocks, N);
  if (C <= SAMPLER_BWD_MAXC)   // the carried shares of up to four channels live in registers
    hipLaunchKernelGGL(sampler_bwd_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, x, grid, gy, ggrid, gx, C, H, W, oH, oW);
  else
    hipLaunchKernelGGL(sampler_bwd_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, x, grid, gy, ggrid, gx, C, H, W, oH, oW);
  return check_launch("sfm_sampler_bwd");
}

int sfm_sampler_interp_fwd(const float* x, const float* grid, float* y, int N, int C, int H, int W, int oH, int oW,
                           void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && grid && y, SFM_ERR_NULL, "sfm_sampler_interp_fwd: NULL pointer");
  if (int e = check_sampler_shape("sfm_sampler_interp_fwd", N, C, H, W, oH, oW)) return e;
  if (N == 0 || oH * oW == 0) return SFM_OK;
  hipLaunchKernelGGL(interp_fwd_kernel, dim3((oH * oW + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, x, grid, y, C, H, W,
                     oH * oW);
  return check_launch("sfm_sampler_interp_fwd");
}

int sfm_sampler_interp_bwd(const float* x, const float* grid, const float* gy, float* ggrid, float* gx, int N, int C, int H,
                           int W, int oH, int oW, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && grid && gy && ggrid, SFM_ERR_NULL, "sfm_sampler_interp_bwd: NULL pointer");
  if (int e = check_sampler_shape("sfm_sampler_interp_bwd", N, C, H, W, oH, oW)) return e;
  if (gx) {  // spational_transformer_sampler_interp.py:148: gx = zeros_like(x)
    hipError_t e = hipMemsetAsync(gx, 0, (size_t)N * C * H * W * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "sfm_sampler_interp_bwd: memset: %s", hipGetErrorString(e));
  }
  if (N == 0 || oH * oW == 0) return SFM_OK;
  hipLaunchKernelGGL(interp_bwd_kernel, dim3((oH * oW + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, x, grid, gy, ggrid, C,
                     H, W, oH * oW);
  return check_launch("sfm_sampler_interp_bwd");
}

int sfm_pyramid_variant(int variant) {
  SFM_REQUIRE(variant >= 0 && variant <= 1, SFM_ERR_CONFIG, "sfm_pyramid_variant: %d (0 = automatic, 1 = per-pixel kernel)", variant);
  sfm::g_pyramid_variant = variant;
  return SFM_OK;
}

int sfm_pyramid_fwd(const float* x, float* const* y, int N, int C, int H, int W, int n_scales, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && y, SFM_ERR_NULL, "sfm_pyramid_fwd: NULL pointer");
  SFM_REQUIRE(n_scales >= 1 && n_scales <= SFM_MAX_SCALES, SFM_ERR_SHAPE, "sfm_pyramid_fwd: n_scales=%d", n_scales);
  SFM_REQUIRE(N >= 0 && C >= 1 && H >= 1 && W >= 1 && (long long)N * C <= 65535, SFM_ERR_SHAPE, "sfm_pyramid_fwd: bad shape");
  if (n_scales == 1) return SFM_OK;
  PyramidArgs A;
  A.x = x; A.H = H; A.W = W; A.n_scales = n_scales;
  A.begin[0] = A.begin[1] = 0;
  for (int s = 1; s < n_scales; ++s) {
    SFM_REQUIRE(y[s], SFM_ERR_NULL, "sfm_pyramid_fwd: y[%d] is NULL", s);
    A.y[s] = y[s];
    A.oH[s] = H >> s;                                                   // H // 2**s, base_model.py:70
    A.oW[s] = W >> s;
    SFM_REQUIRE(A.oH[s] >= 1 && A.oW[s] >= 1, SFM_ERR_SHAPE, "sfm_pyramid_fwd: scale %d is empty", s);
    A.begin[s + 1] = A.begin[s] + A.oH[s] * A.oW[s];
  }
  pyramid_steps(A);
  const int total = A.begin[n_scales];
  hipLaunchKernelGGL(pyramid_fwd_kernel, dim3((total + 255) / 256, N * C), dim3(256), 0, (hipStream_t)stream, A);
  return check_launch("sfm_pyramid_fwd");
}

int sfm_pyramid_hwc_fwd(const float* x, float* const* y, int N, int G, int H, int W, int n_scales, void* stream) {
  const int variant = sfm::take_pyramid_variant();
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && y, SFM_ERR_NULL, "sfm_pyramid_hwc_fwd: NULL pointer");
  SFM_REQUIRE(n_scales >= 1 && n_scales <= SFM_MAX_SCALES, SFM_ERR_SHAPE, "sfm_pyramid_hwc_fwd: n_scales=%d", n_scales);
  SFM_REQUIRE(N >= 0 && G >= 1 && H >= 1 && W >= 1 && (long long)N * G <= 65535, SFM_ERR_SHAPE, "sfm_pyramid_hwc_fwd: bad shape");
  PyramidArgs A;
  A.x = x; A.H = H; A.W = W; A.n_scales = n_scales;
  A.begin[0] = 0;
  for (int s = 0; s < n_scales; ++s) {
    SFM_REQUIRE(y[s], SFM_ERR_NULL, "sfm_pyramid_hwc_fwd: y[%d] is NULL", s);
    A.y[s] = y[s];
    A.oH[s] = H >> s;                                                   // H // 2**s, base_model.py:70
    A.oW[s] = W >> s;
    SFM_REQUIRE(A.oH[s] >= 1 && A.oW[s] >= 1, SFM_ERR_SHAPE, "sfm_pyramid_hwc_fwd: scale %d is empty", s);
    SFM_REQUIRE((long long)A.begin[s] + (long long)A.oH[s] * A.oW[s] < (1ll << 31), SFM_ERR_SHAPE, "sfm_pyramid_hwc_fwd: image too large");
    // threads: scale 0 four pixels each, the other scales one pixel each
    A.begin[s + 1] = A.begin[s] + (s == 0 ? (A.oH[0] * A.oW[0] + 3) / 4 : A.oH[s] * A.oW[s]);
  }
  A.quads0 = A.begin[1];
  pyramid_steps(A);
  const int total = A.begin[n_scales];
  if (!launch_pyramid_band<false>(A, N * G, (hipStream_t)stream, variant))
    hipLaunchKernelGGL(pyramid_hwc_fwd_kernel<false>, dim3((total + 255) / 256, N * G), dim3(256), 0, (hipStream_t)stream, A);
  return check_launch("sfm_pyramid_hwc_fwd");
}

int sfm_pyramid_pair_hwc_fwd(const float* tgt, const float* src, float* const* y_tgt, float* const* y_src, int N, int n_src, int H,
                             int W, int n_scales, void* stream) {
  const int variant = sfm::take_pyramid_variant();
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(tgt && src && y_tgt && y_src, SFM_ERR_NULL, "sfm_pyramid_pair_hwc_fwd: NULL pointer");
  SFM_REQUIRE(n_scales >= 1 && n_scales <= SFM_MAX_SCALES, SFM_ERR_SHAPE, "sfm_pyramid_pair_hwc_fwd: n_scales=%d", n_scales);
  SFM_REQUIRE(N >= 0 && n_src >= 1 && n_src <= SFM_MAX_SRC && H >= 1 && W >= 1 && (long long)N * (1 + n_src) <= 65535, SFM_ERR_SHAPE,
              "sfm_pyramid_pair_hwc_fwd: bad shape");
  PyramidArgs A;
  A.x = tgt; A.x2 = src; A.n_first = N; A.G2 = n_src; A.H = H; A.W = W; A.n_scales = n_scales;
  A.begin[0] = 0;
  for (int s = 0; s < n_scales; ++s) {
    SFM_REQUIRE(y_tgt[s] && y_src[s], SFM_ERR_NULL, "sfm_pyramid_pair_hwc_fwd: output of scale %d is NULL", s);
    A.y[s] = y_tgt[s];
    A.y2[s] = y_src[s];
    A.oH[s] = H >> s;                                                   // H // 2**s, base_model.py:70
    A.oW[s] = W >> s;
    SFM_REQUIRE(A.oH[s] >= 1 && A.oW[s] >= 1, SFM_ERR_SHAPE, "sfm_pyramid_pair_hwc_fwd: scale %d is empty", s);
    SFM_REQUIRE((long long)A.begin[s] + (long long)A.oH[s] * A.oW[s] < (1ll << 31), SFM_ERR_SHAPE, "sfm_pyramid_pair_hwc_fwd: image too large");
    // threads: scale 0 four pixels each, the other scales one pixel each
    A.begin[s + 1] = A.begin[s] + (s == 0 ? (A.oH[0] * A.oW[0] + 3) / 4 : A.oH[s] * A.oW[s]);
  }
  A.quads0 = A.begin[1];
  pyramid_steps(A);
  const int total = A.begin[n_scales];
  if (!launch_pyramid_band<true>(A, N * (1 + n_src), (hipStream_t)stream, variant))
    hipLaunchKernelGGL(pyramid_hwc_fwd_kernel<true>, dim3((total + 255) / 256, N * (1 + n_src)), dim3(256), 0, (hipStream_t)stream, A);
  return check_launch("sfm_pyramid_pair_hwc_fwd");
}

int sfm_augment_fwd(const float* imgs, const float* params, float* out, int B, int F, int C, int H, int W, void* stream) {
  if (B == 0) return SFM_OK;
  SFM_REQUIRE(imgs && params && out, SFM_ERR_NULL, "sfm_augment_fwd: NULL pointer");
  SFM_REQUIRE(B >= 0 && F >= 1 && C >= 1 && H >= 1 && W >= 1 && (long long)B * F * C <= 65535, SFM_ERR_SHAPE,
              "sfm_augment_fwd: bad shape B=%d F=%d C=%d H=%d W=%d", B, F, C, H, W);
  hipLaunchKernelGGL(augment_fwd_kernel, dim3((H * W + 255) / 256, B * F * C), dim3(256), 0, (hipStream_t)stream, imgs, params,
                     out, F * C, H, W);
  return check_launch("sfm_augment_fwd");
}

int sfm_disp_act_fwd(const float* const* x, float* const* disp, const long long* numel, int n_scales, void* stream) {
  return disp_act_launch(false, x, nullptr, disp, numel, n_scales, stream, "sfm_disp_act_fwd");
}

int sfm_disp_act_bwd(const float* const* disp, const float* const* g_disp, float* const* g_x, const long long* numel,
                     int n_scales, void* stream) {
  return disp_act_launch(true, disp, g_disp, g_x, numel, n_scales, stream, "sfm_disp_act_bwd");
}

int sfm_resize_fwd(const float* x, float* y, int N, int C, int H, int W, int oH, int oW, void* stream) {
  if (N == 0) return SFM_OK;   // empty batch: nothing to do, pointers may be NULL
  SFM_REQUIRE(x && y, SFM_ERR_NULL, "sfm_resize_fwd: NULL pointer");
  SFM_REQUIRE(N >= 0 && C >= 1 && H >= 1 && W >= 1 && oH >= 1 && oW >= 1, SFM_ERR_SHAPE, "sfm_resize_fwd: bad shape");
  SFM_REQUIRE((long long)N * C <= 65535, SFM_ERR_SHAPE, "sfm_resize_fwd: N*C=%lld > 65535", (long long)N * C);
  if (N == 0) return SFM_OK;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3((oH * oW + 255) / 256, N * C), dim3(256), 0, (hipStream_t)stream, x, y, H, W, oH, oW);
  return check_launch("sfm_resize_fwd");
}

}  // extern "C"
