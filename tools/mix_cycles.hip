// What does a realistic fp32 VALU mix cost per instruction on gfx950?  Runs SSIM-like arithmetic on
// register-resident data (no memory traffic) and reports cycles per executed VALU instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float from_left(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float from_right(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ float hsum3(float x) { asm volatile("" : "+v"(x)); return (x + from_left(x)) + from_right(x); }
template <int MODE, int PAD>
__global__ void __launch_bounds__(64) k(unsigned long long* out, float* sink, int iters, float seed) {
  float s0[3], s1[3], s2[3], t0[3], t1[3], t2[3], acc[3] = {0, 0, 0};
  float pad[PAD > 0 ? PAD : 1];
#pragma unroll
  for (int i = 0; i < (PAD > 0 ? PAD : 1); ++i) pad[i] = seed * (i + 1) + threadIdx.x;
#pragma unroll
  for (int c = 0; c < 3; ++c) { s0[c] = seed + c + threadIdx.x; s1[c] = seed * 2 + c; s2[c] = seed * 3 + c; t0[c] = seed * 0.5f + c; t1[c] = seed * 0.25f + c; t2[c] = seed * 0.125f + c; }
  unsigned long long ta, tb;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float Sx, Sy, Sxx, Syy, Sxy;
      if (MODE == 0) {   // with DPP horizontal sums
        Sx = hsum3(s2[c] + s1[c] + s0[c]); Sy = hsum3(t2[c] + t1[c] + t0[c]);
        Sxx = hsum3(fmaf(s2[c], s2[c], fmaf(s1[c], s1[c], s0[c] * s0[c])));
        Syy = hsum3(fmaf(t2[c], t2[c], fmaf(t1[c], t1[c], t0[c] * t0[c])));
        Sxy = hsum3(fmaf(s2[c], t2[c], fmaf(s1[c], t1[c], s0[c] * t0[c])));
      } else {           // same arithmetic without cross-lane ops
        Sx = (s2[c] + s1[c] + s0[c]) * 3.f; Sy = (t2[c] + t1[c] + t0[c]) * 3.f;
        Sxx = fmaf(s2[c], s2[c], fmaf(s1[c], s1[c], s0[c] * s0[c])) * 3.f;
        Syy = fmaf(t2[c], t2[c], fmaf(t1[c], t1[c], t0[c] * t0[c])) * 3.f;
        Sxy = fmaf(s2[c], t2[c], fmaf(s1[c], t1[c], s0[c] * t0[c])) * 3.f;
      }
      const float C1 = 0.0081f, C2 = 0.0729f;
      const float pxy = Sx * Sy, sq = fmaf(Sx, Sx, Sy * Sy);
      const float N1 = fmaf(2.f, pxy, C1), N2 = fmaf(-2.f, pxy, fmaf(18.f, Sxy, C2));
      const float D1 = sq + C1, D2 = fmaf(9.f, Sxx + Syy, C2) - sq;
      const float rD = __builtin_amdgcn_rcpf(D1 * D2);
      const float Sv = N1 * N2 * rD;
      const float e = fmaf(-0.5f, Sv, 0.5f);
      const float kap = (e > 0.f && e < 1.f) ? 0.37f * rD : 0.f;
      const float u3 = fmaf(-(Sv * Sx), D2 - D1, Sy * (N2 - N1));
      acc[c] += 2.f * kap * u3 - 9.f * kap * Sv * D1 + 18.f * kap * N1;
      // rotate the "ring" so that nothing is loop invariant
      const float n = acc[c] * 1e-6f + s0[c];
      s2[c] = s1[c]; s1[c] = s0[c]; s0[c] = n; t2[c] = t1[c]; t1[c] = t0[c]; t0[c] = n * 0.5f;
    }
    if (PAD > 0) {
#pragma unroll
      for (int i = 0; i < PAD; ++i) pad[i] = fmaf(pad[i], 1.0001f, acc[i % 3]);
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
  float r = acc[0] + acc[1] + acc[2];
#pragma unroll
  for (int i = 0; i < (PAD > 0 ? PAD : 1); ++i) r += pad[i];
  if (r == 12345.678f) sink[0] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = tb - ta;
}
template <int MODE, int PAD>
void run(const char* name, unsigned long long* d, float* sink) {
  const int iters = 300;
  for (int w : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL((k<MODE, PAD>), dim3(blocks), dim3(64), 0, 0, d, sink, iters, 1.37f);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    static unsigned long long h[8192];
    hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    printf("%-28s waves/SIMD=%d : %.0f cycles per loop iteration per wave -> %.0f per SIMD\n", name, w, s / blocks / iters, s / blocks / iters / w);
  }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8192 * 8);
  float* sink; hipMalloc(&sink, 4);
  run<0, 0>("ssim mix, DPP", d, sink);
  run<1, 0>("ssim mix, no DPP", d, sink);
  run<0, 120>("ssim mix, DPP, +120 VGPR fma", d, sink);
  return 0;
}
