// What does a hazard s_nop in front of a DPP read cost when 1-4 waves share a SIMD?
//   A: 4 x { v_add ; s_nop 1 ; v_add_dpp (reads the fresh value through DPP) ; v_add_dpp }   (what the compiler emits today)
//   B: v_add x4 ; v_add_dpp x4 ; v_add_dpp x4                                               (same work, hand-interleaved, no nop needed)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define DPPR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define DPPL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long* out, float* sink, int iters, float seed) {
  float x0 = seed + threadIdx.x, x1 = seed * 2, x2 = seed * 3, x3 = seed * 4, t0, t1, t2, t3, c = 1e-3f;
  unsigned long long ta, tb;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0)
        asm volatile(
            "v_mul_f32 %0, %0, %8\n s_nop 1\n v_add_f32_dpp %4, %0, %0" DPPR "v_add_f32_dpp %0, %0, %4" DPPL
            "v_mul_f32 %1, %1, %8\n s_nop 1\n v_add_f32_dpp %5, %1, %1" DPPR "v_add_f32_dpp %1, %1, %5" DPPL
            "v_mul_f32 %2, %2, %8\n s_nop 1\n v_add_f32_dpp %6, %2, %2" DPPR "v_add_f32_dpp %2, %2, %6" DPPL
            "v_mul_f32 %3, %3, %8\n s_nop 1\n v_add_f32_dpp %7, %3, %3" DPPR "v_add_f32_dpp %3, %3, %7" DPPL
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(c));
      else
        asm volatile(
            "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
            "v_add_f32_dpp %4, %0, %0" DPPR "v_add_f32_dpp %5, %1, %1" DPPR "v_add_f32_dpp %6, %2, %2" DPPR "v_add_f32_dpp %7, %3, %3" DPPR
            "v_add_f32_dpp %0, %0, %4" DPPL "v_add_f32_dpp %1, %1, %5" DPPL "v_add_f32_dpp %2, %2, %6" DPPL "v_add_f32_dpp %3, %3, %7" DPPL
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(c));
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
  float r = x0 + x1 + x2 + x3;
  if (r == 12345.678f) sink[0] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = tb - ta;
}
template <int MODE> void run(const char* name, unsigned long long* d, float* sink) {
  const int iters = 200;
  for (int w : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, sink, iters, 1.37f);
    hipDeviceSynchronize();
    static unsigned long long h[8192];
    hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    printf("%-28s waves/SIMD=%d : %.2f cycles per VALU instr per SIMD (48 VALU/iter)\n", name, w, s / blocks / iters / w / 48.0);
  }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8192 * 8); float* sink; hipMalloc(&sink, 4);
  run<0>("A: nop before each DPP pair", d, sink);
  run<1>("B: interleaved, no nops", d, sink);
  return 0;
}
