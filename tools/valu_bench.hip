// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_add_f32_dpp on gfx950, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x3}, p5 = {x5, x7}, p6 = {x0, x2}, p7 = {x4, x6};
  float2_ A = {a, a}, B = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(A), "v"(B));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %3, %4, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %4, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %5, %6, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %6, %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32_dpp %7, %0, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
  }
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (r == 12345.678f) out[0] = r;
}
// DPP semantics check: lane l must get lane l-1 (wave_shr) / l+1 (wave_shl) across all 64 lanes
__global__ void dppcheck(int* out) {
  int v = threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xf, 0xf, false);
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-7, v, 0x130, 0xf, 0xf, false);
  out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}
template <int MODE>
void run(const char* name, int waves_per_simd, float* d) {
  const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double inst = (double)iters * 64.0 * waves_per_simd;   // wave-instructions per SIMD
  printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  int* di; hipMalloc(&di, 192 * 4);
  hipLaunchKernelGGL(dppcheck, dim3(1), dim3(64), 0, 0, di);
  int h[192]; hipMemcpy(h, di, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) { if (l > 0 && h[l] != l - 1) ok = 0; if (l < 63 && h[64 + l] != l + 1) ok = 0; }
  printf("dpp wave_shr/wave_shl semantic ok=%d ; lane0 shr (old=-7, no bound_ctrl)=%d ; lane63 shl=%d ; lane0 shr old=0: %d\n", ok, h[0], h[127], h[128]);
  for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w, d); run<1>("v_pk_fma_f32", w, d); run<2>("v_add_dpp", w, d); }
  return 0;
}
