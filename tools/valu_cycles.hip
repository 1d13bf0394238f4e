// Cycle-exact VALU issue cost on gfx950: s_memtime around an unrolled instruction stream, k waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters, float a, float b, int* flag) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // independent FMAs
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 1) {   // DPP adds (wave_shr), independent
      REP8(asm volatile("v_add_f32_dpp %0, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %1, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %2, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %3, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %4, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %5, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %6, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %7, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (MODE == 2) {   // row_shr DPP (within 16 lanes)
      REP8(asm volatile("v_add_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));)
    } else if (MODE == 3) {   // v_rcp
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (MODE == 4) {   // v_cndmask with vcc
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
    } else if (MODE == 5) {   // dependent FMA chain
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        : "+v"(x0) : "v"(a), "v"(b));)
    } else if (MODE == 6) {   // v_pk_fma_f32
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, A = {a, a}, B = {b, b};
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A), "v"(B));)
      x0 = p0.x + p1.y + p2.x + p3.y;
    } else if (MODE == 7) {   // v_cmp to sgpr pair + cndmask
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");)
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (r == 12345.678f) flag[0] = 1;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, unsigned long long* d, int* flag) {
  const int iters = 200;
  for (int w : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, flag);
    hipDeviceSynchronize();
    static unsigned long long h[8192];
    hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double per_wave = s / blocks / (iters * 64.0);
    printf("%-14s waves/SIMD=%d : %.2f cycles per instr per wave  -> %.2f cycles per instr per SIMD\n", name, w, per_wave, per_wave / w);
  }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8192 * 8);
  int* flag; hipMalloc(&flag, 4);
  run<0>("v_fma indep", d, flag);
  run<5>("v_fma chain", d, flag);
  run<6>("v_pk_fma", d, flag);
  run<1>("dpp wave_shr", d, flag);
  run<2>("dpp row_shr", d, flag);
  run<3>("v_rcp", d, flag);
  run<4>("v_cndmask", d, flag);
  run<7>("cmp+cndmask", d, flag);
  return 0;
}
