// What an LDS atomic costs on gfx950, by kind and by how many lanes of the instruction share an address: s_memtime around an
// unrolled stream of 8 atomics without return value (one wait per 8), on every SIMD of the chip at 1 / 2 / 4 resident waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_atomic_cost.hip -o tools/lds_atomic_cost
// The question behind it (profiles/r06_d_src.txt): ds_add_f32 is unusable for the dL/d(src) window (profiles/r06_op_cost_microbench.txt:
// 768 ticks per wave instruction); are the INTEGER adds (a fixed-point window would use them) full rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X

// MODE: 0 ds_add_u32, 1 ds_add_u64, 2 ds_add_f32, 3 ds_add_rtn_u32, 4 ds_add_f64, 5 ds_write_b32, 6 ds_write_b64, 7 ds_add_rtn_u64
// PAT:  0 every lane its own address, 1 lanes 2k and 2k+1 share one, 2 groups of 8 lanes share one, 3 all 64 lanes on one,
//       4 every lane its own address, scattered (odd multiplier: bank conflicts as a hashed scatter has them)
template <int MODE, int PAT>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters, int* flag) {
  __shared__ unsigned long long lds[64 * 8 + 8];
  const unsigned l = threadIdx.x;
  for (int i = 0; i < 8; ++i) lds[l + 64 * i] = 0;
  unsigned idx = PAT == 0 ? l : PAT == 1 ? (l >> 1) : PAT == 2 ? (l >> 3) : PAT == 3 ? 0 : ((l * 37u) & 63u);
  unsigned la = idx * 8;            // byte address; 8 bytes per slot in every mode (the 32-bit modes use the low word)
  unsigned v0 = l + 1, v1 = l + 2;  // the 64-bit operand lives in a register pair
  unsigned long long v = ((unsigned long long)v1 << 32) | v0;
  float f = 1.0f + l;
  double d = 1.0 + l;
  unsigned r0 = 0;
  unsigned long long r1 = 0;
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#define EIGHT(op, val) \
  asm volatile(op " %0, %1\n " op " %0, %1 offset:512\n " op " %0, %1 offset:1024\n " op " %0, %1 offset:1536\n " \
               op " %0, %1 offset:2048\n " op " %0, %1 offset:2560\n " op " %0, %1 offset:3072\n " op " %0, %1 offset:3584\n s_waitcnt lgkmcnt(0)\n" \
               :: "v"(la), "v"(val) : "memory");
    if (MODE == 0) { EIGHT("ds_add_u32", v0) }
    else if (MODE == 1) { EIGHT("ds_add_u64", v) }
    else if (MODE == 2) { EIGHT("ds_add_f32", f) }
    else if (MODE == 4) { EIGHT("ds_add_f64", d) }
    else if (MODE == 5) { EIGHT("ds_write_b32", v0) }
    else if (MODE == 6) { EIGHT("ds_write_b64", v) }
    else if (MODE == 3) {
      asm volatile("ds_add_rtn_u32 %0, %1, %2\n ds_add_rtn_u32 %0, %1, %2 offset:512\n ds_add_rtn_u32 %0, %1, %2 offset:1024\n ds_add_rtn_u32 %0, %1, %2 offset:1536\n"
                   "ds_add_rtn_u32 %0, %1, %2 offset:2048\n ds_add_rtn_u32 %0, %1, %2 offset:2560\n ds_add_rtn_u32 %0, %1, %2 offset:3072\n ds_add_rtn_u32 %0, %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                   : "+v"(r0) : "v"(la), "v"(v0) : "memory");
    } else if (MODE == 7) {
      asm volatile("ds_add_rtn_u64 %0, %1, %2\n ds_add_rtn_u64 %0, %1, %2 offset:512\n ds_add_rtn_u64 %0, %1, %2 offset:1024\n ds_add_rtn_u64 %0, %1, %2 offset:1536\n"
                   "ds_add_rtn_u64 %0, %1, %2 offset:2048\n ds_add_rtn_u64 %0, %1, %2 offset:2560\n ds_add_rtn_u64 %0, %1, %2 offset:3072\n ds_add_rtn_u64 %0, %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                   : "+v"(r1) : "v"(la), "v"(v) : "memory");
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  __syncthreads();
  // the sums are checked where the result is known: u32 / u64 adds of lane constants
  unsigned long long got = lds[l];
  if (got == 0x123456789abcdefull || r0 == 0xdeadbeefu || r1 == 0xdeadbeefull) flag[0] = 1;
  if (MODE == 1 && PAT == 0 && got != v * 8ull * 0 + v * (unsigned long long)iters) flag[1] = 1;
  if (MODE == 1 && PAT == 3 && l == 0) {
    unsigned long long want = 0;
    for (unsigned j = 0; j < 64; ++j) want += (((unsigned long long)(j + 2) << 32) | (j + 1));
    if (got != want * (unsigned long long)iters) flag[1] = 1;
  }
  if (l == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int PAT>
void run(const char* name, unsigned long long* d, int* flag) {
  const int iters = 50;
  printf("%-46s", name);
  for (int w : {1, 2, 4}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(64), 0, 0, d, iters, flag);
    hipDeviceSynchronize();
    static unsigned long long h[8192];
    hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double per_wave = s / blocks / (iters * 8.0);
    printf("  %dw: %7.2f /wave %7.2f /SIMD %6.2f /CU", w, per_wave, per_wave / w, per_wave / w / 4);
  }
  printf("\n");
}

#define ROWS(MODE, name) \
  run<MODE, 0>(name ", 64 addresses", d, flag); run<MODE, 4>(name ", 64 addresses, scattered", d, flag); run<MODE, 1>(name ", lane pairs share", d, flag); \
  run<MODE, 2>(name ", groups of 8 share", d, flag); run<MODE, 3>(name ", all 64 on one", d, flag);

int main() {
  unsigned long long* d;
  int* flag;
  hipMalloc(&d, 8192 * 8);
  hipMalloc(&flag, 8);
  hipMemset(flag, 0, 8);
  printf("s_memtime ticks per LDS instruction (8 in flight, then one wait): per wave, per SIMD (= / waves per SIMD), per CU (= / 4 SIMDs)\n");
  ROWS(5, "ds_write_b32")
  ROWS(6, "ds_write_b64")
  ROWS(0, "ds_add_u32")
  ROWS(1, "ds_add_u64")
  ROWS(3, "ds_add_rtn_u32")
  ROWS(7, "ds_add_rtn_u64")
  ROWS(2, "ds_add_f32")
  ROWS(4, "ds_add_f64")
  int hf[2];
  hipMemcpy(hf, flag, 8, hipMemcpyDeviceToHost);
  printf("u64 sums checked: %s\n", hf[1] ? "WRONG" : "right");
  return 0;
}
