// Issue cost of the instruction kinds of the row step on gfx950, at k resident waves per SIMD: s_memtime around an unrolled
// stream of 8 independent instances of one instruction.  hipcc --offload-arch=gfx950 tools/op_cost.hip -o tools/op_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
#define V8 "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
#define OP8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)

template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters, float a, float b, int sa, int* flag) {
  __shared__ float lds[64 * 8];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, A = {a, a};
  unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;
  unsigned la = threadIdx.x * 4;
  for (int i = 0; i < 8; ++i) lds[threadIdx.x + 64 * i] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#define F_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define F_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define F_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define F_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define F_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define F_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define F_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define F_BITOP(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x6c\n"
#define F_CLAMP(n) "v_max_f32_e64 %" #n ", %" #n ", %" #n " clamp\n"
#define F_CNDS(n) "v_cndmask_b32_e64 %" #n ", 0, %" #n ", s[20:21]\n"
#define F_CMPS(n) "v_cmp_lt_f32_e64 s[22:23], %" #n ", %8\n"
#define F_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define F_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define F_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define F_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define F_MULCL(n) "v_mul_f32_e64 %" #n ", %" #n ", %8 clamp\n"
#define F_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define F_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define F_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define F_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define F_FRACT(n) "v_fract_f32 %" #n ", %" #n "\n"
#define F_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define F_LSHL(n) "v_lshlrev_b32 %" #n ", 2, %" #n "\n"
#define F_CMPV(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define F_CNDV(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define F_ADDDPP(n) "v_add_f32_dpp %" #n ", %8, %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define F_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define F_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define F_SUBABS(n) "v_sub_f32_e64 %" #n ", |%" #n "|, %8\n"
#define F_FMAK(n) "v_fmac_f32 %" #n ", %8, %9\n"
    if (MODE == 0) { REP8(asm volatile(OP8(F_FMA) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 1) { REP8(asm volatile(OP8(F_MUL) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 2) { REP8(asm volatile(OP8(F_MOV) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 3) { REP8(asm volatile(OP8(F_CVTI) OP8(F_CVTF) : V8 : "v"(a), "v"(b));) }     // 16 instructions
    else if (MODE == 4) { REP8(asm volatile(OP8(F_FLOOR) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 5) { REP8(asm volatile(OP8(F_BITOP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 6) { REP8(asm volatile(OP8(F_CLAMP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 7) { REP8(asm volatile(OP8(F_CNDS) : V8 : "v"(a), "v"(b) : "s20", "s21");) }
    else if (MODE == 8) { REP8(asm volatile(OP8(F_CMPS) : V8 : "v"(a), "v"(b) : "s22", "s23");) }
    else if (MODE == 9) { REP8(asm volatile(OP8(F_MULLO) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 10) { REP8(asm volatile(OP8(F_MAD24) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 11) { REP8(asm volatile(OP8(F_LSHLADD) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 12) {   // v_mad_u64_u32 (4 independent 64-bit accumulators, 8 instructions)
      REP8(asm volatile("v_mad_u64_u32 %0, s[22:23], %4, %5, %0\n v_mad_u64_u32 %1, s[22:23], %4, %5, %1\n v_mad_u64_u32 %2, s[22:23], %4, %5, %2\n v_mad_u64_u32 %3, s[22:23], %4, %5, %3\n"
                        "v_mad_u64_u32 %0, s[22:23], %4, %5, %0\n v_mad_u64_u32 %1, s[22:23], %4, %5, %1\n v_mad_u64_u32 %2, s[22:23], %4, %5, %2\n v_mad_u64_u32 %3, s[22:23], %4, %5, %3\n"
                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(sa), "v"(la) : "s22", "s23");)
    } else if (MODE == 13) {   // v_lshl_add_u64
      REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                        "v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0 + 5));)
    } else if (MODE == 14) {   // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                        "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));)
    } else if (MODE == 15) {   // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));)
    } else if (MODE == 16) {   // ds_write_b32 (8 per group, then one wait)
      REP8(asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:256\n ds_write_b32 %8, %2 offset:512\n ds_write_b32 %8, %3 offset:768\n"
                        "ds_write_b32 %8, %4 offset:1024\n ds_write_b32 %8, %5 offset:1280\n ds_write_b32 %8, %6 offset:1536\n ds_write_b32 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        : V8 : "v"(la) : "memory");)
    } else if (MODE == 17) {   // ds_add_f32 without return
      REP8(asm volatile("ds_add_f32 %8, %0\n ds_add_f32 %8, %1 offset:256\n ds_add_f32 %8, %2 offset:512\n ds_add_f32 %8, %3 offset:768\n"
                        "ds_add_f32 %8, %4 offset:1024\n ds_add_f32 %8, %5 offset:1280\n ds_add_f32 %8, %6 offset:1536\n ds_add_f32 %8, %7 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        : V8 : "v"(la) : "memory");)
    } else if (MODE == 18) {   // ds_read_b32
      REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                        "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                        : V8 : "v"(la) : "memory");)
    } else if (MODE == 19) {   // scalar adds
      REP8(asm volatile("s_add_i32 s20, s20, 1\n s_add_i32 s21, s21, 1\n s_add_i32 s22, s22, 1\n s_add_i32 s23, s23, 1\n s_add_i32 s20, s20, 1\n s_add_i32 s21, s21, 1\n s_add_i32 s22, s22, 1\n s_add_i32 s23, s23, 1\n" ::: "s20", "s21", "s22", "s23", "scc");)
    } else if (MODE == 20) {   // compare + branch not taken
      REP8(asm volatile("s_cmp_lt_i32 %0, 0\n s_cbranch_scc1 1f\n s_cmp_lt_i32 %0, 0\n s_cbranch_scc1 1f\n s_cmp_lt_i32 %0, 0\n s_cbranch_scc1 1f\n s_cmp_lt_i32 %0, 0\n s_cbranch_scc1 1f\n1:\n" :: "s"(sa) : "scc");)
    } else if (MODE == 21) {   // s_nop 0
      REP8(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");)
    } else if (MODE == 22) { REP8(asm volatile(OP8(F_RCP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 23) { REP8(asm volatile(OP8(F_ADD) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 24) { REP8(asm volatile(OP8(F_MULCL) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 25) { REP8(asm volatile(OP8(F_MED3) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 26) { REP8(asm volatile(OP8(F_MAX) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 27) { REP8(asm volatile(OP8(F_AND) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 28) { REP8(asm volatile(OP8(F_BFI) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 29) { REP8(asm volatile(OP8(F_FRACT) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 30) { REP8(asm volatile(OP8(F_ADDU) : V8 : "v"(sa), "v"(b));) }
    else if (MODE == 31) { REP8(asm volatile(OP8(F_LSHL) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 32) { REP8(asm volatile(OP8(F_CMPV) : V8 : "v"(a), "v"(b) : "vcc");) }
    else if (MODE == 33) { REP8(asm volatile(OP8(F_CNDV) : V8 : "v"(a), "v"(b) : "vcc");) }
    else if (MODE == 34) { REP8(asm volatile(OP8(F_ADDDPP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 35) { REP8(asm volatile(OP8(F_MOVDPP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 36) { REP8(asm volatile(OP8(F_EXP) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 37) { REP8(asm volatile(OP8(F_SUBABS) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 38) { REP8(asm volatile(OP8(F_FMAK) : V8 : "v"(a), "v"(b));) }
    else if (MODE == 41) {   // compare into VCC + select on VCC (VOP2 forms: what the compiler emits for `c ? a : b`)
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %8, %1, vcc\n"
                        "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %8, %2, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %8, %3, vcc\n"
                        : V8 : "v"(a), "v"(b) : "vcc");)
    } else if (MODE == 42) {   // the same through an SGPR pair (VOP3 forms)
      REP8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %8, %0, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %1, %8\n v_cndmask_b32_e64 %1, %8, %1, s[22:23]\n"
                        "v_cmp_lt_f32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %8, %2, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %3, %8\n v_cndmask_b32_e64 %3, %8, %3, s[22:23]\n"
                        : V8 : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");)
    } else if (MODE == 43) {   // select on VCC in the VOP3 encoding, VCC written once
      REP8(asm volatile("v_cndmask_b32_e64 %0, %8, %0, vcc\n v_cndmask_b32_e64 %1, %8, %1, vcc\n v_cndmask_b32_e64 %2, %8, %2, vcc\n v_cndmask_b32_e64 %3, %8, %3, vcc\n"
                        "v_cndmask_b32_e64 %4, %8, %4, vcc\n v_cndmask_b32_e64 %5, %8, %5, vcc\n v_cndmask_b32_e64 %6, %8, %6, vcc\n v_cndmask_b32_e64 %7, %8, %7, vcc\n"
                        : V8 : "v"(a), "v"(b) : "vcc");)
    } else if (MODE == 44) {   // two compares first, then the two selects (distance 2 between writer and reader)
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32_e64 s[20:21], %1, %8\n v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32_e64 %1, %8, %1, s[20:21]\n"
                        "v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32_e64 s[20:21], %3, %8\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32_e64 %3, %8, %3, s[20:21]\n"
                        : V8 : "v"(a), "v"(b) : "vcc", "s20", "s21");)
    } else if (MODE == 39) {   // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));)
    } else if (MODE == 40) {   // v_pk_mul_f32 clamp
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %4 clamp\n v_pk_mul_f32 %1, %1, %4 clamp\n v_pk_mul_f32 %2, %2, %4 clamp\n v_pk_mul_f32 %3, %3, %4 clamp\n"
                        "v_pk_mul_f32 %0, %0, %4 clamp\n v_pk_mul_f32 %1, %1, %4 clamp\n v_pk_mul_f32 %2, %2, %4 clamp\n v_pk_mul_f32 %3, %3, %4 clamp\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A));)
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + (float)(q0 + q1 + q2 + q3) + lds[threadIdx.x];
  if (r == 12345.678f) flag[0] = 1;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_group, unsigned long long* d, int* flag) {
  const int iters = 100;
  printf("%-34s", name);
  for (int w : {1, 3, 4}) {
    const int blocks = 256 * 4 * w;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, 3, flag);
    hipDeviceSynchronize();
    static unsigned long long h[8192];
    hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double per_wave = s / blocks / (iters * 8.0 * per_group);
    printf("  %dw: %5.2f /wave %5.2f /SIMD", w, per_wave, per_wave / w);
  }
  printf("\n");
}

int main() {
  unsigned long long* d;
  int* flag;
  hipMalloc(&d, 8192 * 8);
  hipMalloc(&flag, 4);
  printf("cycles per instruction (s_memtime ticks), 1 / 3 / 4 resident waves per SIMD\n");
  run<0>("v_fma_f32", 8, d, flag);
  run<1>("v_mul_f32", 8, d, flag);
  run<23>("v_add_f32", 8, d, flag);
  run<2>("v_mov_b32", 8, d, flag);
  run<3>("v_cvt_i32_f32 + v_cvt_f32_i32", 16, d, flag);
  run<4>("v_floor_f32", 8, d, flag);
  run<5>("v_bitop3_b32", 8, d, flag);
  run<6>("v_max_f32 clamp", 8, d, flag);
  run<7>("v_cndmask_b32_e64 (sgpr mask)", 8, d, flag);
  run<8>("v_cmp_lt_f32_e64 -> sgpr", 8, d, flag);
  run<9>("v_mul_lo_u32", 8, d, flag);
  run<10>("v_mad_u32_u24", 8, d, flag);
  run<11>("v_lshl_add_u32", 8, d, flag);
  run<12>("v_mad_u64_u32", 8, d, flag);
  run<13>("v_lshl_add_u64", 8, d, flag);
  run<14>("v_pk_mul_f32", 8, d, flag);
  run<15>("v_pk_add_f32", 8, d, flag);
  run<22>("v_rcp_f32", 8, d, flag);
  run<24>("v_mul_f32 clamp", 8, d, flag);
  run<25>("v_med3_f32", 8, d, flag);
  run<26>("v_max_f32", 8, d, flag);
  run<27>("v_and_b32", 8, d, flag);
  run<28>("v_bfi_b32", 8, d, flag);
  run<29>("v_fract_f32", 8, d, flag);
  run<30>("v_add_u32", 8, d, flag);
  run<31>("v_lshlrev_b32", 8, d, flag);
  run<32>("v_cmp_lt_f32 -> vcc", 8, d, flag);
  run<33>("v_cndmask_b32 (vcc)", 8, d, flag);
  run<43>("v_cndmask_b32_e64 (vcc)", 8, d, flag);
  run<41>("v_cmp -> vcc ; v_cndmask vcc (x4)", 8, d, flag);
  run<42>("v_cmp -> sgpr ; v_cndmask sgpr (x4)", 8, d, flag);
  run<44>("2 cmp ; 2 cndmask (x2)", 8, d, flag);
  run<34>("v_add_f32_dpp wave_shr:1", 8, d, flag);
  run<35>("v_mov_b32_dpp wave_shr:1", 8, d, flag);
  run<36>("v_exp_f32", 8, d, flag);
  run<37>("v_sub_f32 |a|", 8, d, flag);
  run<38>("v_fmac_f32", 8, d, flag);
  run<39>("v_pk_fma_f32", 8, d, flag);
  run<40>("v_pk_mul_f32 clamp", 8, d, flag);
  run<16>("ds_write_b32 (+1 wait per 8)", 8, d, flag);
  run<17>("ds_add_f32 no return (+1 wait per 8)", 8, d, flag);
  run<18>("ds_read_b32 (+1 wait per 8)", 8, d, flag);
  run<19>("s_add_i32", 8, d, flag);
  run<20>("s_cmp + s_cbranch (not taken)", 8, d, flag);
  run<21>("s_nop 0", 8, d, flag);
  return 0;
}
