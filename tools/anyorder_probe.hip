// Can a small kernel that FOLLOWS a long one in the same HIP stream start before the long one has ended (hipExtAnyOrderLaunch: the
// AQL packet without the barrier bit), and see the long kernel's per-workgroup tokens as they are written?  Question behind it: the
// end of a step (finalize_kernel) polling the main launch's partials instead of waiting behind a launch boundary.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe && tools/anyorder_probe
// Kernel A: 3072 one-wave workgroups holding 160 VGPRs (three per SIMD, like the main launch), each busy for ~`busy_us`, then one
// 8-byte write-through store {value, token} per workgroup.  Kernel B: `nb` one-wave workgroups; stamps its start (s_memrealtime, the
// 100 MHz wall clock all kernels share), polls A's records with agent-scope loads until every token matches (bounded: gives up after
// 20 ms), stamps the time, sums the values, writes one result.  Printed: B's start and B's "seen" relative to A's last end, the time of
// a step (A + B back to back, 200 steps) with B in order and with B any-order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

struct Rec { float v; unsigned tok; };

__global__ void __launch_bounds__(64) kern_a(Rec* rec, unsigned long long* tend, unsigned long long* tbeg, int busy_ticks, unsigned tok, float x) {
  asm volatile("" ::: "v159");   // 160 VGPRs: three waves per SIMD
  const unsigned long long t0 = realtime();
  float a = x + threadIdx.x;
  // staggered ends, like the main launch (the last waves end over ~10 % of its duration)
  const int mine = busy_ticks - (int)((blockIdx.x * 2654435761u >> 22) % (unsigned)(busy_ticks / 8 + 1));
  while ((long long)(realtime() - t0) < mine) {
#pragma unroll
    for (int i = 0; i < 64; ++i) a = fmaf(a, 1.0000001f, 0.5f);
  }
  if (threadIdx.x == 0) {
    Rec r; r.v = (float)(blockIdx.x & 7) + (a == 12345.f ? 1.f : 0.f); r.tok = tok;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(rec + blockIdx.x), __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t1 = realtime();
    tend[blockIdx.x] = t1;      // (per-workgroup stamps: 6144 same-address 64-bit atomics took 29 us to drain in the first version)
    tbeg[blockIdx.x] = t0;
  }
}

template <int VG>
__global__ void __launch_bounds__(64) kern_b(const Rec* rec, int n, unsigned tok, unsigned long long* out /* [nb][4] */, float* sum) {
  if (VG > 32) asm volatile("" ::: "v63");
  const unsigned long long t0 = realtime();
  const int per = (n + gridDim.x - 1) / gridDim.x;
  const int i0 = blockIdx.x * per, i1 = min(n, i0 + per);
  float s = 0.f;
  unsigned long long tseen = 0;
  int polls = 0;
  bool gave_up = false;
  for (;;) {
    bool ok = true;
    s = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 64) {
      const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rec + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const Rec r = __builtin_bit_cast(Rec, w);
      ok = ok && (r.tok == tok);
      s += r.v;
    }
    ++polls;
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
    if (realtime() - t0 > 2000000ull) { gave_up = true; break; }   // 20 ms
    __builtin_amdgcn_s_sleep(8);
  }
  tseen = realtime();
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = tseen;
    out[blockIdx.x * 4 + 2] = (unsigned long long)polls;
    out[blockIdx.x * 4 + 3] = gave_up ? 1ull : 0ull;
    sum[blockIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int n = 3072, nb = argc > 1 ? atoi(argv[1]) : 65;
  const int busy_us = argc > 2 ? atoi(argv[2]) : 50;
  Rec* rec; unsigned long long *tend, *tbeg, *out; float* sum;
  CK(hipMalloc(&rec, n * sizeof(Rec))); CK(hipMemset(rec, 0, n * sizeof(Rec)));
  CK(hipMalloc(&tend, 8 * n)); CK(hipMalloc(&tbeg, 8 * n)); CK(hipMalloc(&out, nb * 32)); CK(hipMalloc(&sum, nb * 4));
  hipStream_t st, st2; CK(hipStreamCreate(&st)); CK(hipStreamCreate(&st2));
  std::vector<unsigned long long> h(nb * 4);
  std::vector<float> hs(nb);
  unsigned tok = 1;
  hipEvent_t ea0, ea1; CK(hipEventCreate(&ea0)); CK(hipEventCreate(&ea1));
  for (int vg = 0; vg < 2; ++vg)
    for (int any = 0; any < 3; ++any) {
      // (1) one step, looked at in detail
      for (int rep = 0; rep < 3; ++rep) {
        ++tok;
        hipExtLaunchKernelGGL(kern_a, dim3(n), dim3(64), 0, st, ea0, ea1, 0, rec, tend, tbeg, busy_us * 100, tok, 1.0f);
        if (vg == 0) hipExtLaunchKernelGGL(kern_b<32>, dim3(nb), dim3(64), 0, any == 2 ? st2 : st, nullptr, nullptr, any == 1 ? hipExtAnyOrderLaunch : 0, (const Rec*)rec, n, tok, out, sum);
        else hipExtLaunchKernelGGL(kern_b<64>, dim3(nb), dim3(64), 0, any == 2 ? st2 : st, nullptr, nullptr, any == 1 ? hipExtAnyOrderLaunch : 0, (const Rec*)rec, n, tok, out, sum);
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        float msA = 0.f; CK(hipEventElapsedTime(&msA, ea0, ea1));
        unsigned long long te = 0, tb = ~0ull;
        {
          std::vector<unsigned long long> he(n), hb(n);
          CK(hipMemcpy(he.data(), tend, 8 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), tbeg, 8 * n, hipMemcpyDeviceToHost));
          for (int i = 0; i < n; ++i) { te = std::max(te, he[i]); tb = std::min(tb, hb[i]); }
        }
        CK(hipMemcpy(h.data(), out, nb * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), sum, nb * 4, hipMemcpyDeviceToHost));
        long long smin = 1ll << 60, smax = -(1ll << 60), seen_max = -(1ll << 60), polls = 0, gave = 0;
        double total = 0;
        for (int b = 0; b < nb; ++b) {
          smin = std::min(smin, (long long)(h[b * 4] - te)); smax = std::max(smax, (long long)(h[b * 4] - te));
          seen_max = std::max(seen_max, (long long)(h[b * 4 + 1] - te)); polls += h[b * 4 + 2]; gave += h[b * 4 + 3];
          total += hs[b];
        }
        if (rep == 2)
          printf("B %s VGPRs, %s: A ran %.2f 'us' by stamps (100 ticks = 1 'us'), %.2f us by its events; B's first start %.2f after A's first start; B's waves started %.2f .. %.2f us and had seen every token %.2f us after A's last end; %lld polls, %lld gave up, sum %s\n",
                 vg ? ">32" : "<=32", any == 2 ? "on a SECOND STREAM" : any ? "ANY-ORDER" : "in order", (te - tb) * 0.01, msA * 1000.f, (double)((long long)(te - tb) + smin) * 0.01, smin * 0.01, smax * 0.01, seen_max * 0.01, polls, gave,
                 total == 3072.0 / 8 * 28 ? "ok" : "WRONG");
      }
      // (2) steps back to back
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int steps = 200;
      for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < steps; ++i) {
          ++tok;
          hipLaunchKernelGGL(kern_a, dim3(n), dim3(64), 0, st, rec, tend, tbeg, busy_us * 100, tok, 1.0f);
          if (vg == 0) hipExtLaunchKernelGGL(kern_b<32>, dim3(nb), dim3(64), 0, any == 2 ? st2 : st, nullptr, nullptr, any == 1 ? hipExtAnyOrderLaunch : 0, (const Rec*)rec, n, tok, out, sum);
          else hipExtLaunchKernelGGL(kern_b<64>, dim3(nb), dim3(64), 0, any == 2 ? st2 : st, nullptr, nullptr, any == 1 ? hipExtAnyOrderLaunch : 0, (const Rec*)rec, n, tok, out, sum);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 1 && any != 2) printf("   %d steps back to back: %.2f us per step\n", steps, ms * 1000.f / steps);
      }
    }

  // (3) A on a PRIVATE stream, B (the poller) on the caller's stream; the private stream waits for an event of the caller's stream
  //     before every A (the inputs of a step come from the caller's stream), or for nothing (upper bound, not a usable protocol)
  for (int ev = 0; ev < 2; ++ev) {
    hipEvent_t e0, e1, dep; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    const int steps = 200;
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < steps; ++i) {
        ++tok;
        if (ev) { CK(hipEventRecord(dep, st)); CK(hipStreamWaitEvent(st2, dep, 0)); }
        hipLaunchKernelGGL(kern_a, dim3(n), dim3(64), 0, st2, rec, tend, tbeg, busy_us * 100, tok, 1.0f);
        hipExtLaunchKernelGGL(kern_b<32>, dim3(nb), dim3(64), 0, st, nullptr, nullptr, 0, (const Rec*)rec, n, tok, out, sum);
      }
      CK(hipEventRecord(e1, st));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass == 1) printf("A on a private stream, poller B on the caller's stream, %s: %.2f us per step\n", ev ? "private stream waits for the caller's event before every A" : "NO dependency (upper bound)", ms * 1000.f / steps);
    }
  }
  // A alone
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 200;
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(kern_a, dim3(n), dim3(64), 0, st, rec, tend, tbeg, busy_us * 100, ++tok, 1.0f);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass == 1) printf("A alone, %d launches back to back: %.2f us per launch\n", steps, ms * 1000.f / steps);
    }
  }
  return 0;
}
