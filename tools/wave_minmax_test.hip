// development check of wave_max_i / wave_min_i (csrc/sfm_common.h):  hipcc --offload-arch=gfx950 tools/wave_minmax_test.hip -o /tmp/t && /tmp/t
#include <cstdio>
#include "../sfm-learner-chainer_amd/csrc/sfm_common.h"
namespace sfm { void set_error(const char*, ...) {} int fail(int c, const char*, ...) { return c; } int check_launch(const char*) { return 0; } }
__global__ void k(const int* in, int* out) {
  const int v = in[threadIdx.x];
  out[threadIdx.x] = sfm::wave_max_i(v);
  out[64 + threadIdx.x] = sfm::wave_min_i(v);
}
int main() {
  int h[64], o[128], *d, *e;
  int bad = 0;
  for (int t = 0; t < 200; ++t) {
    int mx = -0x7fffffff, mn = 0x7fffffff;
    for (int i = 0; i < 64; ++i) { h[i] = (rand() % 2001) - 1000; if (t % 3 == 0 && i % 5) h[i] = 0x3fffffff; if (t % 3 == 1 && i % 7) h[i] = -0x3fffffff; mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
    hipMalloc(&d, 256); hipMalloc(&e, 512);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, e);
    hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) if (o[i] != mx || o[64 + i] != mn) { if (bad < 5) printf("trial %d lane %d: max %d (want %d) min %d (want %d)\n", t, i, o[i], mx, o[64 + i], mn); ++bad; }
    hipFree(d); hipFree(e);
  }
  printf("mismatches: %d\n", bad);
  return bad != 0;
}
