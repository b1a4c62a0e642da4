// Probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950 (prints who ends up where).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  const unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[l] = r[0]; o[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *nm[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int v = 0; v < 4; v++) { printf("%s:", nm[v]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[v * 64 + l]); printf("\n"); }
  return 0;
}
