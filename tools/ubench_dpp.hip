// tools/ubench_dpp.hip -- dependent-issue cost of cross-lane moves on gfx950 (one wave): DPP wave_shr:1, row_shr:1,
// v_readlane -> SGPR -> VALU, taken scalar branches.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dpp ubench_dpp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int CTRL> __device__ __forceinline__ double dpp64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__global__ void k(double *out, unsigned long long *cyc, double a, double b, int n) {
  double x = a + threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; i++) x = __builtin_fma(dpp64<0x138>(x), b, a);    // wave_shr:1 + fma
  unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; i++) x = __builtin_fma(dpp64<0x111>(x), b, a);    // row_shr:1 + fma
  unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; i++) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), i & 63);
    x = __builtin_fma(__hiloint2double(hi, lo), b, x);
  }
  unsigned long long t3 = __builtin_readcyclecounter();
  // a loop with several taken uniform branches per iteration (n is a kernel argument == 4)
  double y = a;
  for (int i = 0; i < N; i++) {
    if (i < n * 1000000) { y = __builtin_fma(y, b, a); asm volatile("" ::: "memory"); }
    if (i + 1 < n * 1000000) { y = __builtin_fma(y, b, a); asm volatile("" ::: "memory"); }
    if (i + 2 < n * 1000000) { y = __builtin_fma(y, b, a); asm volatile("" ::: "memory"); }
    if (i + 3 < n * 1000000) { y = __builtin_fma(y, b, a); asm volatile("" ::: "memory"); }
  }
  unsigned long long t4 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N; i++) x = __builtin_fma(dpp64<0x142>(x), b, a);    // row_bcast:15 + fma
  unsigned long long t5 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main() {
  double *o; unsigned long long *c, h[8];
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 64);
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001, 0.9999999, 4); hipDeviceSynchronize(); }
  hipMemcpy(h, c, 40, hipMemcpyDeviceToHost);
  printf("wave_shr:1 (2 dpp) + fma : %.1f cycles\n", (double)h[0] / N);
  printf("row_shr:1  (2 dpp) + fma : %.1f cycles\n", (double)h[1] / N);
  printf("2 readlane + fma         : %.1f cycles\n", (double)h[2] / N);
  printf("4 guarded fma per iter   : %.1f cycles per iteration\n", (double)h[3] / N);
  printf("row_bcast:15 (2 dpp)+fma : %.1f cycles\n", (double)h[4] / N);
  return 0;
}
