// tools/ubench_latency.hip -- dependent-issue latencies on gfx950 (one wave): fp64 fma/add/mul,
// fp64 division, v_readlane round trip, LDS read. Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k(double *out, unsigned long long *cyc, double a, double b) {
  __shared__ double lds[256];
  lds[threadIdx.x] = a + threadIdx.x;
  __syncthreads();
  double x = a;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; i++) x = __builtin_fma(x, b, a);
  unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; i++) x = x + b;
  unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; i++) x = x * b;
  unsigned long long t3 = __builtin_readcyclecounter();
#pragma unroll 4
  for (int i = 0; i < N / 8; i++) x = a / (x + b);
  unsigned long long t4 = __builtin_readcyclecounter();
  // readlane round trip: VALU -> SGPR -> VALU
#pragma unroll 16
  for (int i = 0; i < N; i++) { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3); x = x + __hiloint2double(0x3ff00000, lo & 1); }
  unsigned long long t5 = __builtin_readcyclecounter();
  // dependent LDS read chain (pointer chase)
  int idx = threadIdx.x & 7;
#pragma unroll 8
  for (int i = 0; i < N / 8; i++) { double v = lds[idx]; idx = ((int)v) & 63; }
  unsigned long long t6 = __builtin_readcyclecounter();
  // 8 independent fma chains (throughput)
  double y0 = a, y1 = a + 1, y2 = a + 2, y3 = a + 3, y4 = a + 4, y5 = a + 5, y6 = a + 6, y7 = a + 7;
#pragma unroll 4
  for (int i = 0; i < N; i++) { y0 = __builtin_fma(y0, b, a); y1 = __builtin_fma(y1, b, a); y2 = __builtin_fma(y2, b, a); y3 = __builtin_fma(y3, b, a); y4 = __builtin_fma(y4, b, a); y5 = __builtin_fma(y5, b, a); y6 = __builtin_fma(y6, b, a); y7 = __builtin_fma(y7, b, a); }
  unsigned long long t7 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + idx + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; cyc[5] = t6 - t5; cyc[6] = t7 - t6; }
}
int main() {
  double *o; unsigned long long *c, h[8];
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 64);
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 1.0000001, 0.9999999); hipDeviceSynchronize(); }
  hipMemcpy(h, c, 56, hipMemcpyDeviceToHost);
  printf("dependent v_fma_f64 : %.1f cycles\n", (double)h[0] / N);
  printf("dependent v_add_f64 : %.1f cycles\n", (double)h[1] / N);
  printf("dependent v_mul_f64 : %.1f cycles\n", (double)h[2] / N);
  printf("dependent add+div   : %.1f cycles\n", (double)h[3] / (N / 8));
  printf("readlane+add loop   : %.1f cycles\n", (double)h[4] / N);
  printf("dependent LDS read  : %.1f cycles\n", (double)h[5] / (N / 8));
  printf("8 indep fma chains  : %.1f cycles per fma\n", (double)h[6] / (8.0 * N));
  return 0;
}
