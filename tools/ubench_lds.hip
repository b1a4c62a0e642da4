// tools/ubench_lds.hip -- LDS read issue cost on gfx950 for one wave: 64 distinct addresses,
// one broadcast address, and a single active lane; b64 and b128; plus the same with W waves
// per CU contending.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lds tools/ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
template <int MODE>
__device__ unsigned long long run(const double *lds, double &sink) {
  // MODE 0: distinct b64, 1: broadcast b64, 2: distinct b128 (16B per lane), 3: broadcast b128, 4: single-lane b64, 5: single-lane b128
  const int l = threadIdx.x & 63;
  const int base = (MODE == 0) ? l : (MODE == 2) ? 2 * l : 0;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const bool active = (MODE < 4) || l == 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (active) {
#pragma unroll 8
    for (int i = 0; i < N; i++) {
      const int o = base + ((i * 16) & 1023);
      if (MODE == 0 || MODE == 1 || MODE == 4) { acc0 += lds[o]; }
      else { const double2 v = *reinterpret_cast<const double2 *>(lds + (o & ~1)); acc0 += v.x; acc1 += v.y; }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  sink += acc0 + acc1 + acc2 + acc3;
  return t1 - t0;
}
__global__ void k(double *out, unsigned long long *cyc) {
  __shared__ __attribute__((aligned(16))) double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = i * 0.5;
  __syncthreads();
  double sink = 0;
  unsigned long long c[6];
  c[0] = run<0>(lds, sink); __syncthreads();
  c[1] = run<1>(lds, sink); __syncthreads();
  c[2] = run<2>(lds, sink); __syncthreads();
  c[3] = run<3>(lds, sink); __syncthreads();
  c[4] = run<4>(lds, sink); __syncthreads();
  c[5] = run<5>(lds, sink);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 6; i++) cyc[i] = c[i];
}
int main() {
  double *o; unsigned long long *c, h[6];
  hipMalloc(&o, 1024 * 1024 * 8); hipMalloc(&c, 64);
  const char *nm[6] = {"distinct b64", "broadcast b64", "distinct b128", "broadcast b128", "one lane b64", "one lane b128"};
  for (int waves : {1, 4, 8, 16}) {
    // `waves` one-wave workgroups per CU: 256 CUs x waves blocks of 64 threads
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(256 * waves), dim3(64), 0, 0, o, c); hipDeviceSynchronize(); }
    hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
    printf("%2d waves/CU:", waves);
    for (int i = 0; i < 6; i++) printf("  %s %.1f", nm[i], (double)h[i] / N);
    printf("  (cycles per load+add)\n");
  }
  return 0;
}
