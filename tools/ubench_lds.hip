// tools/ubench_lds.hip -- LDS read issue cost on gfx950, one wave alone on a CU: back-to-back
// independent loads (16 per batch, then one wait), for b64 / read2_b64 / b128, with all 64
// lanes on distinct addresses, all lanes on one address, and a single active lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_lds tools/ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
#define LD16(INSTR, STEP)                                                                                     \
  asm volatile(INSTR " %0, %16 offset:%17\n" INSTR " %1, %16 offset:%18\n" INSTR " %2, %16 offset:%19\n"      \
               INSTR " %3, %16 offset:%20\n" INSTR " %4, %16 offset:%21\n" INSTR " %5, %16 offset:%22\n"      \
               INSTR " %6, %16 offset:%23\n" INSTR " %7, %16 offset:%24\n" INSTR " %8, %16 offset:%25\n"      \
               INSTR " %9, %16 offset:%26\n" INSTR " %10, %16 offset:%27\n" INSTR " %11, %16 offset:%28\n"    \
               INSTR " %12, %16 offset:%29\n" INSTR " %13, %16 offset:%30\n" INSTR " %14, %16 offset:%31\n"   \
               INSTR " %15, %16 offset:%32\n s_waitcnt lgkmcnt(0)"                                           \
               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), \
                 "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) \
               : "v"(addr), "n"(0 * STEP), "n"(1 * STEP), "n"(2 * STEP), "n"(3 * STEP), "n"(4 * STEP), "n"(5 * STEP), \
                 "n"(6 * STEP), "n"(7 * STEP), "n"(8 * STEP), "n"(9 * STEP), "n"(10 * STEP), "n"(11 * STEP),   \
                 "n"(12 * STEP), "n"(13 * STEP), "n"(14 * STEP), "n"(15 * STEP))

template <int W, int MODE> __device__ unsigned long long run(unsigned addr_distinct, double &sink) {
  // W: 0 = ds_read_b64, 1 = ds_read_b128.  MODE: 0 distinct, 1 broadcast, 2 single lane
  const unsigned addr = MODE == 0 ? addr_distinct : 0u;
  const bool active = MODE != 2 || (threadIdx.x & 63) == 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  if (active) {
    for (int i = 0; i < REP; i++) {
      // xor-fold on integer halves: cheap, keeps every load live
      unsigned f = 0;
      if (W == 0) { double r[16]; LD16("ds_read_b64", 1024); for (int q = 0; q < 16; q++) f ^= (unsigned)__double2loint(r[q]); }
      else { double2 r[16]; LD16("ds_read_b128", 1024); for (int q = 0; q < 16; q++) f ^= (unsigned)__double2loint(r[q].y); }
      sink += (double)(f & 1);
    }
  }
  return __builtin_readcyclecounter() - t0;
}
__global__ void k(double *out, unsigned long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.5;
  __syncthreads();
  double sink = 0;
  unsigned long long c[6];
  c[0] = run<0, 0>(threadIdx.x * 8, sink);  c[1] = run<0, 1>(0, sink);  c[2] = run<0, 2>(0, sink);
  c[3] = run<1, 0>(threadIdx.x * 16, sink); c[4] = run<1, 1>(0, sink);  c[5] = run<1, 2>(0, sink);
  out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 6; i++) cyc[i] = c[i];
}
int main() {
  double *o; unsigned long long *c, h[6];
  hipMalloc(&o, 1024 * 1024 * 8); hipMalloc(&c, 64);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  const char *nm[6] = {"b64 distinct", "b64 broadcast", "b64 one-lane", "b128 distinct", "b128 broadcast", "b128 one-lane"};
  for (int waves : {1, 4}) {
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(256 * waves), dim3(64), 32768, 0, o, c); hipDeviceSynchronize(); }
    hipMemcpy(h, c, 48, hipMemcpyDeviceToHost);
    printf("%d wave(s)/CU, cycles per load:", waves);
    for (int i = 0; i < 6; i++) printf("  %s %.1f", nm[i], (double)h[i] / (REP * 16.0));
    printf("\n");
  }
  return 0;
}
