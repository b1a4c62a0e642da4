// tools/probe_cumask.hip -- does hipExtStreamCreateWithCUMask partition an MI355X, and how are the mask bits
// enumerated over the 8 XCDs?  Every workgroup records where it ran (XCC_ID, HW_ID: se / sh / cu) and spins for a while;
// two kernels on two streams with disjoint masks are timed alone and together.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_cumask tools/probe_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

__global__ void where(unsigned *out, long long spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void summarize(const char *name, const std::vector<unsigned> &h, int nb) {
  std::map<unsigned, int> per_xcc;
  std::map<unsigned, int> cus;   // key: xcc<<16 | se<<8 | sh<<4.. -> count
  for (int b = 0; b < nb; b++) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc]++;
    cus[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  std::printf("%s: %d workgroups on %zu distinct CUs; per XCC:", name, nb, cus.size());
  for (auto &kv : per_xcc) std::printf(" x%u=%d", kv.first, kv.second);
  std::printf("\n   CUs per XCC:");
  std::map<unsigned, int> cpx;
  for (auto &kv : cus) cpx[kv.first >> 16]++;
  for (auto &kv : cpx) std::printf(" x%u=%d", kv.first, kv.second);
  std::printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  std::printf("device %s, %d CUs\n", prop.gcnArchName, ncu);
  const int words = (ncu + 31) / 32;
  const int nb = 2048;
  unsigned *d; hipMalloc(&d, 1 << 20);
  std::vector<unsigned> h(nb * 2);
  auto run = [&](const char *name, const std::vector<unsigned> &mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { std::printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, st, d, 20000LL);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    summarize(name, h, nb);
    hipStreamDestroy(st);
  };
  { std::vector<unsigned> m(words, 0xffffffffu); run("all bits", m); }
  { std::vector<unsigned> m(words, 0); m[0] = 0xffffffffu; run("bits 0..31", m); }
  { std::vector<unsigned> m(words, 0); for (int i = 0; i < words / 2; i++) m[i] = 0xffffffffu; run("low half", m); }
  { std::vector<unsigned> m(words, 0); for (int i = words / 2; i < words; i++) m[i] = 0xffffffffu; run("high half", m); }
  { std::vector<unsigned> m(words, 0x55555555u); run("even bits", m); }
  { std::vector<unsigned> m(words, 0); for (int i = 0; i < ncu; i++) if ((i % 8) < 2) m[i / 32] |= 1u << (i % 32); run("bits i%8<2", m); }
  { std::vector<unsigned> m(words, 0); for (int i = 0; i < ncu; i++) if ((i / 8) % 4 == 0) m[i / 32] |= 1u << (i % 32); run("bits (i/8)%4==0", m); }
  // concurrency: two long kernels on disjoint halves, alone and together
  {
    std::vector<unsigned> lo(words, 0), hi(words, 0);
    for (int i = 0; i < ncu; i++) (((i / 8) % 2 == 0) ? lo : hi)[i / 32] |= 1u << (i % 32);
    hipStream_t a, b;
    hipExtStreamCreateWithCUMask(&a, (unsigned)lo.size(), lo.data());
    hipExtStreamCreateWithCUMask(&b, (unsigned)hi.size(), hi.data());
    hipEvent_t e0, e1, e2, e3;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    const long long spin = 2000000;   // ~1 ms per workgroup wave
    const int big = 128 * 8 * 4;      // 8 waves per CU of half the chip, 4 rounds
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, a); hipLaunchKernelGGL(where, dim3(big), dim3(64), 0, a, d, spin); hipEventRecord(e1, a);
      hipStreamSynchronize(a);
      float alone = 0; hipEventElapsedTime(&alone, e0, e1);
      hipEventRecord(e0, a); hipLaunchKernelGGL(where, dim3(big), dim3(64), 0, a, d, spin); hipEventRecord(e1, a);
      hipEventRecord(e2, b); hipLaunchKernelGGL(where, dim3(big), dim3(64), 0, b, d + 2 * big, spin); hipEventRecord(e3, b);
      hipStreamSynchronize(a); hipStreamSynchronize(b);
      float ta = 0, tb = 0; hipEventElapsedTime(&ta, e0, e1); hipEventElapsedTime(&tb, e2, e3);
      std::printf("disjoint halves: alone %.2f ms, together %.2f / %.2f ms\n", alone, ta, tb);
    }
  }
  hipFree(d);
  return 0;
}
