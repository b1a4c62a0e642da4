// tools/probe_newbcast.hip -- semantics and dependent-chain rate of v_fmac_f64_dpp row_newbcast:N on gfx950 (DP-ALU DPP):
// lane l of every row of 16 takes src0 from lane N of ITS row.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_newbcast tools/probe_newbcast.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, unsigned long long *cyc) {
  const int l = threadIdx.x;
  double L = 100.0 * (l >> 4) + (l & 15), w = 1.0, s = 0.0, z;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(L), "v"(w));
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(z) : "v"(L));
  out[l] = s; out[64 + l] = z;
  // dependent chain: 16 x 256 fmacs
  double a = 1e-3 * (l & 15), acc = 0.0, ww = 1.0000001;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 256; i++) {
#define F(N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(ww));
    F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[128 + l] = acc;
  if (l == 0) cyc[0] = t1 - t0;
}
int main() {
  double *o, h[192]; unsigned long long *c, hc;
  hipMalloc(&o, sizeof(h)); hipMalloc(&c, 8);
  for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c); hipDeviceSynchronize(); }
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  printf("fmac row_newbcast:3 -> lanes 0,15,16,31,48,63: %g %g %g %g %g %g (expect 3 3 103 103 303 303)\n", h[0], h[15], h[16], h[31], h[48], h[63]);
  printf("mov  row_newbcast:5 -> lanes 0,17,40,63: %g %g %g %g (expect 5 105 205 305)\n", h[64], h[64 + 17], h[64 + 40], h[64 + 63]);
  printf("dependent v_fmac_f64_dpp row_newbcast: %.2f cycles per term\n", (double)hc / 4096.0);
  return 0;
}
