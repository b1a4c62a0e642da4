// sac_amd/csrc/kernels_coder.hip -- gfx950 kernels: bitplane/SSE range coder streams and the
// sparse-PCM residual remap (Remap::Map + CalcRemapError, map.cpp:175-187, libsac.cpp:230-251).
#include "coder.h"
#include <atomic>
#include <cstdlib>
#include "kernels.h"

namespace sacamd {

// Several independent streams (one wave each) per workgroup share one copy of the read-only tables.
constexpr int kCoderStreamsPerWg = 6;   // upper bound; ~153 KB of LDS: one workgroup per CU (1536 streams resident on 256 CUs)
struct CoderLdsLayout {
  static constexpr size_t o_tabs = 0;
  static constexpr size_t o_stream = (o_tabs + sizeof(CoderTabs) + 15) / 16 * 16;
  // per stream
  static constexpr size_t s_model = 0;
  static constexpr size_t s_win = (s_model + sizeof(CoderModel) + 15) / 16 * 16;
  static constexpr size_t s_map = (s_win + sizeof(CoderWin) + 15) / 16 * 16;
  static constexpr size_t s_total = (s_map + sizeof(MapModel) + 15) / 16 * 16;
  static constexpr size_t bytes(int streams) { return o_stream + (size_t)streams * s_total; }
};

size_t coder_state_bytes() { return sizeof(CntL) * 65536; }

__global__ __launch_bounds__(64 * kCoderStreamsPerWg) void k_coder(const CoderJob *jobs, int count, const int *s2u, const int *s2u_map, const unsigned char *used,
                                               const unsigned short *laplace, const short *gfwd, const unsigned short *ginv,
                                               unsigned char *state, size_t stride, unsigned char *out, int *len,
                                               const int *mb, unsigned char *compact, long long *compact_at, int serial_chain) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CoderTabs &T = *reinterpret_cast<CoderTabs *>(smem + CoderLdsLayout::o_tabs);
  coder_tabs_init(T, gfwd, ginv, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int ji = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (ji >= count) return;                       // no further workgroup-wide barriers below
  CoderJob job = jobs[ji];
  if (job.maxbpn < 0) job.maxbpn = mb[-1 - job.maxbpn];
  char *sb = smem + CoderLdsLayout::o_stream + (size_t)wave * CoderLdsLayout::s_total;
  CoderModel &M = *reinterpret_cast<CoderModel *>(sb + CoderLdsLayout::s_model);
  CoderWin &W = *reinterpret_cast<CoderWin *>(sb + CoderLdsLayout::s_win);
  MapModel &MM = *reinterpret_cast<MapModel *>(sb + CoderLdsLayout::s_map);
  CntL *csig0 = reinterpret_cast<CntL *>(state + (size_t)ji * stride);
  const unsigned short *plap = laplace + (size_t)kLaplacePlanes * kLaplaceAvg;
  ExecDevWave ex;
  const int *src = job.with_map ? s2u_map : s2u;       // remapped residual stream for the MapEncoder variant
  const int l = coder_stream(ex, src + job.off_in, job.n, job.maxbpn, job.with_map ? used + job.off_used : nullptr, laplace, gfwd, ginv,
                             plap, csig0, out + job.off_out, job.cap, M, T, W, MM, serial_chain);
  const int lane = threadIdx.x & 63;
  if (lane == 0) len[ji] = l;
  if (compact) {
    // the finished payload moves behind the payloads finished so far: the host then fetches ONE contiguous block instead of
    // one copy per stream out of the sparsely filled capacity buffers
    const int lw = __builtin_amdgcn_readfirstlane(l);
    const int keep = lw < job.cap ? lw : job.cap;
    long long at = 0;
    if (lane == 0) { at = (long long)atomicAdd((unsigned long long *)&compact_at[count], (unsigned long long)((keep + 15) & ~15)); compact_at[ji] = at; }
    at = __shfl(at, 0, 64);
    __threadfence();
    const unsigned char *src = out + job.off_out;
    for (int i = lane * 16; i < keep; i += 64 * 16) {     // off_out and at are multiples of 16; capacities are padded to 16
      *reinterpret_cast<uint4 *>(compact + at + i) = *reinterpret_cast<const uint4 *>(src + i);
    }
  }
}

void launch_coder(hipStream_t s, const CoderJob *d_jobs, int count, const int *d_s2u, const int *d_s2u_map, const unsigned char *d_used,
                  const unsigned short *d_laplace, const short *d_fwd, const unsigned short *d_inv, unsigned char *d_state,
                  size_t state_stride, unsigned char *d_out, int *d_len, const int *d_maxbpn, unsigned char *d_compact, long long *d_compact_at) {
  if (count <= 0) return;
  if (d_compact && hipMemsetAsync(d_compact_at + count, 0, sizeof(long long), s) != hipSuccess) return;
  {   // per DEVICE opt-in to > 64 KB dynamic LDS (idempotent; launchers may be called from several host threads)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute((const void *)k_coder, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CoderLdsLayout::bytes(kCoderStreamsPerWg)) != hipSuccess) return;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  // the fewest streams per CU that still keep every stream resident at once (256 CUs): one-stream
  // workgroups (two per CU) up to 512 streams, then one workgroup per CU with three to six streams
  // (a serial stream issues an instruction every few cycles, so two of them share a SIMD well;
  // a second round of workgroups would double the latency of the whole launch)
  static_assert(CoderLdsLayout::bytes(kCoderStreamsPerWg) <= 160 * 1024, "coder workgroup exceeds the LDS of a CU");
  int spw = 1;
  if (count > 512) { spw = 3; while (spw < kCoderStreamsPerWg && count > 256 * spw) spw++; }
  const int wgs = (count + spw - 1) / spw;
  // parity tap: SACAMD_CODER_SERIAL=1 runs every decision on lane 0 with the body the CPU emulation runs (coder_step)
  static const int serial_chain = [] { const char *e = std::getenv("SACAMD_CODER_SERIAL"); return (e && e[0] == '1') ? 1 : 0; }();
  hipLaunchKernelGGL(k_coder, dim3(wgs), dim3(64 * spw), CoderLdsLayout::bytes(spw), s, d_jobs, count, d_s2u, d_s2u_map, d_used, d_laplace,
                     d_fwd, d_inv, d_state, state_stride, d_out, d_len, d_maxbpn, d_compact, d_compact_at, serial_chain);
}

// ------------------------------------------------------------------ entropy DEcoder (FrameCoder::DecodeMonoFrame, libsac.cpp:280-298)
// one wave per stream: [MapEncoder::Decode ->] BitplaneCoder::Decode -> U2S; output = the signed residual (mapped residual
// for mapped streams) in `err`, the decoded used-value flags in `used`
__global__ __launch_bounds__(64 * kCoderStreamsPerWg) void k_decoder(const DecJob *jobs, int count, const unsigned char *in, int *err, unsigned char *used,
                                                 const unsigned short *laplace, const short *gfwd, const unsigned short *ginv,
                                                 unsigned char *state, size_t stride, int *consumed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CoderTabs &T = *reinterpret_cast<CoderTabs *>(smem + CoderLdsLayout::o_tabs);
  coder_tabs_init(T, gfwd, ginv, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int ji = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (ji >= count) return;                       // no further workgroup-wide barriers below
  const DecJob job = jobs[ji];
  char *sb = smem + CoderLdsLayout::o_stream + (size_t)wave * CoderLdsLayout::s_total;
  CoderModel &M = *reinterpret_cast<CoderModel *>(sb + CoderLdsLayout::s_model);
  CoderWin &W = *reinterpret_cast<CoderWin *>(sb + CoderLdsLayout::s_win);
  MapModel &MM = *reinterpret_cast<MapModel *>(sb + CoderLdsLayout::s_map);
  CntL *csig0 = reinterpret_cast<CntL *>(state + (size_t)ji * stride);
  const unsigned short *plap = laplace + (size_t)kLaplacePlanes * kLaplaceAvg;
  ExecDevWave ex;
  int *dst = err + job.off_out;
  const int used_bytes = coder_stream_dec(ex, in + job.off_in, job.inlen, job.n, job.maxbpn, job.with_map ? used + job.off_used : nullptr, laplace,
                                          plap, csig0, dst, M, T, W, MM);
  const int l = threadIdx.x & 63;
  for (int i = l; i < job.n; i += 64) { const int v = dst[i]; dst[i] = (v & 1) ? ((v + 1) >> 1) : -(v >> 1); }   // MathUtils::U2S (utils.h:268-273)
  if (l == 0) consumed[ji] = used_bytes;
}

void launch_decoder(hipStream_t s, const DecJob *d_jobs, int count, const unsigned char *d_in, int *d_err, unsigned char *d_used,
                    const unsigned short *d_laplace, const short *d_fwd, const unsigned short *d_inv, unsigned char *d_state,
                    size_t state_stride, int *d_consumed) {
  if (count <= 0) return;
  {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      if (hipFuncSetAttribute((const void *)k_decoder, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CoderLdsLayout::bytes(kCoderStreamsPerWg)) != hipSuccess) return;
      done.fetch_or(bit, std::memory_order_release);
    }
  }
  int spw = 1;
  if (count > 512) { spw = 3; while (spw < kCoderStreamsPerWg && count > 256 * spw) spw++; }
  const int wgs = (count + spw - 1) / spw;
  hipLaunchKernelGGL(k_decoder, dim3(wgs), dim3(64 * spw), CoderLdsLayout::bytes(spw), s, d_jobs, count, d_in, d_err, d_used, d_laplace,
                     d_fwd, d_inv, d_state, state_stride, d_consumed);
}


// ------------------------------------------------------------------ parity tap: the device's exp / pow ports and PredictLaplace
// kind 0: sa_exp(x)  1: sa_pow(x, y)  2: laplace_direct((unsigned)x, (int)y) as a double
__global__ void k_libm_tap(int kind, const double *x, const double *y, int n, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = kind == 0 ? sa_exp(x[i]) : kind == 1 ? sa_pow(x[i], y[i]) : (double)laplace_direct((unsigned)x[i], (int)y[i]);
}
void launch_libm_tap(hipStream_t s, int kind, const double *d_x, const double *d_y, int n, double *d_out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_libm_tap, dim3((n + 255) / 256), dim3(256), 0, s, kind, d_x, d_y, n, d_out);
}

// ------------------------------------------------------------------ remap
// grid: one block (256) per (frame,channel) job.  prefix scratch: 65540 ints per job.
__device__ __forceinline__ int used_flag(const unsigned char *u, int v) {
  if (v == 0) return 1;
  return v > 0 ? u[32769 + v] : u[-v];
}

__global__ __launch_bounds__(256) void k_remap(const RemapJob *jobs, const unsigned char *used, const int *pred, const int *err, int *s2u_map,
                                                int *prefix_scratch, long long *out3) {
  __shared__ int part[256];
  __shared__ long long s_ll[4];
  __shared__ int s_i[4];
  const RemapJob job = jobs[blockIdx.x];
  const unsigned char *u = used + job.off_used;
  int *prefix = prefix_scratch + (size_t)blockIdx.x * 65540;
  // prefix[j] = #used values in [-32768, -32768 + j - 1], j = 0..65537
  constexpr int N = 65537, PER = (N + 255) / 256;
  const int j0 = threadIdx.x * PER, j1 = (j0 + PER < N) ? j0 + PER : N;
  int loc = 0;
  for (int j = j0; j < j1; j++) loc += used_flag(u, j - 32768);
  part[threadIdx.x] = loc;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int j = j0; j < j1; j++) { prefix[j] = run; run += used_flag(u, j - 32768); }
  if (j1 == N && j0 < N) prefix[N] = run;
  __syncthreads();
  auto cnt = [&](int a, int b) {
    a = a < -32768 ? -32768 : a; b = b > 32768 ? 32768 : b;
    return a > b ? 0 : prefix[b + 32768 + 1] - prefix[a + 32768];
  };
  const int *p = pred + job.off, *e = err + job.off;
  int *um = s2u_map + job.off;
  long long se = 0, sm = 0; int mx = 0;
  for (int i = threadIdx.x; i < job.n; i += 256) {
    const int ev = e[i], pv = p[i];
    int m = 0;
    if (ev > 0) m = cnt(pv + 1, pv + ev); else if (ev < 0) m = -cnt(pv + ev, pv - 1);
    const int v = m < 0 ? 2 * (-m) : (m > 0 ? 2 * m - 1 : 0);
    um[i] = v;
    se += ev < 0 ? -(long long)ev : ev; sm += m < 0 ? -(long long)m : m; mx = v > mx ? v : mx;
  }
  // block reductions (butterfly + waves in order)
  for (int d = 32; d >= 1; d >>= 1) { se += __shfl_xor(se, d, 64); sm += __shfl_xor(sm, d, 64); const int o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
  const int w = threadIdx.x >> 6;
  __shared__ long long s_ll2[4];
  if ((threadIdx.x & 63) == 0) { s_ll[w] = se; s_ll2[w] = sm; s_i[w] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long a = 0, b = 0; int m = 0;
    for (int i = 0; i < 4; i++) { a += s_ll[i]; b += s_ll2[i]; m = s_i[i] > m ? s_i[i] : m; }
    out3[3 * blockIdx.x] = a; out3[3 * blockIdx.x + 1] = b; out3[3 * blockIdx.x + 2] = m;
  }
}

void launch_remap(hipStream_t s, const RemapJob *d_jobs, int count, const unsigned char *d_used, const int *d_pred, const int *d_err,
                  int *d_s2u_map, int *d_prefix_scratch, long long *d_out3) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_remap, dim3(count), dim3(256), 0, s, d_jobs, d_used, d_pred, d_err, d_s2u_map, d_prefix_scratch, d_out3);
}

}  // namespace sacamd
