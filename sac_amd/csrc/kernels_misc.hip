// sac_amd/csrc/kernels_misc.hip -- frame analysis, search objective (cost functions), S2U.
#include "kernels.h"
#include "simt.h"

namespace sacamd {

// ------------------------------------------------------------------ block reductions
template <class T, class OP>
__device__ __forceinline__ T block_reduce(T v, T *scratch, OP op) {
  // fixed order: butterfly inside each wave, then waves in order -> deterministic
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = op(v, __shfl_xor(v, d, 64));
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  T r = scratch[0];
  for (int i = 1; i < nw; i++) r = op(r, scratch[i]);
  return r;
}

// ------------------------------------------------------------------ analyse
// FrameCoder::AnalyseMonoChannel (libsac.cpp:626-651) + mean removal (:452-458) +
// Remap::Analyse used-value flags (map.cpp:126-146).  grid (frame, ch), block 256.
template <class LOAD>
__device__ __forceinline__ void analyse_body(int n, LOAD load, int zero_mean, int *dst, FrameStatsD *st, unsigned char *used) {
  __shared__ long long s_sum[4];
  __shared__ int s_i[4];
  long long sum = 0;
  int mn = 2147483647, mx = -2147483647 - 1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int v = load(i);
    sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
    if (used) {
      // usedl[-v] for v<0, usedh[v] for v>0; |v| <= 32768 (map.cpp:130-145)
      if (v > 0 && v <= 32768) used[32769 + v] = 1;
      else if (v < 0 && -v <= 32768) used[-v] = 1;
    }
  }
  sum = block_reduce(sum, s_sum, [](long long a, long long b) { return a + b; });
  mn = block_reduce(mn, s_i, [](int a, int b) { return a < b ? a : b; });
  mx = block_reduce(mx, s_i, [](int a, int b) { return a > b ? a : b; });
  int mean = 0;
  if (n > 0 && zero_mean) mean = (int)floor((double)sum / (double)n);
  if (n == 0) { mn = 0; mx = 0; }
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = load(i) - mean;
  if (threadIdx.x == 0) { st->mean = mean; st->minval = mn - mean; st->maxval = mx - mean; st->numsamples = n; }
}

__global__ __launch_bounds__(256) void k_analyse_s16(int nch, const int16_t *il, const long long *frame_off, const int *nsamp,
                                                      int zero_mean, int *pcm, long long fs, long long cs, FrameStatsD *stats,
                                                      unsigned char *used) {
  const int f = blockIdx.x, ch = blockIdx.y;
  const int16_t *src = il + frame_off[f] * nch + ch;
  unsigned char *u = used ? used + ((size_t)f * nch + ch) * 65540 : nullptr;
  analyse_body(nsamp[f], [&](int i) { return (int)src[(long long)i * nch]; }, zero_mean, pcm + f * fs + ch * cs, stats + f * nch + ch, u);
}
__global__ __launch_bounds__(256) void k_analyse_i32(int nch, const int *raw, long long rfs, long long rcs, const int *nsamp,
                                                      int zero_mean, int *pcm, long long fs, long long cs, FrameStatsD *stats,
                                                      unsigned char *used) {
  const int f = blockIdx.x, ch = blockIdx.y;
  const int *src = raw + f * rfs + ch * rcs;
  unsigned char *u = used ? used + ((size_t)f * nch + ch) * 65540 : nullptr;
  analyse_body(nsamp[f], [&](int i) { return src[i]; }, zero_mean, pcm + f * fs + ch * cs, stats + f * nch + ch, u);
}

void launch_analyse_s16(hipStream_t s, int nframes, int nch, const int16_t *d_il, const long long *d_frame_off, const int *d_nsamp,
                        int zero_mean, int *d_pcm, long long fs, long long cs, FrameStatsD *d_stats, unsigned char *d_used) {
  hipLaunchKernelGGL(k_analyse_s16, dim3(nframes, nch), dim3(256), 0, s, nch, d_il, d_frame_off, d_nsamp, zero_mean, d_pcm, fs, cs, d_stats, d_used);
}
void launch_analyse_i32(hipStream_t s, int nframes, int nch, const int *d_raw, long long rfs, long long rcs, const int *d_nsamp,
                        int zero_mean, int *d_pcm, long long fs, long long cs, FrameStatsD *d_stats, unsigned char *d_used) {
  hipLaunchKernelGGL(k_analyse_i32, dim3(nframes, nch), dim3(256), 0, s, nch, d_raw, rfs, rcs, d_nsamp, zero_mean, d_pcm, fs, cs, d_stats, d_used);
}


// ------------------------------------------------------------------ sparse-PCM block analysis
// SparsePCM::Analyse (libsac/sparse.h:31-73) for one block of one channel per workgroup:
// out[4b..] = { sum |val|, sum |rank(val)|, #used values, range } with rank = number of used
// values in (0, val] resp. [val, 0) (val2rank_fast with p = 0).  The used set is a bitmap over
// [min, max] in LDS (16-bit material: <= 65536 values); range > kSparseMaxRange -> out[3] = -1.
constexpr int kSparseMaxRange = 1 << 17;
constexpr int kSparseWords = kSparseMaxRange / 32;
// the sums over one block of samples, given zeroed bitmap words `bits` [nw] and prefix storage `pre` [nw + 1] (LDS for
// 16-bit material, global scratch for wider ranges)
__device__ void sparse_cost_body(const int *x, int n, int mn, int N, unsigned *bits, int *pre, long long *o) {
  __shared__ int part[256];
  __shared__ long long s_ll[4];
  const int nw = (N + 31) >> 5;
  for (int i = threadIdx.x; i < n; i += 256) { const int t = x[i] - mn; atomicOr(&bits[t >> 5], 1u << (t & 31)); }
  __threadfence_block();
  __syncthreads();
  // exclusive prefix of the per-word popcounts: each thread owns a contiguous run of words
  const int per = (nw + 255) / 256, w0 = threadIdx.x * per < nw ? threadIdx.x * per : nw, w1 = (w0 + per < nw) ? w0 + per : nw;
  int loc = 0;
  for (int w = w0; w < w1; w++) loc += __popc(bits[w]);
  part[threadIdx.x] = loc;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; } pre[nw] = run; }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int w = w0; w < w1; w++) { pre[w] = run; run += __popc(bits[w]); }
  __threadfence_block();
  __syncthreads();
  auto prefix = [&](int idx) {           // number of used values with index < idx, idx in [0, N]
    const int w = idx >> 5, r = idx & 31;
    return pre[w] + (r ? __popc(bits[w] & ((1u << r) - 1u)) : 0);
  };
  const long long pidx = 0 - (long long)mn;
  const int pa = prefix((int)(pidx + 1 < 0 ? 0 : (pidx + 1 > N ? N : pidx + 1)));
  const int pb = prefix((int)(pidx < 0 ? 0 : (pidx > N ? N : pidx)));
  long long s0 = 0, s1 = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int v = x[i];
    int r = 0;
    if (v > 0) r = prefix(v - mn + 1) - pa; else if (v < 0) r = prefix(v - mn) - pb;
    s0 += v < 0 ? -(long long)v : v;
    s1 += r < 0 ? -(long long)r : r;
  }
  s0 = block_reduce(s0, s_ll, [](long long a, long long c) { return a + c; });
  s1 = block_reduce(s1, s_ll, [](long long a, long long c) { return a + c; });
  if (threadIdx.x == 0) { o[0] = s0; o[1] = s1; o[2] = pre[nw]; o[3] = N; }
}

// wide != nullptr: this launch is the second pass over the blocks whose range did not fit the LDS bitmap; wide[2b] = start
// (32-bit words) of block b's scratch in `scratch` (bitmap words, then prefix ints), wide[2b+1] = words available
__global__ __launch_bounds__(256) void k_sparse_cost(const int *pcm, const long long *off, const int *nn, long long *out, unsigned *scratch, const long long *wide) {
  __shared__ unsigned bits[kSparseWords];
  __shared__ int pre[kSparseWords + 1];
  __shared__ int s_i[4];
  const int b = blockIdx.x;
  const int *x = pcm + off[b];
  const int n = nn[b];
  long long *o = out + 4LL * b;
  if (n <= 0) { if (threadIdx.x == 0) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; } return; }
  int mn = 2147483647, mx = -2147483647 - 1;
  for (int i = threadIdx.x; i < n; i += 256) { const int v = x[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  mn = block_reduce(mn, s_i, [](int a, int c) { return a < c ? a : c; });
  mx = block_reduce(mx, s_i, [](int a, int c) { return a > c ? a : c; });
  const long long range = (long long)mx - mn + 1;
  if (!wide) {
    if (range > kSparseMaxRange) { if (threadIdx.x == 0) { o[0] = 0; o[1] = 0; o[2] = range; o[3] = -1; } return; }   // -> second pass
    const int N = (int)range, nw = (N + 31) >> 5;
    for (int i = threadIdx.x; i < nw; i += 256) bits[i] = 0u;
    __syncthreads();
    sparse_cost_body(x, n, mn, N, bits, pre, o);
  } else {
    const long long nw = (range + 31) >> 5;
    if (range > 0x7fffffe0LL || 2 * nw + 1 > wide[2 * b + 1]) { if (threadIdx.x == 0) { o[0] = 0; o[1] = 0; o[2] = range; o[3] = -2; } return; }
    unsigned *gb = scratch + wide[2 * b];
    for (long long i = threadIdx.x; i < nw; i += 256) gb[i] = 0u;
    __threadfence_block();
    __syncthreads();
    sparse_cost_body(x, n, mn, (int)range, gb, reinterpret_cast<int *>(gb + nw), o);
  }
}
void launch_sparse_cost(hipStream_t s, const int *d_pcm, const long long *d_off, const int *d_n, int count, long long *d_out4,
                        unsigned *d_scratch, const long long *d_wide) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_sparse_cost, dim3(count), dim3(256), 0, s, d_pcm, d_off, d_n, d_out4, d_scratch, d_wide);
}

// ------------------------------------------------------------------ cost functions (cost.h)
constexpr int kHistGlobal = 1 << 18;   // residual range of 16-bit material: < 2^17+1
size_t cost_hist_scratch_ints() { return kHistGlobal; }

__device__ __forceinline__ int s2u_dev(int v) { return v < 0 ? 2 * (-v) : (v > 0 ? 2 * v - 1 : 0); }

// one block (256) per residual vector.  L1 (cost.h:15-26), RMS (:28-39), Entropy (:70-116).
// Entropy: order-0 histogram; the reference adds the per-bin terms sequentially, here each
// thread adds its bins in order and the partials are combined in a fixed tree -> deterministic,
// ~1e-15 relative from the reference's sequential sum (tolerance 1e-12 in the tests).
// hist_off / hist_cap (nullable): start and capacity (ints) of vector b's histogram in hist_scratch; default b * kHistGlobal
// and kHistGlobal.  A residual range beyond the capacity (material wider than 16 bits) is not clamped: cost[b] = -(range) tells
// the host to run that vector again with a histogram of its own size (entropy is never negative).
__global__ __launch_bounds__(256) void k_cost(int kind, const int *err, const long long *off, const int *nn, int *hist_scratch, double *cost,
                                              const long long *hist_off, const long long *hist_cap) {
  __shared__ long long s_ll[4];
  __shared__ int s_i[4];
  __shared__ double s_d[4];
  const int b = blockIdx.x;
  const int *e = err + off[b];
  const int n = nn[b];
  if (n <= 0) { if (threadIdx.x == 0) cost[b] = 0.0; return; }
  if (kind == 0 || kind == 1) {
    long long sum = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int v = e[i];
      sum += (kind == 0) ? (long long)(v < 0 ? -(long long)v : v) : (long long)(int)((unsigned)v * (unsigned)v);
    }
    sum = block_reduce(sum, s_ll, [](long long a, long long c) { return a + c; });
    if (threadIdx.x == 0) cost[b] = (kind == 0) ? (double)sum / (double)n : sqrt((double)sum / (double)n);
    return;
  }
  // entropy
  int mn = 2147483647, mx = -2147483647 - 1;
  for (int i = threadIdx.x; i < n; i += 256) { const int v = e[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  mn = block_reduce(mn, s_i, [](int a, int c) { return a < c ? a : c; });
  mx = block_reduce(mx, s_i, [](int a, int c) { return a > c ? a : c; });
  const long long range = (long long)mx - mn + 1;
  int *hist = hist_scratch + (hist_off ? (size_t)hist_off[b] : (size_t)b * kHistGlobal);
  const long long cap = hist_cap ? hist_cap[b] : (long long)kHistGlobal;
  if (range > cap) { if (threadIdx.x == 0) cost[b] = -(double)range; return; }
  const int nb = (int)range;
  for (int i = threadIdx.x; i < nb; i += 256) hist[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) atomicAdd(&hist[(long long)e[i] - mn], 1);
  __syncthreads();
  const double invs = 1.0 / (double)n;
  double ent = 0.0;
  // contiguous chunk of bins per thread, in ascending order
  const int per = (nb + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = (b0 + per < nb) ? b0 + per : nb;
  for (int i = b0; i < b1; i++) {
    const int c = hist[i];
    if (c == 0) continue;
    const double p = c * invs;
    ent = fma((double)c, log2(p), ent);
  }
  ent = block_reduce(ent, s_d, [](double a, double c) { return a + c; });
  if (threadIdx.x == 0) cost[b] = -ent / 8.0;
}

// CostGolomb (cost.h:43-66): inherently serial running mean -> one lane per vector
__global__ void k_cost_golomb(const int *err, const long long *off, const int *nn, int count, double *cost) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= count) return;
  const int *e = err + off[b];
  const int n = nn[b];
  double rm = 0.0; long long nbits = 0;
  for (int i = 0; i < n; i++) {
    int m = (int)rm; if (m < 1) m = 1;
    const int uval = s2u_dev(e[i]);
    nbits += uval / m + 1;
    if (m > 1) nbits += 32 - __clz(m);
    rm = fma(0.97, rm, (double)uval);
  }
  cost[b] = n ? nbits / 8. : 0.0;
}

void launch_cost(hipStream_t s, int kind, const int *d_err, const long long *d_off, const int *d_n, int count, int *d_hist, double *d_cost,
                 const long long *d_hist_off, const long long *d_hist_cap) {
  if (count <= 0) return;
  if (kind == 3) hipLaunchKernelGGL(k_cost_golomb, dim3((count + 63) / 64), dim3(64), 0, s, d_err, d_off, d_n, count, d_cost);
  else hipLaunchKernelGGL(k_cost, dim3(count), dim3(256), 0, s, kind, d_err, d_off, d_n, d_hist, d_cost, d_hist_off, d_hist_cap);
}

// ------------------------------------------------------------------ S2U + maxbpn (libsac.cpp:429-441)
__global__ __launch_bounds__(256) void k_s2u(const int *err, int *s2u, const long long *off, const int *nn, int *maxbpn) {
  __shared__ int s_i[4];
  const int b = blockIdx.x;
  const int *e = err + off[b];
  int *u = s2u + off[b];
  const int n = nn[b];
  int mx = 0;
  for (int i = threadIdx.x; i < n; i += 256) { const int v = s2u_dev(e[i]); u[i] = v; mx = v > mx ? v : mx; }
  mx = block_reduce(mx, s_i, [](int a, int c) { return a > c ? a : c; });
  if (threadIdx.x == 0) { int nb = 0; int v = mx; while (v >>= 1) nb++; maxbpn[b] = nb; }
}
void launch_s2u(hipStream_t s, const int *d_err, int *d_s2u, const long long *d_off, const int *d_n, int count, int *d_maxbpn) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_s2u, dim3(count), dim3(256), 0, s, d_err, d_s2u, d_off, d_n, d_maxbpn);
}

}  // namespace sacamd
