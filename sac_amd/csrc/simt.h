// sac_amd/csrc/simt.h -- minimal SIMT executor abstraction.
//
// Kernel bodies are written once, as templates over an executor E:
//   E::Reg<T> r;          per-lane register            r[lane]
//   ex.par(f)             run f(lane) on every lane     (no implicit barrier)
//   ex.sync()             workgroup barrier
//   ex.allsum(r)          butterfly all-reduce within 64-lane waves, then waves in order
//   ex.lane_get(r, k)     value of lane k's register (uniform k)
// ExecDev<NL> maps this onto a real gfx950 workgroup of NL threads; ExecEmu<NL> (host only,
// used by the CPU logic tests) runs the lanes one after another, with exactly the same
// floating-point operation order, so the kernels' results can be checked without a GPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SA_HD __host__ __device__ __forceinline__
#define SA_D __device__ __forceinline__
#else
#define SA_HD inline
#endif

// hide an integer's value from the optimiser (device: it lives in one VGPR; host: no-op)
#if defined(__HIP_DEVICE_COMPILE__)
#define SA_OPAQUE_INT(x) asm volatile("" : "+v"(x))
#define SA_OPAQUE_SINT(x) asm volatile("" : "+s"(x))      // same for a wave-uniform value held in an SGPR
#define SA_PIN_F64(x) asm volatile("" : "+v"(x))          // the computation of x stays in this basic block (not sunk / merged across branches)
#define SA_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0) // the instruction scheduler moves nothing across this point (software-pipelined straight-line code)
#else
#define SA_SCHED_FENCE() ((void)0)
#define SA_PIN_F64(x) ((void)0)
#define SA_OPAQUE_INT(x) ((void)0)
#define SA_OPAQUE_SINT(x) ((void)0)
#endif

namespace sacamd {

// progress counters between a producer and a consumer KERNEL running at the same time (final pass: the cascade
// follows the OLS stage chunk by chunk): release store / acquire load at device scope
SA_HD void sa_publish(int *p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}
SA_HD int sa_acquire(const int *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
SA_HD void sa_backoff() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_sleep(32);
#endif
}

// ---- decoder: per-sample hand-offs between the three stage kernels of a channel (and its stereo partner), which run at
// the same time: the decoder only learns a sample after predicting it (FrameCoder::UnpredictFrame, libsac.cpp:144-199),
// so the stages cannot run one after the other as in the encoder.  Counters are release-stored by their producer and
// acquire-loaded by their consumers (agent scope); data is written before the counter and read after it.
struct DecLink {
  int *self_w;               // bias stage: the channel's decoded (centred) samples -- the `self` input of the other two
  const int *prog_self;      // samples of this channel decoded so far
  const int *prog_other;     // ... of the stereo partner (OLS regressor; == prog_self for mono)
  const int *prog_in;        // values the upstream stage has produced (cascade: p_lpc; bias: p_lpc + p_lms)
  int *prog_out;             // values this stage has produced (OLS: p_lpc; cascade: p_lpc + p_lms; bias: decoded samples)
  int *fail;                 // set by whoever waited too long (a partner kernel is not running): all waits end
  const int *merr;           // bias stage: the entropy-decoded (mapped) residuals
  const int *prefix;         // bias stage, mapped streams: prefix[j] = number of used values in [-32768, -32768 + j - 1]; else null
};
// true when *p >= need; false when the link has failed.  The wait is bounded by a number of polls (about a microsecond each:
// s_sleep + two L2 round trips -> a few seconds), NOT by s_memtime differences: when the device's hardware queues are
// oversubscribed (a second process with its own queues on the same GPU) waves are saved and restored by the queue scheduler,
// and a wave restored elsewhere read a time base that made `now - t0` jump past any bound -- every wait "timed out" at once
// and the decoder failed beside an idle HIP process although it ran alone (profiles/r03/README.md).
SA_HD bool sa_wait_ge(const int *p, int need, int *fail) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
  for (unsigned polls = 0;; polls++) {
    __builtin_amdgcn_s_sleep(8);
    if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    if (polls > 4000000u) { __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
#else
  (void)fail;
  return *p >= need;
#endif
}

// ------------------------------------------------------------------ emulation executor
template <int NL>
struct ExecEmu {
  static constexpr int nl = NL;
  static constexpr bool is_device = false;
  template <class T> struct Reg {
    T v[NL];
    T &operator[](int l) { return v[l]; }
    const T &operator[](int l) const { return v[l]; }
  };
  template <class F> void par(F &&f) { for (int l = 0; l < NL; l++) f(l); }
  // code that every lane executes with identical (uniform) operands: run once
  template <class F> void uni(F &&f) { f(); }
  // code only wave 0 executes (uniformly): run once
  template <class F> void leader(F &&f) { f(); }
  void sync() {}
  void gsync() {}     // sync() that also orders the group's GLOBAL memory traffic (stores visible to its own later loads)
  template <class T> T lane_get(const Reg<T> &r, int k) { return r[k]; }
  double lane_bcast(const Reg<double> &r, int k) { return r[k]; }
  template <class R> double lane_bcast_col(const R &r, int col, int k) { return r[k].v[col]; }   // element `col` of lane k's register array
  template <int K> double lane_bcast_at(const Reg<double> &r) { return r[K]; }                   // lane_bcast, K a compile-time lane, not moved by the compiler
  // inside par(): the value lane k (0..63) of lane l's own wave holds
  double wave_lane(const Reg<double> &r, int l, int k) { return r[(l & ~63) | k]; }
  int lane_geti(const Reg<int> &r, int k) { return r[k]; }
  // groups of GL lanes: every lane takes the value lane k of ITS group holds (k uniform, < GL)
  template <int GL> void grp_bcast(Reg<double> &dst, const Reg<double> &src, int k) { for (int l = 0; l < NL; l++) dst[l] = src[(l / GL) * GL + k]; }
  template <int GL, class R> void grp_bcast_col(Reg<double> &dst, const R &src, int col, int k) { for (int l = 0; l < NL; l++) dst[l] = src[(l / GL) * GL + k].v[col]; }
  // every lane takes the value of lane-1 (lane 0 keeps its own): DPP wave_shr:1 on the device
  void shift_up1(Reg<double> &r) { for (int l = NL - 1; l > 0; l--) r[l] = r[l - 1]; }
  // the same inside wave w only (its lane 0 keeps its value; the other waves' registers are untouched)
  void wave_shift_up1(int w, Reg<double> &r) { for (int l = 64 * w + 63; l > 64 * w; l--) r[l] = r[l - 1]; }
  // uniform code: fma(value lane Q of the calling lane's row of 16 lanes holds in r, w, s) resp. that value itself.  The device runs
  // uniform code on all lanes, every row of 16 holding the same 16 values (the emulator's single instance stands for row 0).
  template <int Q> double row_bcast_fma(const Reg<double> &r, double w, double s) { return fma(r[Q], w, s); }
  template <int Q> double row_bcast(const Reg<double> &r) { return r[Q]; }
  // trip count of a loop whose length differs between the waves of a workgroup: lanes < split run a iterations, the others
  // b (device: the calling wave's own count, wave-uniform; the emulator runs the lanes of all waves in one loop)
  int wave_hops(int, int a, int b) { return a > b ? a : b; }
  // a value every lane of the wave holds alike, made known as such to the compiler (device: v_readfirstlane -> SGPR, so that
  // branches on it are scalar branches)
  static int uniform(int v) { return v; }
  // code only lane 0 executes: run once
  template <class F> void lane0(F &&f) { f(); }
  static unsigned long long clock() { return 0; }
  static bool is_lane0() { return true; }     // inside uni(): the single emulated instance stands for lane 0
  static bool is_leader() { return true; }    // top-level code of wave 0 (the emulator runs top-level code once)
  static bool is_lane0w() { return true; }    // inside wave(): lane 0 of that wave
  // per-lane code of wave 0 only; wsync orders LDS traffic inside one wave
  template <class F> void leader_par(F &&f) { for (int l = 0; l < (NL < 64 ? NL : 64); l++) f(l); }
  // uniform code / per-lane code of wave w only; the lane index handed to f is the workgroup-wide
  // one (64*w + lane) so that registers are addressed the same way everywhere
  template <class F> void wave(int, F &&f) { f(); }
  template <class F> void wave_par(int w, F &&f) { for (int l = 0; l < 64; l++) f(64 * w + l); }
  void wsync() {}
  // butterfly sum of K values per lane within each 64-lane wave (all lanes get the wave total)
  template <int K, class R> void wave_sum(R &r) {
    constexpr int W = NL < 64 ? NL : 64;
    for (int q = 0; q < K; q++)
      for (int d = W / 2; d >= 1; d >>= 1) {
        double t[NL];
        for (int l = 0; l < NL; l++) t[l] = r[l].v[q] + r[(l & ~(W - 1)) | ((l & (W - 1)) ^ d)].v[q];
        for (int l = 0; l < NL; l++) r[l].v[q] = t[l];
      }
  }
  // transpose-reduce of 8 values per lane within each 64-lane wave: afterwards lane 16*r of the
  // wave (r = 0..3) holds the wave totals of values 2r and 2r+1 in v[0], v[1] (other lanes and
  // slots: unspecified).  Summation tree, per value, over the wave's lanes:
  //   b[i] = (x[i] + x[i+32]) + (x[i+16] + x[i+48]),  i < 16
  //   c[i] = b[i] + b[(i+8) % 16];  d[i] = c[i] + c[7-i];  total = (d[0] + d[1]) + (d[2] + d[3])
  template <class R> void wave_sum8x(R &r) {
    static_assert(NL % 64 == 0, "whole waves");
    for (int w = 0; w < NL / 64; w++) {
      double tot[8];
      for (int q = 0; q < 8; q++) {
        double b[16], c[8], d[4];
        for (int i = 0; i < 16; i++) {
          const int l = w * 64 + i;
          b[i] = (r[l].v[q] + r[l + 32].v[q]) + (r[l + 16].v[q] + r[l + 48].v[q]);
        }
        for (int i = 0; i < 8; i++) c[i] = b[i] + b[i + 8];
        for (int i = 0; i < 4; i++) d[i] = c[i] + c[7 - i];
        tot[q] = (d[0] + d[1]) + (d[2] + d[3]);
      }
      for (int rr = 0; rr < 4; rr++) { r[w * 64 + 16 * rr].v[0] = tot[2 * rr]; r[w * 64 + 16 * rr].v[1] = tot[2 * rr + 1]; }
    }
  }
  // butterfly within waves of 64 (or NL if smaller), then sequential over waves
  void allsum(Reg<double> &r, double *scratch /*>= NL/64 doubles*/) {
    constexpr int W = NL < 64 ? NL : 64;
    for (int d = W / 2; d >= 1; d >>= 1) {
      double t[NL];
      for (int l = 0; l < NL; l++) t[l] = r[l] + r[(l & ~(W - 1)) | ((l & (W - 1)) ^ d)];
      for (int l = 0; l < NL; l++) r[l] = t[l];
    }
    if (NL > 64) {
      double s = r[0];
      for (int w = 1; w < NL / 64; w++) s = s + r[w * 64];
      for (int l = 0; l < NL; l++) r[l] = s;
    }
    (void)scratch;
  }
};

#if defined(__HIPCC__)
// ------------------------------------------------------------------ device executor
template <int NL>
struct ExecDev {
  static constexpr int nl = NL;
  static constexpr bool is_device = true;
  // Wave roles can be rotated: lane index l = (threadIdx.x + 64 * rot) mod NL, so that "wave 0" of the kernel body (the wave
  // that runs a serial chain while the others wait at a barrier) is hardware wave (NL/64 - rot) mod NL/64.  Measured in round 4
  // on the search cascade (rot = block index, and block/8 + block/256): no change in saturated throughput to three digits
  // (profiles/r04/throughput_lms_rot*.txt) -- the serial chains of co-resident workgroups do not pile up on one SIMD.  Unused.
  int base_ = 0;
  SA_D ExecDev() {}
  SA_D explicit ExecDev(int rot_waves) : base_(64 * rot_waves) {}
  SA_D int tid() const { return ((int)threadIdx.x + base_) & (NL - 1); }
  template <class T> struct Reg {
    T v;
    SA_D T &operator[](int) { return v; }
    SA_D const T &operator[](int) const { return v; }
  };
  template <class F> SA_D void par(F &&f) { f(tid()); }
  template <class F> SA_D void uni(F &&f) { f(); }
  template <class F> SA_D void leader(F &&f) { if (tid() < 64) f(); }
  SA_D void sync() { __syncthreads(); }
  template <int K, class R> SA_D void wave_sum(R &r) {
    constexpr int W = NL < 64 ? NL : 64;
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
#pragma unroll
      for (int q = 0; q < K; q++) r.v.v[q] = r.v.v[q] + __shfl_xor(r.v.v[q], d, 64);
    }
  }
  // see ExecEmu::wave_sum8x.  v_permlane32_swap / v_permlane16_swap halve the value count while
  // folding lane halves / 16-lane rows; the last four levels are DPP moves inside a row.
  template <int CTRL> static SA_D double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  template <class R> SA_D void wave_sum8x(R &r) {
    double *v = r.v.v;
#pragma unroll
    for (int j = 0; j < 4; j++) {     // A' = [A.lo | B.lo], B' = [A.hi | B.hi]
      const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v[j]), (unsigned)__double2loint(v[j + 4]), false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v[j]), (unsigned)__double2hiint(v[j + 4]), false, false);
      v[j] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {     // A' = rows [A0 B0 A2 B2], B' = rows [A1 B1 A3 B3]
      const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v[j]), (unsigned)__double2loint(v[j + 2]), false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v[j]), (unsigned)__double2hiint(v[j + 2]), false, false);
      v[j] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      v[j] = v[j] + dpp_f64<0x128>(v[j]);   // row_ror:8
      v[j] = v[j] + dpp_f64<0x141>(v[j]);   // row_half_mirror
      v[j] = v[j] + dpp_f64<0xB1>(v[j]);    // quad_perm [1,0,3,2]
      v[j] = v[j] + dpp_f64<0x4E>(v[j]);    // quad_perm [2,3,0,1]
    }
  }
  template <class T> SA_D T lane_get(const Reg<T> &r, int k) { return __shfl(r.v, k, 64); }
  SA_D int lane_geti(const Reg<int> &r, int k) { return __builtin_amdgcn_readlane(r.v, k); }
  // lane k of the calling lane's group of GL lanes (one wave): ds_bpermute_b32 (the LDS crossbar, no memory access)
  template <int GL> static SA_D double grp_pick(double x, int k) {
    const int src = (((int)threadIdx.x & 63 & ~(GL - 1)) + k) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x));
    return __hiloint2double(hi, lo);
  }
  template <int GL> SA_D void grp_bcast(Reg<double> &dst, const Reg<double> &src, int k) { dst.v = grp_pick<GL>(src.v, k); }
  template <int GL, class R> SA_D void grp_bcast_col(Reg<double> &dst, const R &src, int col, int k) { dst.v = grp_pick<GL>(src.v.v[col], k); }
  SA_D void shift_up1(Reg<double> &r) {
    int lo = __double2loint(r.v), hi = __double2hiint(r.v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    r.v = __hiloint2double(hi, lo);
  }
  SA_D void wave_shift_up1(int w, Reg<double> &r) { if ((tid() >> 6) == w) shift_up1(r); }
  // DP-ALU DPP (gfx90a+): src0 of the instruction = lane Q of the reading lane's row of 16 lanes; ONE instruction per term of a
  // serial fp64 chain whose operands arrive sixteen to a register (pred_ols_grid.h: backward substitution).  All lanes must be
  // active (a disabled source lane is not read).
  template <int Q> static SA_D double row_bcast_fma(const Reg<double> &r, double w, double s) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(r.v), "v"(w), "n"(Q));
    return s;
  }
  template <int Q> static SA_D double row_bcast(const Reg<double> &r) {
    double z;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(z) : "v"(r.v), "n"(Q));
    return z;
  }
  static SA_D int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
  SA_D int wave_hops(int split, int a, int b) { return __builtin_amdgcn_readfirstlane(tid() < split ? a : b); }
  template <class F> SA_D void lane0(F &&f) { if (tid() == 0) f(); }
  static SA_D unsigned long long clock() { return __builtin_readcyclecounter(); }
  SA_D bool is_lane0() const { return tid() == 0; }
  SA_D bool is_leader() const { return tid() < 64; }
  static SA_D bool is_lane0w() { return (threadIdx.x & 63) == 0; }
  template <class F> SA_D void leader_par(F &&f) { if (tid() < 64) f(tid()); }
  template <class F> SA_D void wave(int w, F &&f) { if ((tid() >> 6) == w) f(); }
  template <class F> SA_D void wave_par(int w, F &&f) { if ((tid() >> 6) == w) f(tid()); }
  SA_D void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  // value of lane k (k uniform across the wave) -> scalar broadcast via v_readlane_b32
  SA_D double lane_bcast(const Reg<double> &r, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(r.v), k & 63);   // k may be a workgroup-wide lane index
    const int hi = __builtin_amdgcn_readlane(__double2hiint(r.v), k & 63);
    return __hiloint2double(hi, lo);
  }
  template <class R> SA_D double lane_bcast_col(const R &r, int col, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(r.v.v[col]), k & 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(r.v.v[col]), k & 63);
    return __hiloint2double(hi, lo);
  }
  // v_readlane as volatile inline assembly: stays where it is written (the compiler otherwise hoists the broadcasts of an
  // unrolled column loop to its top and spills the scalar registers they occupy)
  template <int K> SA_D double lane_bcast_at(const Reg<double> &r) {
    int lo, hi;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(lo) : "v"(__double2loint(r.v)), "n"(K));
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(hi) : "v"(__double2hiint(r.v)), "n"(K));
    return __hiloint2double(hi, lo);
  }
  SA_D double wave_lane(const Reg<double> &r, int, int k) { return lane_bcast(r, k); }
  SA_D void allsum(Reg<double> &r, double *scratch) {
    constexpr int W = NL < 64 ? NL : 64;
    double v = r.v;
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, 64);
    if (NL > 64) {
      const int w = tid() >> 6;
      if ((threadIdx.x & 63) == 0) scratch[w] = v;
      __syncthreads();
      double s = scratch[0];
#pragma unroll
      for (int i = 1; i < NL / 64; i++) s = s + scratch[i];
      v = s;
      __syncthreads();
    }
    r.v = v;
  }
};
// One wave of a larger workgroup acting on its own (64 lanes, lane = threadIdx.x & 63).  The
// workgroup's waves are independent streams: sync() only orders the wave's own LDS traffic.
struct ExecDevWave {
  static constexpr int nl = 64;
  static constexpr bool is_device = true;
  template <class T> struct Reg {
    T v;
    SA_D T &operator[](int) { return v; }
    SA_D const T &operator[](int) const { return v; }
  };
  template <class F> SA_D void par(F &&f) { f((int)(threadIdx.x & 63)); }
  SA_D void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  SA_D void gsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __builtin_amdgcn_wave_barrier(); }
  SA_D void wsync() { sync(); }
  SA_D int lane_geti(const Reg<int> &r, int k) { return __builtin_amdgcn_readlane(r.v, k); }
  template <class F> SA_D void lane0(F &&f) { if ((threadIdx.x & 63) == 0) f(); }
};
#endif


}  // namespace sacamd
