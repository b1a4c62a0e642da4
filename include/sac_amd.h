/* include/sac_amd.h -- C ABI of libsac_amd.so: the MI355X (gfx950) implementation of Sac's
 * per-frame encode hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (slmdev/sac v0.7.25, /root/reference) has no FFI: its seam is the C++ class
 * surface of libsac.  Each entry point below names the reference interface it stands in for.
 * A FrameCoder-shaped C++ wrapper over this ABI is in sac_amd/csrc/framecoder.h and the binding
 * a maintainer would add to the reference is shown in INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success or a negative sacamd_status; nothing throws
 * across the ABI; the caller owns all host buffers; the context owns all device buffers and its main
 * HIP stream; a context is single-submitter (concurrency is expressed by batching frames and
 * candidates).  Distinct contexts hold independent state and may be driven from different threads; on one
 * device they share a pool of 13 side streams (the hardware runs a limited number of queues at once; a second PROCESS with its
 * own queues on the same GPU oversubscribes them, which costs time but no longer correctness: profiles/r03/README.md) and
 * take turns with their search phases (one search saturates the chip).  There is no CPU fallback: if no
 * gfx950 device/kernel image is available, sacamd_ctx_create fails with SACAMD_ERR_NOGPU.
 *
 * Hardware queues: HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read at the process's first HIP call.  The
 * library sets it to 24 when it is loaded unless the environment already has it.  sacamd_ctx_create warns once on stderr (round 6; it
 * used to refuse) when the caller's own value is below 16, and when the library had to set the variable although a HIP-using module
 * (torch) was loaded first -- HIP may have read the default already.  A host that initialises HIP before loading the library exports it itself.
 *
 * Environment switches (read once per process; every one is an A/B or diagnostic aid, the defaults are the product):
 *   SACAMD_TRACE=1             every predictor launch with its duration and start offset on stderr; at context destruction the search
 *                              cascade's work by tap count.  SACAMD_DUMP_VN=file (with TRACE) writes the stage lengths of every search item.
 *   SACAMD_OLS_GRID=0          33..64-tap OLS items on the round-4 KERNELS (one-wave row layout in the search, four-wave panel kernel in
 *                              the final pass: SACAMD_OLS_FINAL_PANEL=0 keeps the one-wave kernel there too) instead of k_ols_grid.  Not the
 *                              round-4 SCHEDULE: the panel slot budget and the host-side head start of run_predict went with round 5, so every
 *                              33..64-tap class of the final pass starts on the panel kernel at once (the round-3 set-up) -- a kernel A/B only.
 *   SACAMD_OLS_GRID_SHORT=0    17..32-tap items on the packed kernels everywhere.   SACAMD_OLS_PACK=0: one item per wave up to 32 taps.
 *   SACAMD_FAST_OLS=n          search: OLS classes [0, n) form the cascade's first launch group (default 3).
 *   SACAMD_FINAL_GROUPS=1      final pass: one cascade launch group per OLS class (measured as a loss).
 *   SACAMD_CODER_SERIAL=1      parity tap: the coder's decision chain on one lane (the body the CPU emulation runs).
 *   SACAMD_DEC_SINGLE=1        decoder: every frame group as one cooperative launch (the fallback form).  SACAMD_DEC_ZERO=0 skips zeroing its planes.
 * (Rounds 2-4 had more: tail priorities, chase mode, pipelining, the panel-kernel slot budget and head start -- all measured, documented in
 *  DESIGN.md 9 and removed.)
 */
#ifndef SAC_AMD_H
#define SAC_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SACAMD_NUM_COEFS 58 /* SacProfile::LoadBaseProfile, libsac/profile.cpp:9 */

typedef enum sacamd_status {
  SACAMD_OK = 0,
  SACAMD_ERR_ARG = -1,      /* bad argument / out of declared capacity */
  SACAMD_ERR_HIP = -2,      /* HIP runtime error (see sacamd_last_error) */
  SACAMD_ERR_NOGPU = -3,    /* no usable gfx950 device */
  SACAMD_ERR_STATE = -4,    /* call order violated (e.g. evaluate before analyse) */
  SACAMD_ERR_NONFINITE = -5,/* a predictor produced a non-finite value (cascade.h:40-41) */
  SACAMD_ERR_COMM = -6      /* RCCL error in the record gather (see sacamd_comm_last_error) */
} sacamd_status;

/* FrameCoder::SearchCost / SearchMethod, libsac/libsac.h:14-15 */
enum { SACAMD_COST_L1 = 0, SACAMD_COST_RMS = 1, SACAMD_COST_ENTROPY = 2, SACAMD_COST_GOLOMB = 3, SACAMD_COST_BITPLANE = 4 };
enum { SACAMD_SEARCH_DDS = 0, SACAMD_SEARCH_DE = 1, SACAMD_SEARCH_CMA = 2 };

/* FrameCoder::tsac_cfg + toptim_cfg (+ OptDDS::DDSCfg), libsac/libsac.h:19-44, opt/dds.h:12-19,
 * flattened.  Presets: cmdline.cpp:127-156. */
typedef struct sacamd_cfg {
  int optimize;      /* tsac_cfg.optimize */
  int sparse_pcm;    /* tsac_cfg.sparse_pcm (default 1) */
  int zero_mean;     /* tsac_cfg.zero_mean (default 1) */
  int reset;         /* toptim_cfg.reset  (--opt-reset) */
  double fraction;   /* toptim_cfg.fraction */
  int maxnfunc;      /* toptim_cfg.maxnfunc == DDSCfg.nfunc_max */
  int num_threads;   /* toptim_cfg.num_threads == DDSCfg.num_threads: 0 = sequential run_single,
                        N>0 = run_mt with N candidates per generation (--opt-cfg=dds,N) */
  double sigma;      /* DDSCfg.sigma_init */
  int optk;          /* toptim_cfg.optk (default 4) */
  int optimize_cost; /* SACAMD_COST_* */
  int optimize_search; /* toptim_cfg.optimize_search, SACAMD_SEARCH_*: --opt-cfg=dds|de|cma (cmdline.cpp:195-207).  DE
                          (opt/de.cpp: population 30, current-to-pbest/1/bin) and CMA (opt/cma.cpp: (1+1)-CMA-ES) take
                          maxnfunc and sigma as DDS does (cmdline.cpp:221-241); num_threads only sets how the reference
                          spreads evaluations over threads and does not change their results.  (ABI version 3.) */
} sacamd_cfg;

typedef struct sacamd_ctx sacamd_ctx;

/* ---- context -------------------------------------------------------------------------
 * Replaces: FrameCoder::FrameCoder(numchannels, framesize, cfg) buffer ownership
 * (libsac.cpp:13-35), for a BATCH of max_frames frames of <= max_framesize samples/channel. */
int sacamd_ctx_create(int device, int nch, int max_framesize, int max_frames, sacamd_ctx **out);
void sacamd_ctx_destroy(sacamd_ctx *ctx);
const char *sacamd_last_error(const sacamd_ctx *ctx);
int sacamd_default_profile(float *vmin, float *vmax, float *vdef); /* profile.cpp:3-89; each [58] */
void sacamd_default_cfg(sacamd_cfg *cfg);                          /* libsac.h:19-44 defaults */

/* ---- (1) frame staging ------------------------------------------------------------------
 * Replaces: the caller filling FrameCoder::samples[ch][0..n) + SetNumSamples
 * (libsac.cpp:822-825).  `framesize` is the reference's max frame size (max_framelen*rate,
 * libsac.cpp:784) used for the search-window length.  Host or device pointers. */
int sacamd_frames_upload_i32(sacamd_ctx *ctx, int nframes, int framesize, const int32_t *pcm_planar,
                             long long frame_stride, long long ch_stride, const int *numsamples);
/* interleaved little-endian int16 (L R L R ...), frame f starts at sample-frame frame_offset[f] */
int sacamd_frames_upload_s16(sacamd_ctx *ctx, int nframes, int framesize, const int16_t *pcm_interleaved,
                             const long long *frame_offset, const int *numsamples);
/* same, but pcm_interleaved is a DEVICE pointer already resident in HBM (no copy) */
int sacamd_frames_attach_s16_device(sacamd_ctx *ctx, int nframes, int framesize, const int16_t *d_pcm_interleaved,
                                    const long long *frame_offset, const int *numsamples);

/* ---- (2) analyse --------------------------------------------------------------------------
 * Replaces: the head of FrameCoder::Predict (libsac.cpp:445-459): AnalyseMonoChannel
 * (mean/min/max, :626-651), Remap::Analyse (map.cpp:126-157), mean removal. */
int sacamd_analyse(sacamd_ctx *ctx, const sacamd_cfg *cfg);
int sacamd_get_stats(sacamd_ctx *ctx, int32_t *out /* [nframes][nch][4] = mean,min,max,numsamples */);

/* ---- (3) batched candidate evaluation ----------------------------------------------------
 * Replaces: the search objective cost_func(x) (libsac.cpp:389-397) == PredictFrame(optimize=
 * true) over the centred search window + GetCost, and its batch form Opt::eval_points_mt
 * (opt/opt.cpp:11-43), for ncand (frame, profile) pairs at once.  coefs are the float32-
 * narrowed profiles.  costs[i] is the sum over channels, in bytes (cost.h). */
int sacamd_evaluate(sacamd_ctx *ctx, const sacamd_cfg *cfg, int ncand, const int *cand_frame,
                    const float *coefs /* [ncand][58] */, double *costs /* [ncand] */);

/* Within one staged batch sacamd_evaluate never computes the same channel evaluation twice: the cost of
 * a channel's residual is a pure function of (frame, channels, window, that slot's predictor parameters),
 * and DDS candidates share most of them with their parent and with each other.  Identical evaluations
 * are looked up (memo cleared by every frames_upload / frames_attach / analyse, i.e. it never outlives
 * the staged PCM) or computed once; results are bit-identical to evaluating every candidate in full.
 * out2 = { channel evaluations requested, of those answered without recomputation } since the last reset. */
int sacamd_eval_stats(sacamd_ctx *ctx, long long *out2, int reset);

/* ---- (4) final prediction pass -------------------------------------------------------------
 * Replaces: PredictFrame(base_profile, error, 0, numsamples, false) + CnvError_S2U
 * (libsac.cpp:477-478, 429-441).  One profile per staged frame. */
int sacamd_predict_final(sacamd_ctx *ctx, const sacamd_cfg *cfg, const float *coefs /* [nframes][58] */);
/* FrameCoder public buffers error / pred / s2u_error, framestats[].maxbpn (libsac.h:54-56) */
int sacamd_get_residuals(sacamd_ctx *ctx, int frame, int32_t *error, int32_t *pred, int32_t *s2u,
                         int *maxbpn /* each [nch][numsamples]; any may be NULL */);

/* ---- (5) entropy coding -------------------------------------------------------------------
 * Replaces: FrameCoder::Encode (libsac.cpp:486-494): per channel EncodeMonoFrame (:253-278) =
 * BitplaneCoder + RangeCoderSH over s2u_error (:201-212), CalcRemapError (:230-251) and, when
 * the ratio is > 1.05, the MapEncoder + mapped variant (:214-228), keeping the smaller. */
int sacamd_encode(sacamd_ctx *ctx, const sacamd_cfg *cfg);
int sacamd_get_encoded(sacamd_ctx *ctx, int frame, int ch, uint8_t *out, int cap, int *len,
                       int *mapped, int *maxbpn);
/* Both coder variants of a channel after sacamd_encode, as the reference leaves them in FrameCoder::enc_temp1 (variant 0, Normal) and
 * enc_temp2 (variant 1, Mapped: coded only when the L1 ratio exceeded 1.05, libsac.cpp:253-278); *len = 0 for a variant that was not
 * coded.  And CalcRemapError's products (libsac.cpp:230-251, with sparse_pcm): s2u_error_map [nch][n] and framestats[].maxbpn_map. */
int sacamd_get_encoded_variant(sacamd_ctx *ctx, int frame, int ch, int variant, uint8_t *out, int cap, int *len, int *maxbpn);
int sacamd_get_residuals_map(sacamd_ctx *ctx, int frame, int32_t *s2u_map, int *maxbpn_map);

/* ---- (6) whole batch: Predict + Encode + WriteEncoded -------------------------------------
 * Replaces: the per-frame sequence FrameCoder::Predict(); Encode(); WriteEncoded()
 * (libsac.cpp:827-829, 443-479, 565-578) for every staged frame, including the search
 * (opt/dds.cpp, opt/de.cpp or opt/cma.cpp by cfg.optimize_search) run in lock-step generations across frames.
 * profiles_io [nframes][58]: in = search start point per frame (cfg.reset!=0: ignored, the base
 * profile is used, == --opt-reset); out = profile written into each record.
 * out receives the frame records back to back; rec_off[f]..rec_off[f+1] delimit frame f. */
int sacamd_encode_frames(sacamd_ctx *ctx, const sacamd_cfg *cfg, float *profiles_io, uint8_t *out,
                         long long cap, long long *rec_off /* [nframes+1] */);

/* The search alone.  Replaces: FrameCoder::Optimize (libsac.cpp:365-427, called from Predict, :461-476) with OptDDS
 * (opt/dds.cpp), OptDE (opt/de.cpp:77-164) or OptCMA (opt/cma.cpp:49-92) for every staged frame.  profiles_io [nframes][58]: in = start point (ignored with cfg.reset != 0),
 * out = the profile the search settled on.  Follow with sacamd_predict_final + sacamd_encode for the
 * Predict() / Encode() split of the reference's FrameCoder (sac_amd/csrc/framecoder.h). */
int sacamd_search_frames(sacamd_ctx *ctx, const sacamd_cfg *cfg, float *profiles_io);

/* The DDS search in instalments (ABI 6).  Replaces: the same FrameCoder::Optimize + OptDDS::run_mt / run_single (opt/dds.cpp:33-106) as
 * sacamd_search_frames, cut into calls of at most max_generations lock-step generations -- the --best preset (cmdline.cpp:127-156: 1000
 * evaluations of the CostBitplane objective, libsac/cost.h:144-176) is 125 generations of tens of seconds each at full frame size, more than
 * one call of a time-boxed job holds.  `state` is an opaque blob the caller keeps between calls (every frame's mt19937, best point and
 * cost, sigma, SSC counters, evaluation count; capacity: sacamd_search_state_bytes): *state_len = 0 starts a search (start point =
 * profiles_io, or the base profile with cfg.reset), otherwise the search goes on from the blob; on return *state_len = bytes written,
 * *done = 1 once maxnfunc evaluations are spent, profiles_io = the best profile so far.  The candidates, costs and the final profile are
 * those of one uninterrupted sacamd_search_frames (same frames staged, same cfg).  DDS only (SACAMD_ERR_ARG for DE / CMA). */
int sacamd_search_frames_resume(sacamd_ctx *ctx, const sacamd_cfg *cfg, float *profiles_io, int max_generations, uint8_t *state,
                                long long state_cap, long long *state_len, int *done);
long long sacamd_search_state_bytes(const sacamd_ctx *ctx);

/* ---- (7) adaptive sub-frame split -------------------------------------------------------------
 * Replaces: Codec::Analyse + AnalyseSparse + PushState (libsac/libsac.cpp:696-780) with SparsePCM::Analyse
 * (libsac/sparse.h:31-96): one read of samples_read samples per channel (planar int32, host memory, un-centred)
 * is cut into blocks of blocksamples; a block is "sparse" when the mean over channels of
 * sum|val| / sum|rank(val)| exceeds 1.35; runs of equal state become sub-frames, a run shorter than
 * min_frame_length is appended to its predecessor.  The reference calls it with blocksamples ==
 * min_frame_length == 3 * rate (libsac.cpp:812-816).  The block sums are computed on the GPU. */
typedef struct sacamd_subframe { int start, length, state; } sacamd_subframe;
int sacamd_plan_subframes(sacamd_ctx *ctx, const int32_t *pcm_planar, long long ch_stride, int nch, int samples_read,
                          int blocksamples, int min_frame_length, sacamd_subframe *out, int cap, int *count);
/* the PushState state machine alone (host only, no device work): block_state / block_len per block */
int sacamd_subframes_from_states(const int *block_state, const int *block_len, int nblocks, int min_frame_length,
                                 sacamd_subframe *out, int cap, int *count);

/* ---- (8) decode ------------------------------------------------------------------------------------
 * Replaces: per frame FrameCoder::ReadEncoded (libsac.cpp:580-594) + Decode (:496-505: RangeCoderSH + [MapEncoder +]
 * BitplaneCoder::Decode per channel) + UnpredictFrame (:144-199), i.e. the body of Codec::DecodeFile's frame loop
 * (:857-883), for nframes frame records at once.  recs holds the records back to back, rec_off[f]..rec_off[f+1]
 * delimit frame f (as sacamd_encode_frames writes them / as they lie in a .sac file behind its header).  Output: planar
 * int32 PCM, pcm_out[f*frame_stride + ch*ch_stride + i], numsamples_out[f] (nullable), profiles_out [nframes][58]
 * (nullable).  The context must have been created for the channel count and frame sizes of the records. */
int sacamd_decode_frames(sacamd_ctx *ctx, int nframes, int framesize, const uint8_t *recs, const long long *rec_off,
                         int32_t *pcm_out, long long frame_stride, long long ch_stride, int *numsamples_out, float *profiles_out);

/* ---- multi-GPU sharding (host only) ----------------------------------------------------------
 * Frames are independent units (with cfg.reset): owner[f] = rank that encodes frame f, assigned longest-first by the
 * caller's cost estimate (channels * (evaluations * search window + frame length)) to the least loaded of `world`
 * ranks.  Each rank stages its frames, runs sacamd_encode_frames and sends its records to rank 0 with
 * sacamd_gather_records (the only communication on this path). */
int sacamd_assign_frames(const double *cost, int nframes, int world, int *owner);

/* ---- cascade launch plan (host only) ------------------------------------------------------------
 * The planner sacamd_encode_frames / sacamd_evaluate use for the cascade launches of one predictor pass
 * (sac_amd/csrc/launch_plan.h), callable without a GPU.  Replaces: nothing in the reference, whose FrameCoder
 * evaluates one candidate at a time (libsac.cpp:323-344); the batch's launches are this library's own.
 * Launch q belongs to group[q] (a group waits for its own OLS classes) and has work[q] (taps x samples summed
 * over its items; 1e300 = the whole-CU layout); group g may use the streams pool[pool_begin[g] .. pool_begin[g+1])
 * (ids < 4096).  order[] = the launches in issue order, stream[q] = the stream of launch q: throughput-bound
 * launches (work >= small_work) longest first onto the least loaded stream, small ones onto streams without a
 * throughput-bound launch where the pool has one, balanced by their number.  SACAMD_ERR_ARG for a launch
 * whose group has no streams.  (ABI version 7.) */
int sacamd_plan_cascade_streams(int nlaunches, const int *group, const double *work, int ngroups, const int *pool_begin, const int *pool,
                                double small_work, int *order, int *stream);

/* ---- multi-GPU record gather (RCCL over xGMI) ----------------------------------------------------
 * Replaces: nothing in the single-process reference; what is gathered is what FrameCoder::WriteEncoded
 * (libsac.cpp:565-578) appends to the file, frame after frame, in Codec::EncodeFile's loop (:822-829) -- rank 0 ends up with
 * the records of all ranks in frame order and writes them behind the header exactly as the reference does.
 * One process per GPU.  Rank 0 calls sacamd_comm_unique_id and hands the 128 bytes to the other ranks by whatever
 * means the launcher offers (bench.py: the torch.distributed store); every rank then calls sacamd_comm_create
 * (ncclCommInitRank) for its device.  sacamd_gather_records is collective: every rank passes its nrec records
 * (back to back in recs, rec_off[i]..rec_off[i+1] delimit record i, as sacamd_encode_frames writes them) with their
 * global frame numbers frame_id[i]; the ranks' frame ids must partition 0..total_frames-1.  Rank 0 receives all
 * records in frame order in out (capacity cap bytes) with out_off[total_frames+1]; other ranks may pass NULL / 0.
 * Wire traffic: one all-gather of (count, bytes, status, rank 0's capacity), one all-gather of (frame, length) pairs, then ONE
 * group of ncclSend / ncclRecv (rank 0 posts a receive per peer inside a single ncclGroupStart/End, so its xGMI links fill
 * concurrently); payloads are staged through device buffers owned by the communicator.
 * Failure behaviour: no rank leaves the call alone.  Bad arguments on ANY rank (incl. rank 0's out / cap) are seen by all
 * ranks in the first all-gather and all return an error before a payload moves; a rank whose own work failed calls with
 * nrec = -1 and every rank returns SACAMD_ERR_COMM ("rank r reported a failure").  After a HIP / RCCL error under the gather
 * the communicator is aborted (ncclCommAbort), so that peers blocked in the collective return, and must be destroyed. */
typedef struct sacamd_comm sacamd_comm;
#define SACAMD_COMM_ID_BYTES 128
int sacamd_comm_unique_id(uint8_t *id128);
int sacamd_comm_create(int device, int rank, int world, const uint8_t *id128, sacamd_comm **out);
void sacamd_comm_destroy(sacamd_comm *comm);
const char *sacamd_comm_last_error(const sacamd_comm *comm);
int sacamd_gather_records(sacamd_comm *comm, int nrec, const int *frame_id, const uint8_t *recs, const long long *rec_off,
                          int total_frames, uint8_t *out, long long cap, long long *out_off);
/* The same gather over a caller-supplied transport (host buffers): MPI in a host application; the gloo transport of this
 * repository's CPU tests.  allgather_i64: recv[r*count + i] = rank r's send[i].  send / recv between group_begin and
 * group_end may complete as late as group_end; all of them are complete when group_end returns.  Callbacks return 0
 * or a negative status, which the gather passes on. */
typedef struct sacamd_transport {
  void *self;
  int rank, world;
  int (*allgather_i64)(void *self, const long long *send, long long *recv, int count);
  int (*group_begin)(void *self);
  int (*send)(void *self, int peer, const void *buf, long long bytes);
  int (*recv)(void *self, int peer, void *buf, long long bytes);
  int (*group_end)(void *self);
} sacamd_transport;
int sacamd_gather_records_via(const sacamd_transport *t, int nrec, const int *frame_id, const uint8_t *recs,
                              const long long *rec_off, int total_frames, uint8_t *out, long long cap, long long *out_off);

/* ---- parity taps (tests) -------------------------------------------------------------------
 * Per-stage streams of one frame for one profile: p_lpc, p_lpc+p_lms (file-channel order),
 * residual and pred, over window [start,start+n).  == oracle predict_trace. */
int sacamd_debug_predict(sacamd_ctx *ctx, int frame, const float *coefs, int start, int n,
                         int optimize, int optk, double *plpc, double *psum, int32_t *err, int32_t *pred);
/* bitplane coder on an arbitrary s2u vector (== BitplaneCoder::Encode + RangeCoderSH) */
int sacamd_debug_bitplane(sacamd_ctx *ctx, const int32_t *s2u, int n, int maxbpn, uint8_t *out, int cap, int *len);
/* the device's exp / pow (the glibc ports every kernel uses where the reference calls std::exp / std::pow) and the in-kernel
 * BitplaneCoder::PredictLaplace (vle.cpp:70-79) on arbitrary arguments.  kind 0: out = exp(x); 1: out = pow(x, y);
 * 2: out = PredictLaplace(avg_sum = (uint32)x, bpn = (int)y) */
int sacamd_debug_libm(sacamd_ctx *ctx, int kind, const double *x, const double *y, int n, double *out);
/* cost function on an arbitrary residual vector (== CostFunction::Calc, cost.h) */
int sacamd_debug_cost(sacamd_ctx *ctx, int kind, const int32_t *err, int n, double *cost);
/* time spent (ms, HIP events on the context's stream) in each kernel family since the last call:
 * [0] analyse [1] tables [2] ols [3] lms [4] bias [5] cost [6] s2u/remap [7] coder; launches in [8..15] */
int sacamd_kernel_times(sacamd_ctx *ctx, double *out16, int reset);

/* per kernel instance of the two heavy predictor stages, since the last reset: out[(kind*16 + slot)*4 + {0,1,2,3}] =
 * total ms (HIP events on the launch's stream), launches, item-steps processed, algorithmic fp64 flops (FMA = 2,
 * SURVEY.md 8d formula with each item's actual regressor length / tap counts).  kind 0 = OLS: slots 0..7 = capacity
 * classes 16,24,32,40,48,56,64,96 taps.  Which kernel a slot is (kernels_pred.hip: launch_ols, since round 5): slot 0
 * k_ols_pack<16,16>; slot 1 k_ols_pack<24,32> in the search, k_ols_grid<3> in the final pass; slot 2 k_ols_grid<4>;
 * slots 3..6 k_ols_grid<5..8> (one wave, matrix 2D-cyclic over the lanes); slot 7 k_ols<256,96> (four waves); slots
 * 11..14 stay zero unless SACAMD_OLS_GRID=0 selects the retired four-wave panel kernels.  kind 1 = cascade layout classes
 * 0..15 (0..6 and, since round 6, 14 / 15: search layouts; 7..9 and 10..13: canonical-order layouts of the final pass).  out receives 2 * 16 * 4 = 128
 * entries; cap = capacity of out in doubles (SACAMD_ERR_ARG if smaller).  ABI version 2 (version 1 had no cap and 80 entries). */
int sacamd_class_times(sacamd_ctx *ctx, double *out128, int cap, int reset);

/* Progress of a running sacamd_encode_frames on this context; may be called from another thread while that call
 * is in flight (the only entry point that may).  phase: 0 idle, 1 DDS search, 2 final prediction pass, 3 entropy
 * coding; generation: DDS generations evaluated so far.  Lets a caller that keeps several contexts busy stagger
 * them so that one batch's latency-bound final pass runs under another batch's search. */
int sacamd_progress(const sacamd_ctx *ctx, int *phase, int *generation);

/* debug: on!=0 enables per-section cycle counters in the predictor kernels (slows them slightly);
 * out16 (nullable, 16 entries) receives the counters of the last launch (k_ols_grid and the round-1..4 OLS kernels; the packed kernels of
 * the <= 16-tap class -- and of 17..24 taps in the search -- carry no counters and leave zeros).  One-wave OLS kernel:
 * [0] regressor+predict+pow [1] covariance update [2] LDL^T factor [3] forward solve [4] backward solve [5] tail.
 * Cascade kernel (wave 0): [8] tap sweep [9] wave reduction [10] barrier [11] predict+targets
 * [12] stage gains / experts / P x [13] RLS scalars + blend [14] P update [15] closing barrier */
int sacamd_debug_ols_profile(sacamd_ctx *ctx, int on, unsigned long long *out16);

/* ---- Predictor surface ------------------------------------------------------------------
 * Replaces: Predictor (libsac/pred.h:9-42, pred.cpp:4-46) for the ENCODER: its tparam, flattened, and the streams a
 * Predictor(r0, r1, tparam) yields when FrameCoder::PredictFrame (libsac.cpp:113-141) drives it over a frame -- pd[slot][t] = what
 * predict(slot) returns at sample t, p_lpc[slot][t], p_lms[slot][t] (nullable).  src0 / src1: the mean-removed samples of the
 * slot-0 / slot-1 channel (what the caller passes to fillbuf_ch0 / fillbuf_ch1; ch_ref has already been applied by the caller,
 * as in libsac.cpp:119-125); range4 = {r0.lo, r0.hi, r1.lo, r1.hi}.  The cascade sums in slmath::dot order whatever k, so pd and
 * p_lpc are the reference's to the last bit.  The context must have been created for nch channels and >= numsamples samples; the
 * call replaces the context's staged batch.  sac_amd/csrc/predictor.h wraps this in the reference's class (ABI version 5). */
typedef struct sacamd_pred_tparam {
  int nA, nB, nM0, nS0, nS1, k;
  int vn0[4], vn1[4];
  double vmu0[4], vmu1[4], vmudecay0[4], vmudecay1[4], vpowdecay0[4], vpowdecay1[4];
  double lambda0, lambda1, ols_nu0, ols_nu1, mu_mix0, mu_mix1, mu_mix_beta0, mu_mix_beta1;
  double beta_sum0, beta_pow0, beta_add0, beta_sum1, beta_pow1, beta_add1;
  int ch_ref;
  double bias_mu0, bias_mu1;
  int bias_scale0, bias_scale1, lm_n;
  double lm_alpha, proj_alpha0, proj_alpha1;
} sacamd_pred_tparam;
int sacamd_predictor_streams(sacamd_ctx *ctx, int nch, const int32_t *src0, const int32_t *src1, int numsamples, const int32_t *range4,
                             const sacamd_pred_tparam *tp, double *pd, double *p_lpc, double *p_lms);

int sacamd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
