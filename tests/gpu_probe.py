"""Diagnostic run on a GPU box: prints parity/timing facts (does not assert)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from oracle_api import Checker, center_frame, frame_cfg
from golden_cases import trace_cases, frame_cases, FRAMESIZE, RATE
from sac_amd.synth import synth_pcm

orc = Checker("orc")
P = orc.profile()
print("== predictor stage parity (vs oracle)", flush=True)
for name, (raw, g, opt, start, n) in trace_cases(P).items():
    nch = raw.shape[0]
    ctx = api.Context(nch, FRAMESIZE, 1)
    ctx.upload_i32([raw], FRAMESIZE)
    ctx.analyse(api.make_cfg("normal"))
    smp, stats = center_frame(raw)
    st = ctx.stats()[0]
    ok_stats = np.array_equal(st[:, [1, 2, 0]], stats)
    t = time.time(); plpc, psum, err, pred = ctx.debug_predict(0, g, start, n, opt); dt = time.time() - t
    pd, rl, rm, re = orc.predict_trace(smp, stats, g, start, n, opt)
    ps = rl + rm
    print(f"{name:28s} stats {ok_stats} plpc bit-eq {np.array_equal(plpc, rl)} (maxabs {np.abs(plpc-rl).max():.2e}) psum maxrel {np.max(np.abs(psum-ps)/(np.abs(ps)+1)):.2e} err mism {(err!=re).sum()}/{err.size} t {dt:.2f}s", flush=True)
    ctx.close()

print("== costs", flush=True)
rng = np.random.default_rng(1)
ctx = api.Context(1, 1000, 1)
for n, sc in [(1, 5), (50, 3), (2000, 40), (5000, 3000), (300, 20000)]:
    e = np.rint(rng.laplace(size=n) * sc).astype(np.int32)
    for k in range(4):
        a, b = ctx.debug_cost(k, e), orc.cost(k, e)
        print(f"cost kind {k} n {n}: gpu {a!r} orc {b!r} rel {abs(a-b)/max(abs(b),1e-300):.2e}", flush=True)
print("== bitplane coder", flush=True)
for n, sc, seed in [(1, 3, 0), (17, 5, 1), (65, 50, 6), (3000, 200, 2), (5000, 2, 3), (700, 20000, 7)]:
    r2 = np.random.default_rng(seed)
    e = np.rint(r2.laplace(size=n) * sc).astype(np.int32)
    u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)
    mb = max(int(u.max()), 1).bit_length() - 1
    t = time.time(); a = ctx.debug_bitplane(u, mb); dt = time.time() - t
    b = orc.bitplane_encode(u, mb)
    print(f"bitplane n {n} maxbpn {mb}: eq {a == b} len {len(a)} {len(b)} t {dt:.3f}s", flush=True)
ctx.close()

print("== whole frames (records vs oracle)", flush=True)
for name, (raw, cfg) in frame_cases().items():
    nch = raw.shape[0]
    ctx = api.Context(nch, FRAMESIZE, 1)
    ctx.upload_i32([raw], FRAMESIZE)
    gcfg = api.Cfg(cfg.optimize, cfg.sparse_pcm, cfg.zero_mean, cfg.reset, cfg.fraction, cfg.maxnfunc, cfg.num_threads, cfg.sigma, cfg.optk, cfg.cost)
    try:
        t = time.time(); recs, prof = ctx.encode_frames(gcfg); dt = time.time() - t
        want = orc.encode_frame(raw, cfg, FRAMESIZE)
        dec, _ = orc.decode_frame(recs[0], nch, FRAMESIZE)
        print(f"{name:24s} rec eq {recs[0] == want['record']} len {len(recs[0])}/{len(want['record'])} lossless {np.array_equal(dec, raw)} prof eq {np.array_equal(prof[0], want['profile'])} t {dt:.2f}s", flush=True)
        print("     ", {k: (round(v['ms'], 1), v['launches']) for k, v in ctx.kernel_times().items()}, flush=True)
    except Exception as ex:
        print(f"{name:24s} FAILED: {ex}", flush=True)
    ctx.close()

print("== throughput probe: 8 frames x 2 s stereo 44.1k, high dds,8", flush=True)
rate = 44100
frames = [synth_pcm(2 * rate, 2, seed=500 + i, rate=rate) for i in range(8)]
ctx = api.Context(2, 20 * rate, 8)
ctx.upload_i32(frames, 20 * rate)
cfg = api.make_cfg("high", num_threads=8)
t = time.time(); recs, prof = ctx.encode_frames(cfg); dt = time.time() - t
ns = sum(f.size for f in frames)
print(f"time {dt:.2f}s  samples {ns}  MSamples/s {ns/dt/1e6:.4f} bps {8*sum(len(r) for r in recs)/ns:.3f}", flush=True)
print({k: (round(v['ms'], 1), v['launches']) for k, v in ctx.kernel_times().items()}, flush=True)
