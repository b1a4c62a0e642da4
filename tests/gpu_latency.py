"""Per-step latency of the predictor stage kernels for chosen OLS orders / tap counts (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
P = api.default_profile()
n = 4000
raw = synth_pcm(n, 1, 5, 44100)
ctx = api.Context(1, 882000, 1)
ctx.upload_i32([raw], 882000)
ctx.analyse(api.make_cfg("normal"))
print("mono work-item, %d steps; us/step per stage" % n)
for nA, nM0, taps in [(8, 0, None), (16, 0, None), (32, 0, None), (32, 16, None), (32, 32, None), (16, 0, (3383, 1168, 614, 273)), (16, 0, (8192, 4096, 2048, 1024))]:
    for opt in (0, 1):
        g = P[:, 2].copy(); g[24] = nA; g[9] = nM0
        if taps: g[28], g[29], g[30], g[37] = taps
        ctx.kernel_times()
        ctx.ols_profile(True)
        ctx.debug_predict(0, g, 0, n, opt)
        prof = ctx.ols_profile(True).astype(float) / n
        kt = ctx.kernel_times()
        print("     ols cycles/step: predict %.0f cov %.0f factor %.0f fwd %.0f bwd %.0f tail %.0f" % tuple(prof[:6]), flush=True)
        print("     lms cycles/step: sweep %.0f wsum %.0f bar %.0f head %.0f gains %.0f rls %.0f pupd %.0f bar2 %.0f" % tuple(prof[8:16]), flush=True)
        print(f"n_ols {nA+nM0:3d} taps {taps if taps else 'default'} k={4 if opt else 1}: ols {kt['ols']['ms']*1e3/n:8.2f}  lms {kt['lms']['ms']*1e3/n:7.2f}  bias {kt['bias']['ms']*1e3/n:6.2f}", flush=True)
# coder latency
rng = np.random.default_rng(0)
e = np.rint(rng.laplace(size=20000) * 300).astype(np.int32)
u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)
mb = int(u.max()).bit_length() - 1
ctx.kernel_times(); ctx.debug_bitplane(u, mb); kt = ctx.kernel_times()
print(f"coder: {kt['coder']['ms']*1e3/(u.size*(mb+1)):.3f} us per decision ({mb+1} planes)")
# stereo: slot-1 regressor a+b+c up to 96 (class 3 = generic two-wave path)
raw2 = synth_pcm(n, 2, 6, 44100)
ctx2 = api.Context(2, 882000, 1)
ctx2.upload_i32([raw2], 882000)
ctx2.analyse(api.make_cfg("normal"))
for nB, nS0, nS1 in [(16, 8, 8), (32, 16, 16), (32, 24, 24), (32, 32, 32)]:
    for opt in (0, 1):
        g = P[:, 2].copy(); g[24] = 4; g[25] = nB; g[26] = nS0; g[27] = nS1
        ctx2.kernel_times()
        ctx2.debug_predict(0, g, 0, n, opt)
        kt = ctx2.kernel_times()
        print(f"stereo slot1 n_ols {nB+nS0+nS1:3d} k={4 if opt else 1}: ols {kt['ols']['ms']*1e3/n:8.2f}  lms {kt['lms']['ms']*1e3/n:7.2f}", flush=True)
