"""Per-sample latency of the OLS stage for chosen regressor lengths, k = 1 (final pass) and k = 4 (search), with the kernel's
section counters (GPU box).  Usage: gpu_ols_latency.py 40,48,56,64"""
import sys, os
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
for no in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40,48,56,64").split(",")]:
    for opt in (0, 1):
        g = P[:, 2].copy(); g[24] = min(no, 32); g[9] = max(no - 32, 0)
        ctx.kernel_times(); ctx.ols_profile(True)
        ctx.debug_predict(0, g, 0, n, opt)
        prof = ctx.ols_profile(True).astype(float) / n
        kt = ctx.kernel_times()
        print(f"n_ols {no:3d} k={4 if opt else 1}: ols {kt['ols']['ms']*1e3/n:8.2f} us/step   cycles/step: predict %.0f cov %.0f factor %.0f fwd %.0f bwd %.0f tail %.0f" % tuple(prof[:6]), flush=True)
