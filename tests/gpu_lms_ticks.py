"""Section cycle counters of the cascade kernel for given stage lengths (GPU box): python tests/gpu_lms_ticks.py n0,n1,n2,n3 [optimize]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
P = api.default_profile()
taps = tuple(int(x) for x in sys.argv[1].split(","))
opt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 1500
raw = synth_pcm(n, 1, 5, 44100)
ctx = api.Context(1, 882000, 1)
ctx.upload_i32([raw], 882000)
ctx.analyse(api.make_cfg("normal"))
g = P[:, 2].copy(); g[28], g[29], g[30], g[37] = taps
ctx.ols_profile(True); ctx.kernel_times()
ctx.debug_predict(0, g, 0, n, opt)
prof = ctx.ols_profile(True).astype(float) / n
kt = ctx.kernel_times()
print(taps, "k=%d" % (4 if opt else 1), "lms us/step %.2f" % (kt['lms']['ms'] * 1e3 / n),
      "cycles/step: sweep %.0f wsum %.0f bar %.0f head %.0f gains %.0f rls %.0f pupd %.0f bar2 %.0f" % tuple(prof[8:16]), flush=True)
