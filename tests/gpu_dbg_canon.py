"""Debug probe (GPU box): final-pass cascade in canonical order vs the oracle, bit by bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
from oracle_api import Checker, center_frame
P = api.default_profile()
taps = tuple(int(x) for x in sys.argv[1].split(","))
n = int(sys.argv[2])
raw = synth_pcm(n, 1, 5, 44100)
ctx = api.Context(1, 882000, 1)
ctx.upload_i32([raw], 882000)
ctx.analyse(api.make_cfg("normal"))
g = P[:, 2].copy(); g[28], g[29], g[30], g[37] = taps
ctx.kernel_times()
plpc, psum, err, pred = ctx.debug_predict(0, g, 0, n, 0)
kt = ctx.kernel_times()
orc = Checker("orc")
smp, stats = center_frame(raw)
pd, ol, om, oe = orc.predict_trace(smp, stats, g, 0, n, 0)
want = ol + om
bad = np.nonzero(psum.view(np.uint64) != want.view(np.uint64))[1]
print(taps, n, "lms us/step %.2f" % (kt['lms']['ms'] * 1e3 / n), "mismatches", bad.size, "first", bad[:5], "maxabs", np.abs(psum - want).max() if bad.size else 0, flush=True)
