"""Saturated throughput of the predictor stage kernels: work items of one kernel class, growing counts (GPU box).
The items differ slightly in their stage lengths (identical candidates would be answered by the search memo and
share one OLS stream) and in the OLS regulariser, so every item runs both stages."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
P = api.default_profile()
N = 40000
raw = synth_pcm(N, 1, 5, 44100)
ctx = api.Context(1, N, 1)
ctx.upload_i32([raw], N)
cfg = api.make_cfg("high")
ctx.analyse(cfg)
steps = 4000
cases = [(16, 0, None), (32, 0, None), (32, 32, None), (16, 0, (3383, 1168, 614, 273))]
counts = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,1024,2048,4096,8192".split(","))]
if len(sys.argv) > 2:      # OLS regressor lengths, e.g. 16,24,32,40,48,56,64
    cases = [(min(int(v), 32), max(int(v) - 32, 0), None) for v in sys.argv[2].split(",") if v]
if len(sys.argv) > 3:      # cascade stage lengths, e.g. "1280,256,32,4;1500,2500,900,400" (with 16 OLS taps)
    cases += [(16, 0, tuple(int(x) for x in t.split(","))) for t in sys.argv[3].split(";")]
for nA, nM0, taps in cases:
    g = P[:, 2].copy(); g[24] = nA; g[9] = nM0
    if taps: g[28], g[29], g[30], g[37] = taps
    for cnt in counts:
        ctx.kernel_times()
        G = np.tile(g, (cnt, 1)); i = np.arange(cnt)
        G[:, 37] += i % 64; G[:, 30] += (i // 64) % 64; G[:, 29] += i // 4096          # distinct cascade stages, same class
        G[:, 0] = P[0, 0] + (P[0, 1] - P[0, 0]) * (0.25 + 0.5 * i / max(cnt - 1, 1))     # distinct OLS stage (coefficient 0)
        ctx.analyse(cfg)                                                                 # forget memo and kept streams
        ctx.evaluate(cfg, np.zeros(cnt, np.int32), G.astype(np.float32))
        kt = ctx.kernel_times()
        o, l, b = (kt[k]["ms"] for k in ("ols", "lms", "bias"))
        print(f"n_ols {nA+nM0:2d} taps {'dflt' if not taps else sum(taps)} items {cnt:5d}: ols {o:8.1f} ms ({o*1e3/steps:6.2f} us/step, {cnt*steps/o/1e3:7.1f} M item-steps/s)  "
              f"lms {l:8.1f} ms ({l*1e3/steps:6.2f} us/step, {cnt*steps/l/1e3:7.1f} M/s)  bias {b:6.1f} ms", flush=True)
