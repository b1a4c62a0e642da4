"""Saturated throughput of the predictor stage kernels: identical work items, growing counts (GPU box)."""
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
for nA, nM0, taps in cases:
    g = P[:, 2].copy(); g[24] = nA; g[9] = nM0
    if taps: g[28], g[29], g[30], g[37] = taps
    for cnt in counts:
        ctx.kernel_times()
        ctx.evaluate(cfg, np.zeros(cnt, np.int32), np.tile(g, (cnt, 1)))
        kt = ctx.kernel_times()
        o, l, b = (kt[k]["ms"] for k in ("ols", "lms", "bias"))
        print(f"n_ols {nA+nM0:2d} taps {'dflt' if not taps else sum(taps)} items {cnt:5d}: ols {o:8.1f} ms ({o*1e3/steps:6.2f} us/step, {cnt*steps/o/1e3:7.1f} M item-steps/s)  "
              f"lms {l:8.1f} ms ({l*1e3/steps:6.2f} us/step, {cnt*steps/l/1e3:7.1f} M/s)  bias {b:6.1f} ms", flush=True)
