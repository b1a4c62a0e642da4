"""Cascade kernel time for chosen stage lengths (GPU box), to compare tap-slot layouts.
usage: python tests/gpu_lms_layouts.py [lib.so]
Each case evaluates 1536 candidates of which 40 are distinct (stage-3 length varied, same layout class); the
rest are answered by the search memo, so the time printed is that of 40 concurrent work-items x 4000 steps,
i.e. the per-step latency of the layout's kernel.  Pass another build of the library to compare."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
if len(sys.argv) > 1:
    api.LIB_PATH = os.path.abspath(sys.argv[1])
from sac_amd.synth import synth_pcm
P = api.default_profile()
N = 40000
raw = synth_pcm(N, 1, 5, 44100)
ctx = api.Context(1, N, 1)
ctx.upload_i32([raw], N)
cfg = api.make_cfg("high")
ctx.analyse(cfg)
steps, cnt = 4000, 1536
for taps in [(1200, 1200, 1000, 200), (500, 200, 1000, 200), (2300, 200, 200, 300), (1000, 500, 250, 100), (3383, 1168, 614, 273)]:
    G = np.tile(P[:, 2].copy(), (cnt, 1)); G[:, 24] = 8; G[:, 9] = 0
    G[:, 28], G[:, 29], G[:, 30], G[:, 37] = taps
    G[:, 37] = taps[3] - (np.arange(cnt) % 40)          # distinct items, same layout class
    ctx.analyse(cfg); ctx.kernel_times(); ctx.class_times()
    ctx.evaluate(cfg, np.zeros(cnt, np.int32), G.astype(np.float32))
    kt = ctx.kernel_times(); ct = ctx.class_times()
    lms = {k[1]: round(v[0], 1) for k, v in ct.items() if k[0] == "lms"}
    print(f"taps {taps}: lms tail {kt['lms']['ms']:7.1f} ms, ols {kt['ols']['ms']:7.1f} ms, lms instances (class: ms) {lms}  -> {max(lms.values())*1e3/steps:5.2f} us per step", flush=True)
