"""Final pass (k = 1) in isolation on the GPU box: F stereo frames of T samples whose profiles give chosen OLS regressor
lengths, through sacamd_predict_final; prints the span of the OLS / cascade stage and the per-class kernel times.
  python tests/gpu_finalpass.py F T "n0/n1;n0/n1;..."     n0 / n1: regressor length of channel 0 / 1 (mix of frames, round robin)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
mix = [tuple(int(x) for x in m.split("/")) for m in (sys.argv[3] if len(sys.argv) > 3 else "16/32").split(";")]
taps = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else None      # cascade stage lengths of both channels
P = api.default_profile()
raws = [synth_pcm(T, 2, 100 + (i % 16), 44100) for i in range(F)]
ctx = api.Context(2, T, F)
ctx.upload_i32(raws, T)
cfg = api.make_cfg("normal")
ctx.analyse(cfg)
profs = np.tile(P[:, 2], (F, 1)).astype(np.float32)
for f in range(F):
    n0, n1 = mix[f % len(mix)]
    profs[f, 24] = min(n0, 32); profs[f, 9] = max(n0 - 32, 0)             # ch0: nA + nM0
    nb = min(n1, 32); rest = n1 - nb
    profs[f, 25] = nb; profs[f, 26] = min(rest, 32); profs[f, 27] = max(rest - 32, 0)   # ch1: nB + nS0 + nS1
    if taps:
        profs[f, [28, 29, 30, 37]] = taps; profs[f, [31, 32, 33, 38]] = taps
ctx.kernel_times(); ctx.class_times()
t = time.time()
ctx.predict_final(cfg, profs)
dt = time.time() - t
kt = ctx.kernel_times(); ct = ctx.class_times()
print(f"frames {F} x {T} samples, mix {mix}: wall {dt:.2f} s  ols span {kt['ols']['ms']/1e3:.2f} s  cascade behind it {kt['lms']['ms']/1e3:.2f} s  bias {kt['bias']['ms']/1e3:.2f} s")
for (kind, cls), (ms, launches, isteps, flops) in sorted(ct.items()):
    if ms > 0:
        print(f"   {kind} slot {cls:2d}: {ms/1e3:7.2f} s  launches {int(launches)}  items {isteps/T:6.0f}  us/sample/launch {ms*1e3/T/max(launches,1):7.2f}  M item-steps/s {isteps/ms/1e3:7.1f}")
ctx.close()
