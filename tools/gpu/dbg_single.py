import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
N = 20 * 44100
k = int(sys.argv[1])
frames = [synth_pcm(N, 2, seed=1000 + i, rate=44100) for i in range(k)]
ctx = api.Context(2, N, k)
ctx.upload_i32(frames, N)
cfg = api.make_cfg("high", num_threads=0, reset=1)
t = time.time()
try:
    ctx.analyse(cfg); recs, prof = ctx.encode_frames(cfg)
    print("ok", time.time() - t)
except Exception as e:
    print("FAILED after", time.time() - t, e)
