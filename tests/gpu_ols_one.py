import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from sac_amd.synth import synth_pcm
P = api.default_profile()
n = 2000
raw = synth_pcm(n, 1, 5, 44100)
ctx = api.Context(1, 882000, 1)
ctx.upload_i32([raw], 882000)
ctx.analyse(api.make_cfg("normal"))
g = P[:, 2].copy(); g[24] = 32; g[9] = 0
ctx.debug_predict(0, g, 0, n, 0)
print("done")
