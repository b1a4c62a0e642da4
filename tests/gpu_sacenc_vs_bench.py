"""The C++ host program (sac_amd/sacenc: WAV files in, .sac files out, no Python, GPU_MAX_HW_QUEUES NOT in its environment -- the
library sets it when it is loaded) against bench.py on the SAME 256 frames (VERDICT r4 #5c).  Prints one JSON line.
    python tests/gpu_sacenc_vs_bench.py [--frames 256]"""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sac_amd.container import wav_bytes_from_pcm  # noqa: E402
from sac_amd.synth import synth_pcm  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=256); a = ap.parse_args()
RATE, N = 44100, 20 * 44100
per_file = 16
d = tempfile.mkdtemp(prefix="sacenc_vs_bench_")
files = []
for f0 in range(0, a.frames, per_file):
    pcm = np.concatenate([synth_pcm(N, 2, seed=1000 + i, rate=RATE) for i in range(f0, min(f0 + per_file, a.frames))], axis=1)   # bench.py's frames f0..
    p = os.path.join(d, f"in{f0 // per_file:03d}.wav"); open(p, "wb").write(wav_bytes_from_pcm(pcm, RATE, 16)); files.append(p)
out = os.path.join(d, "out"); os.makedirs(out)
env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
t = time.time()
r = subprocess.run([os.path.join(ROOT, "sac_amd", "sacenc"), "--high", "--opt-cfg=dds,8", "--opt-reset", "--adapt-block=no", f"--max-frames={a.frames}"] + files + [out],
                   env=env, capture_output=True, text=True)
dt = time.time() - t
assert r.returncode == 0, r.stderr[-2000:]
sac_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
t = time.time()
b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", str(a.frames), "--steps", "1", "--warmup", "0", "--budget-s", "0", "--no-cpu-baseline",
                    "--verify-sample", "0", "--no-extras"], capture_output=True, text=True)
line = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
print(json.dumps({"frames": a.frames, "sacenc_wall_s": dt, "sacenc_includes": "process start, WAV parsing of 16 files, HIP initialisation, one batch, MD5 + .sac files",
                  "sacenc_bps": 8 * sac_bytes / (a.frames * 2 * N), "bench_step_s": line["ms_per_step"] / 1e3, "bench_bps": line["bps"],
                  "bench_wall_s_incl_synthesis": time.time() - t, "gpu_max_hw_queues_in_sacenc_env": "unset (the library's load-time default applies)",
                  "sacenc_stdout_tail": r.stdout[-300:]}))
