# round 5, call 14: (a) factor-step variant f2 against the tree's library: latency + throughput; (b) decoder time of one 16-frame group;
# (c) repro of the run_single failure at 256 frames with the launch trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
( echo "== v3 (tree library)"; timeout 100 python tests/gpu_ols_latency.py 32,40,48,56,64; timeout 150 python tests/gpu_throughput.py 8192 32,40,48,56,64
  echo "== f2 (parity-specialised steps, next pivot a step ahead)"; SACAMD_LIB_PATH=$PWD/sac_amd/libsac_amd_f2.so timeout 100 python tests/gpu_ols_latency.py 32,40,48,56,64; SACAMD_LIB_PATH=$PWD/sac_amd/libsac_amd_f2.so timeout 150 python tests/gpu_throughput.py 8192 32,40,48,56,64 ) > $O/ols_grid_f2.txt 2>&1
cut -c1-112 $O/ols_grid_f2.txt
python - <<'PY' 2>&1 | grep -v "mse" | tail -3
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, sac_amd.api as api
from sac_amd.synth import synth_pcm
N = 20 * 44100
for k in (4, 16):
    frames = [synth_pcm(N, 2, seed=1000 + i, rate=44100) for i in range(k)]
    ctx = api.Context(2, N, k); ctx.upload_i32(frames, N); cfg = api.make_cfg("normal"); ctx.analyse(cfg); recs, _ = ctx.encode_frames(cfg)
    t = time.time(); dec, _ = ctx.decode_frames(recs, N); dt = time.time() - t
    print("decode", k, "frames:", round(dt, 1), "s", all(np.array_equal(d, f) for d, f in zip(dec, frames))); ctx.close()
PY
SACAMD_TRACE=1 timeout 500 python tools/gpu/dbg_single.py 256 > $O/dbg_single.out 2> $O/dbg_single.err
tail -2 $O/dbg_single.out; grep "sacamd trace" $O/dbg_single.err | tail -12; grep -c "sacamd trace\] ols" $O/dbg_single.err
