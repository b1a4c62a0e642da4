# round 3, last GPU call: where the in-process decoder round trip stalls (stack after 40 s), and the same tests in fresh processes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
cat > /tmp/repro.py <<'PY'
import faulthandler, sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
f = open("gpurun_out/r03/dec_hang_stack.txt", "w")
import numpy as np
import sac_amd.api as api
from oracle_api import Checker
import test_gpu_parity as T
orc = Checker("orc")
t = time.time(); T.test_random_profiles_residuals(api, orc); print("random_profiles_residuals", round(time.time() - t, 1), flush=True)
t = time.time(); T.test_edge_frames_ragged_batch_vs_oracle(api, orc); print("edge_frames", round(time.time() - t, 1), flush=True)
golden = np.load("tests/golden/ref_golden.npz")
for name in ("s16_normal", "s16_high_mt4", "sparse16_normal"):
    t = time.time(); T.test_frame_records_vs_golden(api, orc, golden, name); print("frame_records", name, round(time.time() - t, 1), flush=True)
faulthandler.dump_traceback_later(40, file=f, exit=True)
t = time.time(); T._body_decoder_roundtrip_random_profiles_and_ragged_batch(api, orc); print("decoder roundtrip in process", round(time.time() - t, 1), flush=True)
faulthandler.cancel_dump_traceback_later()
PY
timeout 120 python /tmp/repro.py > $O/dec_hang_repro.log 2>&1; echo "repro rc=$?"; cat $O/dec_hang_repro.log | tail -8; cat $O/dec_hang_stack.txt | head -30
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q --durations=5 -k "random_profiles_residuals or edge_frames or gpu_decoder_roundtrip or decoder_groups" > $O/gputests_isolated.log 2>&1; echo "isolated rc=$?"; tail -12 $O/gputests_isolated.log
