# round 3: why does the GPU suite stall at test 46 (decoder round trip after 45 other tests)?  staged reproducer with stack dumps
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
cat > /tmp/runpy.py <<'PY'
import faulthandler, signal, sys, os, threading, time
faulthandler.register(signal.SIGUSR1, all_threads=True)
def dog(sec):
    time.sleep(sec); faulthandler.dump_traceback(all_threads=True); sys.stderr.flush()
    time.sleep(20); os._exit(99)
threading.Thread(target=dog, args=(float(sys.argv[1]),), daemon=True).start()
import pytest
sys.exit(pytest.main(sys.argv[2:]))
PY
# stage A: a few quick tests, then the decoder tests; without / with zeroed decoder buffers
SACAMD_DEC_ZERO=0 python /tmp/runpy.py 240 tests/test_gpu_parity.py -x -q --durations=8 -k "analyse_stats or frame_records or costs_and_coder or random_profiles_residuals or edge_frames or gpu_decoder_inverts or gpu_decoder_roundtrip" > $O/dec_stageA_nozero.log 2>&1; echo "stage A nozero rc=$?"; tail -15 $O/dec_stageA_nozero.log
python /tmp/runpy.py 240 tests/test_gpu_parity.py -x -q --durations=8 -k "analyse_stats or frame_records or costs_and_coder or random_profiles_residuals or edge_frames or gpu_decoder_inverts or gpu_decoder_roundtrip" > $O/dec_stageA_zero.log 2>&1; echo "stage A zero rc=$?"; tail -8 $O/dec_stageA_zero.log
# stage B: everything but the three long full-size tests, zeroed buffers
python /tmp/runpy.py 540 tests/test_gpu_parity.py -x -q --durations=12 -k "not headline and not full_size" > $O/dec_stageB.log 2>&1; echo "stage B rc=$?"; tail -25 $O/dec_stageB.log
