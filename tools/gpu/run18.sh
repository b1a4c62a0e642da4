# round 3: does the command-line decoder work (a) alone on the GPU, (b) beside another process that holds an idle HIP context?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
python - <<'PY'
import sys; sys.path.insert(0, '.')
from sac_amd import container as C
from sac_amd.synth import synth_pcm
open('/tmp/in.wav', 'wb').write(C.wav_bytes_from_pcm(synth_pcm(3 * 8000 + 13, 2, 501, 8000), 8000, 16))
PY
{
./sac_amd/sacenc --normal --framelen=2 /tmp/in.wav /tmp/f.sac
echo "--- (a) alone"; ./sac_amd/sacenc --decode /tmp/f.sac /tmp/a.wav; echo "rc=$?"; cmp /tmp/in.wav /tmp/a.wav && echo same
echo "--- (a2) alone, one-launch form"; SACAMD_DEC_SINGLE=1 ./sac_amd/sacenc --decode /tmp/f.sac /tmp/a2.wav; echo "rc=$?"
python - <<'PY'
import subprocess, sys, os
sys.path.insert(0, '.')
import sac_amd.api as api
ctx = api.Context(2, 16000, 2)          # this process now holds HIP state (streams, a context) and stays idle
for env_extra, label in (({}, "(b) beside an idle HIP process"), ({"SACAMD_DEC_SINGLE": "1"}, "(b2) same, one-launch form"),
                         ({"GPU_MAX_HW_QUEUES": "4"}, "(c) same, child with GPU_MAX_HW_QUEUES=4")):
    r = subprocess.run(["./sac_amd/sacenc", "--decode", "/tmp/f.sac", "/tmp/b.wav"], capture_output=True, text=True, env=dict(os.environ, **env_extra))
    print("---", label, "rc", r.returncode, (r.stdout + r.stderr).strip().replace("\n", " | ")[:300], flush=True)
ctx.close()
PY
} > $O/decode_cli_probe.log 2>&1
cat $O/decode_cli_probe.log | cut -c1-300
