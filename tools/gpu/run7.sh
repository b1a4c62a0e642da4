set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
timeout 1500 python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r3d.json 2> gpurun_out/bench_r3d.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r3d.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('verified_frames')); print(d['kernel_ms']); print(d['cpu_baseline']); print(d['roofline'])
PY
tail -3 gpurun_out/bench_r3d.err
