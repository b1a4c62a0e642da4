set -x
cd $GRAFT_REPO_ROOT
SACAMD_TRACE=1 timeout 1500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r3c.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['bps']); print(d['kernel_ms']); print(d['kernel_instances_ms'])
PY
grep "trace\] ols.*882000\|trace\] lms class 1[0-3]" gpurun_out/bench_r3c.err | tail -24
