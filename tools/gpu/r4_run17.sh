# round 4, call 17: 768 frames x 20 s, final pass: panel-kernel slot budget 330 (classes 64 + 56 taps) and 20 (64 taps only); default 512
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in 330 20; do
  SACAMD_OLS_PANEL_SLOTS=$v SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_768_slots$v.json 2> $O/bench_768_slots$v.err
  echo == slots=$v; python - <<PY
import json
d=json.loads(open("$O/bench_768_slots$v.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"])
PY
  grep "steps 882000\|lms class 1[0-3].*items [0-9][0-9][0-9]" $O/bench_768_slots$v.err | tail -12 | cut -c1-150
done
