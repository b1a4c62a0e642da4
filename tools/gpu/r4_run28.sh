# round 4, call 28: panel head start (5 ms) with a panel slot budget of 330 (64- and 56-tap classes on the panel kernel), 768 frames
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SACAMD_OLS_PANEL_SLOTS=330 SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_768_stagger330.json 2> $O/bench_768_stagger330.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_768_stagger330.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"])
PY
grep "steps 882000\|lms class 1[0-3].*items [0-9][0-9][0-9]" $O/bench_768_stagger330.err | tail -12 | cut -c1-140
