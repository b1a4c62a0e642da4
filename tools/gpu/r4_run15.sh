# round 4, call 15: chained tail at 768 frames x 20 s (stepwise on the same build: 236.2 s, profiles/r04/bench_768_adaptive.json)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_768_chain.json 2> $O/bench_768_chain.err
python - <<PY
import json
d=json.loads(open("$O/bench_768_chain.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
grep "steps 882000\|lms class 1[0-3]" $O/bench_768_chain.err | tail -22 | cut -c1-150
