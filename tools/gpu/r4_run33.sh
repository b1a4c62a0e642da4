# round 4, call 33: boundary between the two cascade groups: OLS classes [0, k) form the early group; k = 4 and k = 0 (one group); default 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for k in 4 0; do
  SACAMD_FAST_OLS=$k timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_768_fast$k.json 2> $O/bench_768_fast$k.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_768_fast$k.json") if l.startswith("{")][-1])
print($k, d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"])
PY
done
