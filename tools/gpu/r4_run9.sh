# round 4, call 9: whole job at 256 frames x 20 s, one step each: packed OLS kernels off / on (k_lms<0> at 3 workgroups per CU in both)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in 0 1; do
  SACAMD_OLS_PACK=$v timeout 900 python bench.py --frames 256 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_256_pack$v.json 2> $O/bench_256_pack$v.err
  echo == pack=$v; python - <<PY
import json
d=json.loads(open("$O/bench_256_pack$v.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
print({k:v for k,v in d["kernel_instances_ms"].items() if "ols" in k})
PY
done
