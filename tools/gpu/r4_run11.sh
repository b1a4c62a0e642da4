# round 4, call 11: 768 frames x 20 s, one step: adaptive panel / one-wave choice for the 33..64-tap OLS classes vs all-panel final pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in adaptive allpanel; do
  if [ $v = allpanel ]; then export SACAMD_OLS_PANEL_SLOTS=100000; else unset SACAMD_OLS_PANEL_SLOTS; fi
  SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_768_$v.json 2> $O/bench_768_$v.err
  echo == $v; python - <<PY
import json
d=json.loads(open("$O/bench_768_$v.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
  grep "steps 882000\|lms class 1[0-3]" $O/bench_768_$v.err | tail -24 | cut -c1-150
done
