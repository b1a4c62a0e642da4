# round 5, call 6: whole job at 768 frames with the grid OLS kernels: per-OLS-class cascade groups in the final pass (default) vs two groups
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --frames 768 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras"
SACAMD_TRACE=1 timeout 600 $B > $O/bench_768_grid_groups.json 2> $O/bench_768_grid_groups.err
SACAMD_TRACE=1 SACAMD_FINAL_GROUPS=0 timeout 600 $B > $O/bench_768_grid_2groups.json 2> $O/bench_768_grid_2groups.err
for f in grid_groups grid_2groups; do
python - $O/bench_768_$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],3), 'MSamples/s', round(d['ms_per_step']/1e3,1), 's/step bps', round(d['bps'],4)); print(d['kernel_ms'])
PY
grep "sacamd trace" $O/bench_768_$f.err | grep "steps 882000\|lms class 1[0-3]" | tail -22
done
