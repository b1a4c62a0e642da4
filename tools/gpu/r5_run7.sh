# round 5, call 7: final pass on a partitioned chip (CU-masked streams): long OLS classes on nA CUs per XCD, short classes + their cascades on the rest
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
B="python bench.py --frames 768 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras"
SACAMD_TRACE=1 timeout 600 $B > $O/bench_768_part6.json 2> $O/bench_768_part6.err
SACAMD_TRACE=1 SACAMD_LONG_PER_CU=8 timeout 600 $B > $O/bench_768_part8.json 2> $O/bench_768_part8.err
SACAMD_TRACE=1 SACAMD_LONG_PER_CU=4 timeout 600 $B > $O/bench_768_part0.json 2> $O/bench_768_part0.err
for f in part6 part8 part0; do
python - $O/bench_768_$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],3), 'MSamples/s', round(d['ms_per_step']/1e3,1), 's/step bps', round(d['bps'],4)); print(d['kernel_ms'])
PY
grep "sacamd trace" $O/bench_768_$f.err | grep "final pass\|steps 882000\|lms class 1[0-3]" | tail -20
done
