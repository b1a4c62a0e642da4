# round 5, call 19: the driver's bench command on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_1536_final.json 2> $O/bench_driver_cmd_1536_final.err
python - $O/bench_driver_cmd_1536_final.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
for k in ('value','steps','ms_per_step','step_seconds','bps','speedup_vs_cpu_baseline','speedup_vs_reference_threads','reference_records_equal','reference_records','verified_lossless','small_batch','single_frame_s','half_batch','roofline','kernel_ms'):
    print(k, d.get(k))
print(d['cpu_baseline'])
PY
