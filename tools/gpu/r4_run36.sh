# round 4, call 36: rocprofv3 --kernel-trace --stats of one step of the new default batch (1536 frames x 20 s)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 720 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_1536x20s_profiled.json 2> $O/bench_1536x20s_profiled.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_1536x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_1536x20s.csv; done
head -8 $O/kernel_stats_1536x20s.csv | cut -c1-60,150-230
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_1536x20s_profiled.json') if l.startswith('{')][-1])
print(d['value'], d['steps'], d['ms_per_step'], d['bps']); print(d['kernel_ms']); print(d['roofline'])
PY
