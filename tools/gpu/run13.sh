# round 3, final GPU call: full GPU test suite, the driver's bench command, rocprofv3 stats on the headline batch, PMC passes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/ -x -q -m gpu > $O/gputests_final.log 2>&1; tail -3 $O/gputests_final.log
timeout 1400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_768.json 2> $O/bench_driver_cmd_768.err
tail -2 $O/bench_driver_cmd_768.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_driver_cmd_768.json') if l.startswith('{')][-1])
print(d['value'], d['steps'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('verified_frames')); print(d['kernel_ms']); print(d['cpu_baseline']); print(d['roofline'])
PY
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 > $O/bench_768x20s_profiled.json 2> $O/bench_768x20s_profiled.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_768x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_768x20s.csv; done
head -8 $O/kernel_stats_768x20s.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 > $O/bench_pmc2_$c.json 2> $O/bench_pmc2_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s_final.txt 2>&1
python tools/pmc_to_json.py $O/bench_pmc2_FETCH_SIZE.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_hbm_final.json
