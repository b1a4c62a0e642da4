# round 4, call 32 (final tree): complete GPU suite, the driver's bench command, rocprofv3 --kernel-trace --stats on the same batch, PMC passes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 --durations=8 -p no:cacheprovider > $O/gputests_full_04.log 2>&1; echo rc=$?
grep -v "mse:" $O/gputests_full_04.log | tail -14 | cut -c1-160
timeout 1400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_768_final.json 2> $O/bench_driver_cmd_768_final.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_driver_cmd_768_final.json') if l.startswith('{')][-1])
print(d['value'], d['steps'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('verified_frames')); print(d['kernel_ms']); print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_ms'])
print({k: d.get(k) for k in ('small_batch','single_frame_s','speedup_vs_cpu_baseline','speedup_vs_reference_threads')}); print(d['cpu_baseline']['value'], d['cpu_baseline']['threads8']['value'], d['cpu_baseline']['same_record_as_gpu'])
PY
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_768x20s_profiled_final.json 2> $O/bench_768x20s_profiled_final.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_768x20s_final.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_768x20s_final.csv; done
head -8 $O/kernel_stats_768x20s_final.csv | cut -c1-60,150-230
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_pmc_final_$c.json 2> $O/bench_pmc_final_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s_final.txt 2>&1
python tools/pmc_to_json.py $O/bench_pmc_final_FETCH_SIZE.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_hbm_final.json | tail -1
