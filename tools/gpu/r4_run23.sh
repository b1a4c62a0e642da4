# round 4, call 23: the driver's bench command, the same batch under rocprofv3 --kernel-trace --stats, two PMC passes (HBM traffic),
# BASELINE configs[3] / [4] at larger batches
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
timeout 1400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_768.json 2> $O/bench_driver_cmd_768.err
tail -2 $O/bench_driver_cmd_768.err | cut -c1-200
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_driver_cmd_768.json') if l.startswith('{')][-1])
print(d['value'], d['steps'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('verified_frames')); print(d['kernel_ms']); print(d['cpu_baseline']); print(d['roofline'])
print({k: d.get(k) for k in ('small_batch','single_frame_s','speedup_vs_cpu_baseline','speedup_vs_reference_threads','h2d')})
PY
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_768x20s_profiled.json 2> $O/bench_768x20s_profiled.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_768x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_768x20s.csv; done
head -12 $O/kernel_stats_768x20s.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s.txt 2>&1; head -30 $O/pmc_hbm_64x20s.txt | cut -c1-200
python tools/pmc_to_json.py $O/bench_pmc_FETCH_SIZE.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_hbm.json; head -c 600 $O/pmc_hbm.json
timeout 1500 python tests/gpu_baseline_configs.py --frames-best 256 --frames-vh 64 > $O/configs34.json 2> $O/configs34.err; cat $O/configs34.json | cut -c1-700
