# round 3: GPU tests, headline bench line, rocprofv3 stats on the headline batch, PMC passes (64 x 20 s)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout 1500 python bench.py --steps 1 --warmup 1 > $O/bench_384x20s_1step.json 2> $O/bench_384x20s_1step.err
tail -2 $O/bench_384x20s_1step.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_384x20s_1step.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('verified_frames')); print(d['kernel_ms']); print(d['cpu_baseline']); print(d['roofline'])
PY
# same batch (PCM cache on disk now), one step, under rocprofv3
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 > $O/bench_384x20s_profiled.json 2> $O/bench_384x20s_profiled.err
find /tmp/prof_full -name "*stats*.csv" | head; for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_384x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_384x20s.csv; done
head -12 $O/kernel_stats_384x20s.csv
# PMC: kernels are serialised under counter collection, so a smaller batch of full-length frames
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 1200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s.txt 2>&1; head -30 $O/pmc_hbm_64x20s.txt
