set -x
cd $GRAFT_REPO_ROOT
for t in 1280,256,32,4 3267,237,480,156; do
  timeout 300 python tests/gpu_lms_ticks.py $t 0 2>&1 | tail -1
  timeout 300 python tests/gpu_lms_ticks.py $t 1 2>&1 | tail -1
done
SACAMD_TRACE=1 timeout 1500 python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err
tail -c 6000 gpurun_out/bench_r3a.json
grep "trace" gpurun_out/bench_r3a.err | awk '{print $3,$4,$5}' | sort | uniq -c | head -40
