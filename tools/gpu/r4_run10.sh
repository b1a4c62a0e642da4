# round 4, call 10: launch timeline of one step at 256 frames (SACAMD_TRACE=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SACAMD_TRACE=1 timeout 900 python bench.py --frames 256 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 > $O/bench_256_trace.json 2> $O/bench_256_trace.err
grep -c . $O/bench_256_trace.err; tail -60 $O/bench_256_trace.err | cut -c1-220
