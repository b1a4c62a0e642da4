# round 5, call 13: decode throughput beside the reference decoder; the reference's default sequential search (run_single) on 256 frames
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python tests/gpu_measure_r5.py --frames-dec 64 --frames-single 256 > $O/measure_r5.json 2> $O/measure_r5.err
cat $O/measure_r5.json | cut -c1-900; tail -3 $O/measure_r5.err
