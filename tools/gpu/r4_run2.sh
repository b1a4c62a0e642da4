# round 4, call 2: which predictor stage diverges on 24-bit input
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/gpu/bisect24b.py > $O/bisect24_stages.log 2>&1; echo rc=$?; cat $O/bisect24_stages.log | cut -c1-400
