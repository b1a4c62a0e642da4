# round 4, call 3: the complete GPU suite, no -x, no -k (one full green log before any perf work)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --timeout 1200 --durations=0 -p no:cacheprovider > $O/gputests_full.log 2>&1; echo rc=$?
tail -90 $O/gputests_full.log | cut -c1-220
