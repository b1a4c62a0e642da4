# round 4, call 22: the complete GPU suite (no -x, no -k) with the configs[3]/[4] full-size cases run concurrently
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --timeout 1200 --durations=12 -p no:cacheprovider > $O/gputests_full_03.log 2>&1; echo rc=$?
grep -v "mse:" $O/gputests_full_03.log | tail -22 | cut -c1-200
