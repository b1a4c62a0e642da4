# round 4, call 20: the complete GPU suite at the end of the round's kernel work (no -x, no -k)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu --timeout 1200 --durations=15 -p no:cacheprovider > $O/gputests_full_02.log 2>&1; echo rc=$?
tail -30 $O/gputests_full_02.log | cut -c1-200
