# round 5, call 18: complete GPU suite on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu --timeout 1200 --durations=6 > $O/gputests_08_final_tree.log 2>&1
tail -12 $O/gputests_08_final_tree.log
