# round 4, call 34: the gated full-size configs[3]/[4] cases (third case + decoder round trip) on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
SACAMD_SLOW_TESTS=1 timeout 1300 python -m pytest tests -q -m gpu -k "configs_3_and_4" > $O/gputests_slow_final_tree.log 2>&1
tail -5 $O/gputests_slow_final_tree.log
