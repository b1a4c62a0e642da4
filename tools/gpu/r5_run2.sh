# round 5, call 2: full GPU suite after the hygiene changes (incl. the un-gated vh_s16_e25 case)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 1200 --durations=8 > $O/gputests_01_hygiene.log 2>&1
tail -15 $O/gputests_01_hygiene.log
