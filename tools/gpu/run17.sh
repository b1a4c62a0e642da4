# round 3, last GPU call: the tests that never ran to completion on the GPU before (decoder round trip with the fixed oracle
# cross-check, one-launch decoder form, --decode CLI, sacenc multi-rank path on one rank)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -q --durations=12 -k "sacenc_cli or gpu_decoder or decode_cli or decoder_groups or rccl or one_launch or 24bit_subframe" > $O/gputests_last.log 2>&1; echo "rc=$?"; tail -25 $O/gputests_last.log | cut -c1-220
