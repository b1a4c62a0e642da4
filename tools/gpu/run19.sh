# round 3, last call: the poll-bounded waits beside an idle HIP process, and the command-line decoder tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
bash tools/gpu/run18.sh > /dev/null 2>&1; cp $O/decode_cli_probe.log $O/decode_cli_probe_after_fix.log; grep -- "---\|rc=" $O/decode_cli_probe_after_fix.log | cut -c1-200
timeout 120 python -m pytest tests/test_gpu_parity.py -q -k "decode_cli or 24bit_subframe or one_launch or gpu_decoder_roundtrip" > $O/gputests_cli_after_fix.log 2>&1; tail -6 $O/gputests_cli_after_fix.log | cut -c1-200
