# round 4, call 12: cascade wave-role rotation (the serial mixer chain of co-resident workgroups on different SIMDs): saturated
# cascade throughput for rotation off / by block index / by (block/8 + block/256), then the parity subset with the default
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in 0 1 2; do
  SACAMD_LMS_ROT=$v timeout 900 python tests/gpu_throughput.py 4096 "" "1280,256,32,4;1500,2500,900,400;3300,1200,700,250;3900,1900,900,400" > $O/throughput_lms_rot$v.txt 2>&1
  echo == rot=$v; cat $O/throughput_lms_rot$v.txt | cut -c60-200
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -p no:cacheprovider -k "predictor_stages or canonical_cascade or frame_records or evaluate_costs or random_profiles or edge_frames or gpu_decoder_inverts or framecoder_wrapper_writes" > $O/gputests_rot_subset.log 2>&1; echo rc=$?; tail -3 $O/gputests_rot_subset.log | cut -c1-200
