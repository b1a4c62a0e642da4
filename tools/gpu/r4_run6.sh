# round 4, call 6: final pass in isolation (768 stereo frames x 40 000 samples, OLS lengths mixed like the bench's final pass):
# base vs e1 (back-substitution chunk 8, k_lms<0> 3 per CU) vs e5 (e1 + panel OLS kernels at 3 workgroups per CU by launch bounds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
MIX="16/32;24/48;32/56;16/40;24/56;32/48;16/56;32/40;24/32;16/48;32/64;24/40"
for v in base e1 e5; do
  if [ $v = base ]; then unset SACAMD_LIB_PATH; else export SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so; fi
  timeout 900 python tests/gpu_finalpass.py 768 40000 "$MIX" > $O/finalpass_$v.txt 2>&1
  echo == $v; cat $O/finalpass_$v.txt | cut -c1-200
done
