# round 4, call 4: A/B of two cheap register-budget experiments (tools/build_variant.sh e1: k_lms<0> at 3 workgroups per CU
# by launch bounds (42 VGPRs spilled), OLS back-substitution prefetch chunk 8 instead of 16 -> k_ols<64,24> 3 waves/SIMD, <64,40> 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in base e1; do
  if [ $v = base ]; then unset SACAMD_LIB_PATH; else export SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so; fi
  timeout 600 python tests/gpu_throughput.py 8192 16,24,32,40,48,56,64 > $O/throughput_$v.txt 2>&1
  echo == $v; cat $O/throughput_$v.txt | cut -c1-200
done
