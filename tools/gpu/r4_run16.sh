# round 4, call 16: search cascade with mutab / powtab read from global memory (L2) instead of registers, 3 workgroups per CU
# (t1: classes 0, 5, 6 at 3/CU, 30-slot classes at 2/CU; t2: class 0 at 4/CU, 30-slot classes at 3/CU) vs the register tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in base t1 t2; do
  if [ $v = base ]; then unset SACAMD_LIB_PATH; else export SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so; fi
  timeout 900 python tests/gpu_throughput.py 4096 "" "1280,256,32,4;1500,2500,900,400;3300,1200,700,250;3900,1900,900,400" > $O/throughput_tabg_$v.txt 2>&1
  echo == $v; cat $O/throughput_tabg_$v.txt | cut -c1-40,100-200
done
