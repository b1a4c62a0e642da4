# round 4, call 5: cascade residency by launch bounds (compiler spills), per layout class.  Tap sets: class 0 default (1280,256,32,4);
# class 5 (6,10,4,2 slots): 1500,2500,900,400; class 6 (13,5,3,1): 3300,1200,700,250; class 1 (16,8,4,2): 3900,1900,900,400
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in base e3 e4; do
  if [ $v = base ]; then unset SACAMD_LIB_PATH; else export SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so; fi
  timeout 900 python tests/gpu_throughput.py 4096 "" "1280,256,32,4;1500,2500,900,400;3300,1200,700,250;3900,1900,900,400" > $O/throughput_lms_$v.txt 2>&1
  echo == $v; cat $O/throughput_lms_$v.txt | cut -c1-200
done
