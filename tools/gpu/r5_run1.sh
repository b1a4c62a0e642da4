# round 5, call 1: CU-mask probe (does hipExtStreamCreateWithCUMask partition the chip, how are the bits enumerated?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 60 ./tools/probe_cumask > $O/probe_cumask.txt 2>&1
cat $O/probe_cumask.txt
