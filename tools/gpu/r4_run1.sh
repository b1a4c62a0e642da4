# round 4, call 1: bisect the 24-bit record mismatch (wave chain vs lane-0 chain, libm taps)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/gpu/bisect24.py > $O/bisect24_wave.log 2>&1; echo rc=$?; cat $O/bisect24_wave.log | cut -c1-300
SACAMD_CODER_SERIAL=1 timeout 600 python tools/gpu/bisect24.py > $O/bisect24_serial.log 2>&1; echo rc=$?; grep -v mismatches $O/bisect24_serial.log | cut -c1-300
