cd $GRAFT_REPO_ROOT
L="e12 e21 d42"
EVFLY_WINO_MT=1 EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_prev.so timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2j_a.log
EVFLY_WINO_MT=1 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2j_b.log
EVFLY_WINO_MT=2 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2j_c.log
echo "layer  MT2  MT1x3blocks  MT1x4blocks"; paste <(awk '{print $1,$2}' gpurun_out/r2j_c.log) <(awk '{print $2}' gpurun_out/r2j_a.log) <(awk '{print $2}' gpurun_out/r2j_b.log)
