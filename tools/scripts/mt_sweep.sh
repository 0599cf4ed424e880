cd $GRAFT_REPO_ROOT
EVFLY_WINO_MT=1 timeout 600 python tools/conv_sweep.py 200 2>&1 | grep -v amdgpu > gpurun_out/mt1.log
EVFLY_WINO_MT=2 timeout 600 python tools/conv_sweep.py 200 2>&1 | grep -v amdgpu > gpurun_out/mt2.log
timeout 600 python tools/conv_sweep.py 200 2>&1 | grep -v amdgpu > gpurun_out/mta.log
echo "layer MT1 MT2 auto"; paste <(awk '{print $1,$2}' gpurun_out/mt1.log) <(awk '{print $2}' gpurun_out/mt2.log) <(awk '{print $2}' gpurun_out/mta.log)
