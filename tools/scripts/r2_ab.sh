cd $GRAFT_REPO_ROOT
L="e22 e32 e42 d11 d31"
EVFLY_WINO_MT=1 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2i_a.log
EVFLY_WINO_MT=1 EVFLY_WINO_LDS_MIN=81920 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2i_b.log
EVFLY_WINO_MT=2 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2i_c.log
paste <(awk '{print $1,$2}' gpurun_out/r2i_c.log) <(awk '{print $2}' gpurun_out/r2i_a.log) <(awk '{print $2}' gpurun_out/r2i_b.log)
