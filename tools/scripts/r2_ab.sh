cd $GRAFT_REPO_ROOT
L="e21 e22 e32 e42 d41 e51 d12"
EVFLY_WINO_MT=1 EVFLY_WINO_STAGGER=0 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2d_mt1_p.log
EVFLY_WINO_MT=1 EVFLY_WINO_PERSIST=0 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2d_mt1_np.log
EVFLY_WINO_MT=2 EVFLY_WINO_PERSIST=0 timeout 300 python tools/conv_sweep.py 200 $L 2>&1 | grep -v amdgpu > gpurun_out/r2d_mt2_np.log
paste <(awk '{print $1,$2}' gpurun_out/r2d_mt2_np.log) <(awk '{print $2}' gpurun_out/r2d_mt1_np.log) <(awk '{print $2}' gpurun_out/r2d_mt1_p.log)
