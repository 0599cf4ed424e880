cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -3)
for v in prev hip; do EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$v.so timeout 300 python tools/conv_sweep.py 200 2>&1 | grep -v amdgpu > gpurun_out/r2f_$v.log; done
paste <(awk '{print $1,$2,$NF}' gpurun_out/r2f_prev.log) <(awk '{print $2,$NF}' gpurun_out/r2f_hip.log)
