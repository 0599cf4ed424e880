cd $GRAFT_REPO_ROOT
for v in hip pr1 pr3 hip; do EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$v.so timeout 300 python tools/conv_sweep.py 200 2>&1 | grep -v amdgpu > gpurun_out/r2h_$v.log; done
paste <(awk '{print $1,$2}' gpurun_out/r2h_hip.log) <(awk '{print $2}' gpurun_out/r2h_pr1.log) <(awk '{print $2}' gpurun_out/r2h_pr3.log)
