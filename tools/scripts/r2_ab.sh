set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -15) > gpurun_out/r2a_wino_tests.log
(timeout 300 python tools/conv_sweep.py 5 2>&1) > gpurun_out/r2a_sweep_new.log
(EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_hip_r1.so timeout 300 python tools/conv_sweep.py 5 2>&1) > gpurun_out/r2a_sweep_old.log
(timeout 300 python tools/conv_sweep.py 5 2>&1) > gpurun_out/r2a_sweep_new2.log
tail -30 gpurun_out/r2a_wino_tests.log; paste gpurun_out/r2a_sweep_old.log gpurun_out/r2a_sweep_new.log | cut -c1-200
