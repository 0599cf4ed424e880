# usage (GPU box): bash tools/scripts/w4_ab.sh [reps] [layer ...] -- same-box A/B of the Winograd kernels on the plain fp32 layer shapes at
# 320 frames (tools/conv_sweep.py, parity column included): F(2x2,3x3) (wino.hip, shipped) against the F(4x4,3x3) prototype (EVFLY_WINO4=1)
cd $GRAFT_REPO_ROOT
REPS=${1:-100}; shift
LAYERS=${@:-e21 e31 e41 e51 e52 d11 d12 d21 d22 d31 d32 d41}
EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_w4.so python tools/conv_sweep.py $REPS $LAYERS 2>&1 | grep -v amdgpu > /tmp/w4_a.log
EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_w4.so EVFLY_WINO4=1 python tools/conv_sweep.py $REPS $LAYERS 2>&1 | grep -v amdgpu > /tmp/w4_b.log
paste <(awk '{print $1, $2, $4, $NF}' /tmp/w4_a.log) <(awk '{print "| F4:", $2, $4, $NF}' /tmp/w4_b.log)
