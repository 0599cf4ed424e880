# usage: bash tools/scripts/ab_sweep.sh <lib_a.so> <lib_b.so> [layer ...]   (on the GPU box through gpurun)
# Same-box A/B of two builds of libevfly_hip.so on the Winograd layer shapes: tools/conv_sweep.py, 200 launches per layer
# (short loops read the DVFS ramp, not the kernel: DESIGN.md §3), parity column included.
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift; shift
EVFLY_LIB=$GRAFT_REPO_ROOT/$A timeout 600 python tools/conv_sweep.py 200 "$@" 2>&1 | grep -v amdgpu > gpurun_out/ab_a.log
EVFLY_LIB=$GRAFT_REPO_ROOT/$B timeout 600 python tools/conv_sweep.py 200 "$@" 2>&1 | grep -v amdgpu > gpurun_out/ab_b.log
paste <(awk '{print $1,$2,$NF}' gpurun_out/ab_a.log) <(awk '{print $2,$NF}' gpurun_out/ab_b.log)
