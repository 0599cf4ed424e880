# usage: LIBS="prev hip" REPS=2 bash tools/scripts/lib_layer_ab.sh [steps]     (on the GPU box through gpurun)
# Per-layer A/B of library builds evfly_amd/libevfly_<name>.so on the bf16 U-Net (tools/layer_ab.py), alternating repeats.
cd $GRAFT_REPO_ROOT
N=${1:-20}
for rep in $(seq 1 ${REPS:-2}); do for l in ${LIBS:-prev hip}; do
  echo "$l | $(EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$l.so python3 tools/layer_ab.py $N ${LAYERS:-} 2>&1 | grep -v amdgpu | tail -1)"
done; done
