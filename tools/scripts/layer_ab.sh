# usage (GPU box): bash tools/scripts/layer_ab.sh <steps> "<ENV=1 ...>" ["<ENV=1 ...>" ...] -- tools/layer_ab.py once per environment
cd $GRAFT_REPO_ROOT
N=$1; shift
for e in "$@"; do echo "$e | $(env $e python3 tools/layer_ab.py $N ${LAYERS:-} 2>&1 | grep -v amdgpu | tail -1)"; done
