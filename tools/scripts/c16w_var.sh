# usage (GPU box): bash tools/scripts/c16w_var.sh <reps> "<variant names>" [layer ...] -- same-box timing of variant builds of the
# library (evfly_amd/libevfly_<name>.so from tools/scripts/build_variant.sh; "hip" = the shipped build) on tools/probe16.py
cd $GRAFT_REPO_ROOT
REPS=${1:-100}; VARS=${2:-hip}; shift 2
LAYERS=${@:-e42 e32 e52 d11 e41 d21}
for l in $LAYERS; do
  line="$l:"
  for v in $VARS; do
    t=$(EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$v.so python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1 | awk '{print $2, $4}')
    line="$line  $v: $t"
  done
  echo "$line"
done
