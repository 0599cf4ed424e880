# usage (GPU box): bash tools/scripts/c16w_abl.sh [reps] [layer ...]  -- timing of the conv16w ablation builds (tools/scripts/build_variant.sh
# w<bits> conv16w.hip "-DEVFLY_C16W_ABL=<bits>"); results of those builds are garbage by construction.
cd $GRAFT_REPO_ROOT
REPS=${1:-100}; shift
LAYERS=${@:-e42 e32 e52 d11}
for l in $LAYERS; do
  line="$l:"
  for v in hip w1 w2 w6; do
    [ -f evfly_amd/libevfly_$v.so ] || continue
    t=$(EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$v.so python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1 | awk '{print $2, $4}')
    line="$line  $v: $t"
  done
  echo "$line"
done
