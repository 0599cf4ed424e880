# usage: bash tools/scripts/abl_sweep.sh <reps> <layer ...>  -- times every ablation build present (tools/scripts/build_abl.sh; EVFLY_WINO_ABL in wino.hip)
cd $GRAFT_REPO_ROOT
R=$1; shift
for a in $(ls evfly_amd/libevfly_abl*.so | sed 's/.*libevfly_//; s/.so//'); do
  echo "== $a"; EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$a.so python3 tools/conv_sweep.py $R "$@" 2>&1 | grep -v amdgpu.ids
done
