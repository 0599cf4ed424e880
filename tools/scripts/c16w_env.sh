# usage (GPU box): bash tools/scripts/c16w_env.sh <reps> "<ENV=val ...;ENV=val ...>" [layer ...] -- tools/probe16.py under different environment
# switches (semicolon-separated sets; "-" = none), same box
cd $GRAFT_REPO_ROOT
REPS=${1:-100}; SETS=$2; shift 2
LAYERS=${@:-e52 e51 d12 d11 e42 e41}
IFS=';' read -ra ARR <<< "$SETS"
for l in $LAYERS; do
  line="$l:"
  for e in "${ARR[@]}"; do
    if [ "$e" = "-" ]; then t=$(python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1 | awk '{print $2, $4}');
    else t=$(env $e python tools/probe16.py $l $REPS 2>&1 | grep -v amdgpu | tail -1 | awk '{print $2, $4}'); fi
    line="$line  [$e] $t"
  done
  echo "$line"
done
