# usage (GPU box): bash tools/scripts/c5_env.sh "<ENV=val ...;...>" [family/layer ...] -- C5 bench (bf16 U-Net) under environment switches
# (semicolon-separated sets, "-" = none): ms/step and the named conv layers' ms
cd $GRAFT_REPO_ROOT
SETS=$1; shift
IFS=';' read -ra ARR <<< "$SETS"
for e in "${ARR[@]}"; do
  if [ "$e" = "-" ]; then python bench.py --config C5 --no-cpu-baseline --no-stage-rates --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/c5_env.json
  else env $e python bench.py --config C5 --no-cpu-baseline --no-stage-rates --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/c5_env.json; fi
  python - "$e" "$@" <<'PY'
import json, sys
v = sys.argv[1]; want = sys.argv[2:] or ["conv3x3/e12"]
b = json.loads(open("/tmp/c5_env.json").read())
L = {l["name"]: l["ms_per_step"] for l in b["conv_layers"]}
print(v, b["ms_per_step"], {k: L.get(k) for k in want})
PY
done
