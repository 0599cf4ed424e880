# usage (GPU box): bash tools/scripts/c5_var.sh "<variant names>" [family/layer ...] -- C5 bench (bf16 U-Net) under variant builds of the
# library (evfly_amd/libevfly_<name>.so; "hip" = shipped): ms/step and the named conv layers' ms
cd $GRAFT_REPO_ROOT
VARS=${1:-hip}; shift
for v in $VARS; do
  EVFLY_LIB=$GRAFT_REPO_ROOT/evfly_amd/libevfly_$v.so python bench.py --config C5 --no-cpu-baseline --no-stage-rates --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/c5_$v.json
  python - "$v" "$@" <<'PY'
import json, sys
v = sys.argv[1]; want = sys.argv[2:] or ["conv3x3/e12", "conv3x3/e21", "conv3x3/e22"]
b = json.loads(open(f"/tmp/c5_{v}.json").read())
L = {l["name"]: l["ms_per_step"] for l in b["conv_layers"]}
print(v, b["ms_per_step"], {k: L.get(k) for k in want})
PY
done
