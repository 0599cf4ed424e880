# usage (GPU box): bash tools/scripts/c5_layers_ab.sh "<ENV=1 ...>" ["<ENV=1 ...>" ...]  -- C5 per-layer milliseconds for each environment
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  env $e BENCH_DUMP_LAYERS=1 python3 bench.py --config ${CFG:-C5} --no-cpu-baseline --no-other-configs --no-alt > gpurun_out/_ab.json 2>gpurun_out/_ab.err || tail -3 gpurun_out/_ab.err
  python3 - "$e" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1])
agg = {}
for e in j.get("layers", []):
    k = e["name"].split("/")[-1]
    agg[k] = agg.get(k, 0) + e["ms"]
print(sys.argv[1], "|", j["value"], j["ms_per_step"], "conv3x3 TF", j["roofline"]["achieved"])
print("   " + " ".join(f"{k}={v:.3f}" for k, v in agg.items() if v > 0.03))
PY
done
