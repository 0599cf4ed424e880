cd $GRAFT_REPO_ROOT
for c in 640 320 160; do
  EVFLY_CHUNK_FRAMES=$c BENCH_DUMP_LAYERS=1 python3 bench.py --config C3 --no-overlap --no-cpu-baseline --no-other-configs --no-alt > gpurun_out/c3_chunk$c.json 2>/dev/null
  python3 - <<PY
import json
j=json.loads(open("gpurun_out/c3_chunk$c.json").read().strip().splitlines()[-1])
print("chunk $c", j["value"], j["ms_per_step"], j["roofline"]["achieved"])
agg={}
for e in j.get("layers",[]):
    agg[e["name"]]=agg.get(e["name"],0)+e["ms"]
print(" ".join(f"{k.split('/')[-1]}={v:.3f}" for k,v in agg.items()))
PY
done
