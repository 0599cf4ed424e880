import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d["dtype"], "|", d["config"]["workload"][:90])
r = d.get("roofline", {})
print({k: r.get(k) for k in ("kernel", "achieved", "frac", "frac_useful", "frac_algorithmic", "avg_launch_ms", "hbm")}, "step_mfma_util", d.get("step_mfma_util"))
print(d.get("stages"))
print(d.get("stage_rates")); print(d.get("convlstm")); print(d.get("alt_precision")); print(d.get("cpu_baseline"))
for k in d.get("kernels", [])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]: print(k)
if len(sys.argv) > 2:
    for l in d.get("conv_layers", []): print(l)
