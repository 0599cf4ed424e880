"""Per-LAYER counters of the Winograd launches of a C2 step, from rocprofv3 --pmc passes of `bench.py --pmc-pass` (one stream, whole steps).
usage: python tools/pmc_wino_layers.py <bench line with BENCH_DUMP_LAYERS=1> <out.json> <counter_collection.csv> [more csv passes ...]
The k_wino9 dispatches of a step come in the model's launch order; the bench line's `layers` (name, launches) labels them. Counters are
summed over a layer's dispatches of all profiled steps; fractions:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)        matrix pipe busy while the layer's kernels ran
  valu / lds / vmem / wait = SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_LDS, SQ_ACTIVE_INST_VMEM, SQ_WAIT_ANY over SQ_WAVE_CYCLES (quad-cycles)
  valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_VALU_MFMA_MOPS-free estimate: VALU instructions per wave over the layer's issued MFMAs"""
import collections, csv, json, sys

line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
steps = int(line["steps_executed"])
order = []                                   # layer name per k_wino9 launch of one step
for p in line["layers"]:
    if p["name"].startswith("conv3x3/"):
        order += [p["name"][8:]] * int(p["launches"])
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sys.argv[3:]:
    rows = [r for r in csv.DictReader(open(f)) if "k_wino9" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    assert len(ids) % len(order) == 0, (f, len(ids), len(order))
    lab = {d: order[i % len(order)] for i, d in enumerate(ids)}
    nsteps = len(ids) // len(order)
    for r in rows:
        per[lab[int(r["Dispatch_Id"])]][r["Counter_Name"]] += float(r["Counter_Value"]) / nsteps
ms = {p["name"][8:]: p["ms"] for p in line["layers"] if p["name"].startswith("conv3x3/")}
tf = {p["name"][8:]: p["tflops"] for p in line["layers"] if p["name"].startswith("conv3x3/")}
out = {}
print(f"{'layer':5s} {'ms':>6s} {'alg TF/s':>8s} {'mfma busy':>9s} {'valu':>6s} {'lds':>6s} {'vmem':>6s} {'wait':>6s} {'inst-stall':>10s} {'VALU/wave':>9s} {'LDS/wave':>8s}")
for name in dict.fromkeys(order):
    c = per[name]
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    o = {"ms_bracketed": ms.get(name), "tflops_algorithmic": tf.get(name),
         "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else None}
    for key, cn in (("valu", "SQ_ACTIVE_INST_VALU"), ("lds", "SQ_ACTIVE_INST_LDS"), ("vmem", "SQ_ACTIVE_INST_VMEM"), ("wait", "SQ_WAIT_ANY"),
                    ("inst_stall", "SQ_WAIT_INST_ANY")):
        o[key] = c[cn] / wc if wc and cn in c else None
    waves = c.get("SQ_WAVES", 0.0)
    o["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / waves if waves and "SQ_INSTS_VALU" in c else None
    o["lds_insts_per_wave"] = c["SQ_INSTS_LDS"] / waves if waves and "SQ_INSTS_LDS" in c else None
    out[name] = o
    f = lambda v, w=6, d=3: (f"{v:{w}.{d}f}" if v is not None else " " * (w - 1) + "-")
    print(f"{name:5s} {f(o['ms_bracketed'])} {f(o['tflops_algorithmic'], 8, 1)} {f(o['mfma_busy'], 9)} {f(o['valu'])} {f(o['lds'])} {f(o['vmem'])} {f(o['wait'])} "
          f"{f(o['inst_stall'], 10)} {f(o['valu_insts_per_wave'], 9, 0)} {f(o['lds_insts_per_wave'], 8, 0)}")
json.dump({"note": __doc__, "steps_profiled": steps, "layers": out}, open(sys.argv[2], "w"), indent=1)
