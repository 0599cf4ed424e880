"""Copy one tools/scripts/profile_round6.sh output set (gpurun_out/<tag>/) into profiles/r6_*.   usage: python tools/install_profiles_r5.py <tag>"""
import json, os, shutil, sys
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
for c in ("C2", "C5", "C3"):
    for f in ("bench.json", "bench_under_rocprof.json", "rocprofv3_kernel_stats.csv", "pmc_traffic.json", "pmc_mfma.json"):
        shutil.copy(os.path.join(src, f"{c}_{f}"), os.path.join("profiles", f"r6_{c}_{f}"))
for c in ("C3", "C4"):
    shutil.copy(os.path.join(src, f"{c}_bench.json"), os.path.join("profiles", f"r6_{c}_bench.json"))
for c in ("C2", "C3", "C4", "C5"):
    b = json.loads(open(f"profiles/r6_{c}_bench.json").read().strip().splitlines()[-1])
    r = b["roofline"]
    print(c, b["value"], "frames/s", b["ms_per_step"], "ms", b["dtype"], "| frac", r["frac"], "useful", r.get("frac_useful"), "alg", r.get("frac_algorithmic"),
          "hbm", r.get("hbm", {}).get("frac"), "| step_mfma_util", b.get("step_mfma_util"), "| cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"])
for f in ("C2_wino_layers_pmc.json", "C2_wino_layers_pmc.txt"):
    shutil.copy(os.path.join(src, f), os.path.join("profiles", "r6_" + f))
