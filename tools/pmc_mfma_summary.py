"""MFMA-pipe utilisation per kernel family from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass of bench.py.
usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <out_json>
util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs (both summed over a family's
dispatches), i.e. the fraction of all matrix-pipe cycles the family's kernels kept busy while they ran."""
import collections, csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    mm = re.search(r"\b(k_[a-z0-9_]+)", k)
    fam = ("wino_conv3x3" if "k_wino9" in k else "conv16_direct" if "k_conv16pre" in k else "conv16p_deep" if "k_conv16p" in k else
               ("conv16w_deep" if re.search(r"k_conv16w<[^>]*, *0>", k) else "conv16w_up" if re.search(r"k_conv16w<[^>]*, *1>", k) else "conv16w_gemm") if "k_conv16w" in k else
               "conv16_direct" if "k_conv16" in k else
           "igemm16_conv" if ("k_igemm16" in k and "false>" in k.replace(" ", "")) else "igemm16_gemm" if "k_igemm16" in k else
           "mfma_gemm" if "k_igemm" in k else (mm.group(1) if mm else "other"))
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[fam].add(r["Dispatch_Id"])
out = {}
for fam, c in agg.items():
    if c.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out[fam] = {"dispatches": len(disp[fam]), "xcd_cycles": cyc, "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)}
json.dump({"note": "util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8), summed over the family's dispatches of the run",
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["xcd_cycles"]))}, open(sys.argv[2], "w"), indent=1)
for fam, v in sorted(out.items(), key=lambda kv: -kv[1]["xcd_cycles"])[:8]:
    print(f"{fam:28s} dispatches {v['dispatches']:4d}  cycles/XCD {v['xcd_cycles']:.3e}  MFMA util {v['mfma_util']:.3f}")
