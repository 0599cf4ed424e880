import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "igemm"
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k] / cnt[k]:16.0f}  (n={cnt[k]})")
if "GRBM_GUI_ACTIVE" in agg and "SQ_VALU_MFMA_BUSY_CYCLES" in agg:
    g = agg["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"] / 8
    print("cycles/XCD", g, "mfma util", agg["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * g))
