"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into profiles/<tag>_pmc_traffic.json.
usage: python tools/pmc_traffic_summary.py <fetch_csv> <write_csv> <steps_total> <out_json>
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read; calibrated in profiles/README.md)."""
import csv, json, re, sys, collections
fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
def agg(path):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        mm = re.search(r"\b(k_[a-z0-9_]+)", k)
        fam = ("wino_conv3x3" if "k_wino" in k and "weights" not in k else
               "conv16p_deep" if "k_conv16p" in k else
               ("conv16w_deep" if re.search(r"k_conv16w<[^>]*, *0>", k) else "conv16w_up" if re.search(r"k_conv16w<[^>]*, *1>", k) else "conv16w_gemm") if "k_conv16w" in k else
               "conv16_direct" if "k_conv16" in k else
               "igemm16_conv" if ("k_igemm16" in k and "false>" in k.replace(" ", "")) else     # bf16 pipeline: non-plain = the convolutions
               "igemm16_gemm" if "k_igemm16" in k else
               "mfma_gemm" if ("k_igemm" in k or "k_conv3x3_halo" in k) else (mm.group(1) if mm else "other"))
        a[fam][0] += float(r["Counter_Value"]) * 1024.0
        a[fam][1] += 1
    return a
f, w = agg(fetch_csv), agg(write_csv)
res = {}
for fam in sorted(set(f) | set(w)):
    res[fam] = {"fetch_bytes_per_step": 2.0 * f[fam][0] / steps, "write_bytes_per_step": w[fam][0] / steps,
                "launches_per_step": max(f[fam][1], w[fam][1]) / steps}
json.dump({"steps_profiled": steps, "note": "bytes per bench step; fetch = 2 x FETCH_SIZE", "kernels": res},
          open(out, "w"), indent=1)
g = res.get("wino_conv3x3") or res.get("conv16_direct") or res.get("igemm16_conv") or res["mfma_gemm"]
print("mfma_gemm per step: fetch %.2f GB write %.2f GB launches %.0f" % (g["fetch_bytes_per_step"] / 1e9, g["write_bytes_per_step"] / 1e9, g["launches_per_step"]))
