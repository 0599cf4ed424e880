"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --pmc-pass` into profiles/<tag>_pmc_traffic.json.
usage: python tools/pmc_traffic_summary.py <fetch_csv> <write_csv> <bench_line_of_the_pass.json | steps> <out_json>
The step count is `steps_executed` of the line the profiled bench.py printed (every whole step it ran), never a number typed into a
script; every kernel family's dispatch count must be a whole multiple of it (a --pmc-pass run holds nothing but whole steps).
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read; calibrated in profiles/README.md)."""
import csv, json, re, sys, collections


SETUP = re.compile(r"pack|fill|copy|k_f32_to_bf16|k_check_sorted|k_window_ranges|interleave|fragment|repack|other|k_meta|transpose|k_bf16")


def family(k):
    mm = re.search(r"\b(k_[a-z0-9_]+)", k)
    return ("wino_conv3x3" if "k_wino" in k and "weights" not in k else
            "conv16_direct" if "k_conv16pre" in k else "conv16p_deep" if "k_conv16p" in k else
            ("conv16w_deep" if re.search(r"k_conv16w<[^>]*, *0>", k) else "conv16w_up" if re.search(r"k_conv16w<[^>]*, *1>", k) else "conv16w_gemm") if "k_conv16w" in k else
            "conv16_direct" if "k_conv16" in k else
            "igemm16_conv" if ("k_igemm16" in k and "false>" in k.replace(" ", "")) else     # bf16 pipeline: non-plain = the convolutions
            "igemm16_gemm" if "k_igemm16" in k else
            "mfma_gemm" if ("k_igemm" in k or "k_conv3x3_halo" in k) else (mm.group(1) if mm else "other"))


def agg(rows):
    a = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        fam = family(r["Kernel_Name"])
        a[fam][0] += float(r["Counter_Value"]) * 1024.0
        a[fam][1] += 1
    return a


def summarise(fetch_rows, write_rows, steps):
    f, w = agg(fetch_rows), agg(write_rows)
    res = {}
    for fam in sorted(set(f) | set(w)):
        n = max(f[fam][1], w[fam][1])
        if f[fam][1] and w[fam][1] and f[fam][1] != w[fam][1]:
            raise SystemExit(f"{fam}: {f[fam][1]} dispatches in the FETCH pass, {w[fam][1]} in the WRITE pass")
        setup = 0
        if n % steps:
            # one-time kernels of model creation / event upload (weight packing, buffer fills, the voxelizer's pass 1) run outside the
            # steps: their remainder is listed as `setup_launches`; any OTHER family that does not divide is a bookkeeping error
            if not SETUP.search(fam):
                raise SystemExit(f"{fam}: {n} dispatches are not a whole multiple of the {steps} steps the run executed -- not a --pmc-pass run, or a wrong step count")
            setup = n % steps
        res[fam] = {"fetch_bytes_per_step": 2.0 * f[fam][0] / steps, "write_bytes_per_step": w[fam][0] / steps, "launches_per_step": n // steps}
        if setup:
            res[fam]["setup_launches"] = setup
    return {"steps_profiled": steps, "note": "bytes per bench step; fetch = 2 x FETCH_SIZE; steps_profiled = steps_executed of the profiled bench.py --pmc-pass line", "kernels": res}


if __name__ == "__main__":
    fetch_csv, write_csv, steps_arg, out = sys.argv[1:5]
    if steps_arg.isdigit():
        steps = int(steps_arg)
    else:
        steps = int(json.loads(open(steps_arg).read().strip().splitlines()[-1])["steps_executed"])
    doc = summarise(list(csv.DictReader(open(fetch_csv))), list(csv.DictReader(open(write_csv))), steps)
    json.dump(doc, open(out, "w"), indent=1)
    res = doc["kernels"]
    g = res.get("wino_conv3x3") or res.get("conv16_direct") or res.get("igemm16_conv") or res["mfma_gemm"]
    print("steps %d; conv family per step: fetch %.2f GB write %.2f GB launches %d" % (steps, g["fetch_bytes_per_step"] / 1e9, g["write_bytes_per_step"] / 1e9, g["launches_per_step"]))
