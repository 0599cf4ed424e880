"""The committed bench line (profiles/r2_bench.json, produced by `python bench.py` on the MI355X box) carries every
field of the bench contract; guards against a refactor of bench.py dropping one."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(os.path.join(REPO, "profiles", "r2_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "event-frames/s" and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["vs_baseline"] is None and b["dtype"] == "f32" and b["data"] == "synthetic" and "workload" in b["config"]
    assert abs(b["value"] - 320 * 1e3 / b["ms_per_step"]) < 1e-2 * b["value"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # `achieved` / `frac` count the flops the matrix cores issue; the direct-conv (algorithmic) rate sits beside them
    assert 0 < r["frac"] < 1 and r["algorithmic"]["tflops"] > r["achieved"]                     # Winograd: issued < algorithmic
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_bench_source_keeps_the_contract_keys():
    src = open(os.path.join(REPO, "bench.py")).read()
    for k in ('"metric"', '"value"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
              '"data"', '"config"', '"roofline"', '"cpu_baseline"', "--gpus", "--steps", "--warmup", "dist.barrier()"):
        assert k in src, k
