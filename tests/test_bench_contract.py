"""The committed bench lines (profiles/r5_C*_bench.json, produced by `python bench.py [--config ..]` on the MI355X box) carry
every field of the bench contract; guards against a refactor of bench.py dropping one."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    return json.loads(open(os.path.join(REPO, "profiles", name)).read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    b = _line("r5_C2_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "step_mfma_util", "stage_rates"):
        assert k in b, k
    assert b["unit"] == "event-frames/s" and b["higher_is_better"] is True and b["scaling"] == "weak"
    assert b["vs_baseline"] is None and b["dtype"] == "f32" and b["data"] == "synthetic" and b["config"]["workload"].startswith("C2:")
    assert abs(b["value"] - 320 * 1e3 / b["ms_per_step"]) < 1e-2 * b["value"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_useful", "frac_algorithmic", "traffic", "hbm"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # `achieved` / `frac` count the flops the matrix cores issue; useful <= issued < peak < algorithmic (Winograd)
    assert 0 < r["frac_useful"] <= r["frac"] < 1 < r["frac_algorithmic"] and r["algorithmic"]["tflops"] > r["achieved"]
    assert 0 < b["step_mfma_util"] < 1
    for k in ("v_only", "d_only", "p_only", "v_d_p"):
        assert b["stage_rates"][k]["frames_per_s"] > 0
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "threads_1", "threads_nproc"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["threads_1"]["cores"] == 1
    # round 4: the V-stage bytes are the bytes the timed kernel moves (5 B/event + the frames), pass 1 reported beside it
    st = b["stages"]
    assert abs(st["voxelize_bytes_moved"] - (5.0 * b["config"]["events_per_step_per_gpu"] + 4.0 * 320 * 260 * 346)) < 1
    assert abs(st["voxelize_frac_of_hbm_peak"] - st["voxelize_bytes_moved"] / (st["voxelize_ms"] * 1e-3) / 8e12) < 2e-3
    assert st["voxelize_with_pass1_ms"] > st["voxelize_ms"] and st["voxelize_with_pass1_bytes"] > st["voxelize_bytes_moved"]
    assert b["step_ms"]["min"] <= b["step_ms"]["median"] <= b["step_ms"]["max"]


def test_default_line_carries_the_other_baseline_configs():
    """The driver runs `python bench.py` (C2): BASELINE.json's C5 / C3 / C4 ride on the same line as compact objects."""
    oc = _line("r5_C2_bench.json")["other_configs"]
    assert set(oc) == {"C5", "C3", "C4"}
    for name, dtype, frames in (("C5", "bf16", 320), ("C3", "bf16", 2560), ("C4", "f32", 1280)):
        o = oc[name]
        assert "error" not in o and o["dtype"] == dtype and o["workload"].startswith(name + ":")
        assert abs(o["value"] - frames * 1e3 / o["ms_per_step"]) < 1e-2 * o["value"]
        r = o["roofline"]
        assert r["kernel"] == "conv3x3" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_launch_ms"] > 0
    assert oc["C3"]["pipeline"].startswith("two HIP streams") and oc["C3"]["roofline"]["one_stream"]["frac"] > oc["C3"]["roofline"]["frac"]


def test_committed_lines_of_the_other_baseline_configs():
    """BASELINE.json configs[2..4] have a driver-form line each (C3 bf16 ViT-base at 480x640, C4 shard, C5 ConvLSTM seq-16)."""
    c3, c4, c5 = _line("r5_C3_bench.json"), _line("r5_C4_bench.json"), _line("r5_C5_bench.json")
    assert c3["dtype"] == "bf16" and c3["config"]["sensor"] == [480, 640] and c3["config"]["vit_trunk"] == "base" and c3["config"]["streams_per_gpu"] == 256
    assert abs(c3["value"] - 2560 * 1e3 / c3["ms_per_step"]) < 1e-2 * c3["value"]
    assert c4["config"]["streams_per_gpu"] == 256 and c4["config"]["windows"] == 5 and c4["config"]["vit_trunk"] == "base"
    assert c5["dtype"] == "bf16" and c5["config"]["windows"] == 16 and c5["convlstm"]["steps_in_series"] == 16
    for b in (c3, c4, c5):
        assert "roofline" in b and "cpu_baseline" in b and b["value"] > 0


def test_bench_source_keeps_the_contract_keys():
    src = open(os.path.join(REPO, "bench.py")).read()
    for k in ('"metric"', '"value"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
              '"data"', '"config"', '"roofline"', '"cpu_baseline"', "--gpus", "--steps", "--warmup", "dist.barrier()",
              '"frac_useful"', '"frac_algorithmic"', '"step_mfma_util"', '"stage_rates"', '"threads_1"', '"threads_nproc"', "--config",
              "pin_memory()", '"other_configs"', '"voxelize_bytes_moved"', '"voxelize_with_pass1_ms"', '"step_ms"',
              '"ms_per_step_by_rank"', '"all_gather_us"', '"steps_executed"', "--pmc-pass", '"precision_check"', '"c4"'):
        assert k in src, k


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` outside a launcher counts the node's GPUs in the PARENT (no runtime initialisation) and refuses to spawn N
    ranks when fewer are visible -- never a silent smaller run."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "--gpus 64 but this node exposes" in r.stderr


def test_bench_configs_are_the_baseline_configs():
    """--config C2 / C3 / C4 / C5 = BASELINE.json configs[1..4]; C2 is the default and the headline."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    c = b.CONFIGS
    assert {k: v for k, v in c["C2"].items() if k != "overlap"} == dict(streams=64, windows=5, epw=60_000, sensor=(260, 346), vit="tiny", dtype="f32", model="composite")
    assert c["C3"]["streams"] == 256 and c["C3"]["windows"] == 10 and c["C3"]["sensor"] == (480, 640) and c["C3"]["dtype"] == "bf16" and c["C3"]["vit"] == "base"
    assert c["C4"]["streams"] == 256 and c["C4"]["windows"] == 5 and c["C4"]["vit"] == "base"
    assert c["C5"]["windows"] == 16 and c["C5"]["dtype"] == "bf16" and c["C5"]["model"] == "unet"
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert len(base["configs"]) == 5 and "batch=256, 480" in base["configs"][2] and "seq_len=16" in base["configs"][4]


def _bench_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_pmc_traffic_rejects_a_summary_divided_by_the_wrong_step_count():
    """Round 4's C2 PMC summary divided a 7-step run by 3 (wino_conv3x3: 65.33 launches per step) and the line claimed 2.36 x the
    algorithmic traffic. bench.py now checks the file's own bookkeeping instead of trusting it."""
    b = _bench_mod()
    t, note = b.pmc_traffic("C2", "f32", 28, files=("r4_C2_pmc_traffic.json",))
    assert t is None and "rejected" in note and "not a whole number" in note
    # a consistent file passes, and a file whose conv launch count disagrees with the run is refused as well
    import tempfile
    good = {"steps_profiled": 3, "kernels": {"wino_conv3x3": {"fetch_bytes_per_step": 13.8e9, "write_bytes_per_step": 8.8e9, "launches_per_step": 28},
                                              "k_condition": {"fetch_bytes_per_step": 4e8, "write_bytes_per_step": 1e8, "launches_per_step": 1}}}
    with tempfile.NamedTemporaryFile("w", suffix=".json", dir=os.path.join(REPO, "profiles"), delete=False) as f:
        json.dump(good, f)
    try:
        t, note = b.pmc_traffic("C2", "f32", 28.0, files=(os.path.basename(f.name),))
        assert t == round(22.6e9 / 28) and "28 conv3x3" in note
        t, note = b.pmc_traffic("C2", "f32", 24.0, files=(os.path.basename(f.name),))
        assert t is None and "rejected" in note
    finally:
        os.unlink(f.name)


def test_pmc_summary_tool_refuses_partial_steps():
    """tools/pmc_traffic_summary.py: dispatch counts that are not a whole multiple of the executed steps are an error."""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("pmc_sum", os.path.join(REPO, "tools", "pmc_traffic_summary.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rows = lambda n: [{"Kernel_Name": "void k_wino9<2, 5, false, false, 1>(ConvDesc)", "Counter_Value": "1000"}] * n
    doc = m.summarise(rows(84), rows(84), 3)
    assert doc["kernels"]["wino_conv3x3"]["launches_per_step"] == 28 and doc["steps_profiled"] == 3
    with pytest.raises(SystemExit):
        m.summarise(rows(196), rows(196), 3)


def test_round5_line_traffic_and_side_objects():
    """Round 5: `roofline.traffic` comes from a --pmc-pass summary whose bookkeeping bench.py verified (whole launches per step, the
    conv family's count equal to the run's): within [0.95, 1.10] of the algorithmic bytes for the fp32 Winograd family (the review
    found 2.36 x in round 4's line, an artefact of a wrong step count); the compact objects of the other BASELINE configs are
    self-contained (CPU baseline, single-stream latency, precision check)."""
    b = _line("r5_C2_bench.json")
    r = b["roofline"]
    assert r["traffic"] and 0.95 <= r["traffic"] / r["algorithmic"]["bytes_per_launch"] <= 1.10, r["traffic_note"]
    assert "r5_C2_pmc_traffic.json" in r["traffic_note"] and b["steps_executed"] >= b["steps"] + b["warmup"]
    t = json.load(open(os.path.join(REPO, "profiles", "r5_C2_pmc_traffic.json")))
    assert all(float(v["launches_per_step"]).is_integer() for v in t["kernels"].values()) and t["kernels"]["wino_conv3x3"]["launches_per_step"] == 28
    oc = b["other_configs"]
    for name in ("C5", "C3", "C4"):
        assert oc[name]["cpu_baseline"]["value"] > 0 and oc[name]["cpu_baseline"]["kind"] == "port"
    assert oc["C5"]["convlstm"]["single_stream_sequence_ms"] > 0
    pc = oc["C3"]["precision_check"]
    assert len(pc["streams"]) == 8 and 0 < pc["max_rel_dev_velocity_vs_f32"] < 2e-2
    # the bf16 configs' own lines carry their PMC traffic too (tap-masked stores: below the algorithmic full-map bytes)
    for name in ("r5_C5_bench.json", "r5_C3_bench.json"):
        rr = _line(name)["roofline"]
        assert rr["traffic"] and 0.6 <= rr["traffic"] / rr["algorithmic"]["bytes_per_launch"] <= 1.1, rr["traffic_note"]
