#!/usr/bin/env python3
"""Headline benchmark: event-frames/s of the event -> frame -> depth -> velocity hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]     (N > 1: launched by torch.distributed.run)

One step = one pass of the whole path over one batch of synthetic event streams that are already resident in HBM:
voxelize (B streams x T windows) -> q97 conditioning (+ centre crop) -> OrigUNet + ConvLSTM (depth) -> LSTMNetVIT
(velocity), batch-as-time inside every stream; the timed region ends with the velocity rows of all
ranks in pinned HOST memory. With N GPUs every rank runs its own B streams (weak scaling, no data-path
collective) and the ranks all_gather their velocity rows over RCCL.

Workloads (BASELINE.json `configs`); the default, and the one `metric` is quoted on, is C2:
  C2  64 streams x 5 windows x 60 000 events, 260x346, fp32, reference-size ("tiny") Mix-Transformer
  C3  256 streams x 10 windows x 200 000 events at 480x640 -> centre crop 260x346, "ViT-base" trunk, bf16 pipeline
  C4  one GPU's shard of the 8-GPU config: 256 streams x 5 windows, 260x346, "ViT-base", fp32 (run with --gpus N)
  C5  OrigUNet + ConvLSTM only, 20 streams x 16 windows (seq_len 16, batch-as-time), bf16 pipeline; reports the serial
      ConvLSTM critical path and the single-stream sequence latency beside the throughput

Prints ONE JSON line (contract in the task brief) with these extra objects:
  roofline       the dominant kernel family, timed with HIP events on the launch stream inside the timed region
                 (evfly_model_set_profiling + _filter). `achieved` / `frac` = matrix-core flops ISSUED per second (for the
                 fp32 Winograd kernel 16/36 of the direct-conv count plus tile padding); `frac_useful` = without the padding;
                 `frac_algorithmic` = SURVEY.md §8d's 2*M*N*K over the same time (exceeds 1 for Winograd: it computes the
                 convolution with 2.25x fewer multiplies); `hbm` = algorithmic bytes / time against the 8 TB/s peak
  step_mfma_util matrix-core flops issued by EVERY kernel of a step / ms_per_step / peak
  stage_rates    labelled V-only (voxelize + condition), D-only (U-Net + ConvLSTM), P-only (ViT + LSTM) frames/s
  cpu_baseline   the CPU oracle (oracle/, a port) on a bounded sample of the same workload, at 1 thread and at nproc
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

H, W = 260, 346
# dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md. bf16x3 issues 3 bf16 MFMAs per algorithmic product,
# so its algorithmic-flop ceiling is a third of the bf16 peak.
PEAK = {"f32": 157.3, "bf16": 2500.0, "bf16x3": 2500.0 / 3}
HBM_PEAK_GBS = 8000.0
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script, summarised per kernel family (first existing file wins)
PMC_TRAFFIC = {"C2": ("r6_C2_pmc_traffic.json", "r5_C2_pmc_traffic.json", "r4_C2_pmc_traffic.json", "r3_C2_pmc_traffic.json"),
               "C5": ("r6_C5_pmc_traffic.json", "r5_C5_pmc_traffic.json", "r4_C5_pmc_traffic.json"), "C3": ("r6_C3_pmc_traffic.json", "r5_C3_pmc_traffic.json")}

CONFIGS = {
    # The composite configs run the velocity model on a second HIP stream (evfly_amd/pipeline.py; --no-overlap: one stream). With the
    # ViT-base trunk in bf16 (C3) it is a third of the step and made of launches that leave most of the chip idle (80.2 -> 76.1 ms per
    # step when it was introduced). With the fp32 Winograd kernels holding every CU's LDS the gain is small but repeatable -- C2 same box,
    # three alternating runs: 21.11 / 21.12 / 21.13 -> 20.91 / 20.88 / 20.86 ms (+ 1.2 %) -- and kernels of two streams sharing the chip stretch
    # the HIP-event durations the roofline is computed from (`frac` 0.661 -> 0.644 inside the timed region): the line carries
    # `roofline.one_stream`, the same family over three one-stream steps outside the timed region, beside it.
    "C2": dict(streams=64, windows=5, epw=60_000, sensor=(260, 346), vit="tiny", dtype="f32", model="composite", overlap=True),
    "C3": dict(streams=256, windows=10, epw=200_000, sensor=(480, 640), vit="base", dtype="bf16", model="composite", overlap=True),
    "C4": dict(streams=256, windows=5, epw=60_000, sensor=(260, 346), vit="base", dtype="f32", model="composite", overlap=True),
    "C5": dict(streams=20, windows=16, epw=60_000, sensor=(260, 346), vit="tiny", dtype="bf16", model="unet"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2", help="BASELINE.json workload (default C2, the headline)")
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (overrides the config)")
    ap.add_argument("--windows", type=int, default=None, help="time windows per stream (overrides the config)")
    ap.add_argument("--events-per-window", type=int, default=None)
    ap.add_argument("--vit", choices=["tiny", "base"], default=None)
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x3"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational bf16x3 pass (C2 only)")
    ap.add_argument("--no-other-configs", action="store_true", help="default C2 run: skip the compact C5 / C3 / C4 objects")
    ap.add_argument("--no-overlap", action="store_true", help="run the velocity model back to back behind the depth model on one "
                    "stream (the composite call) even where the config's default is the two-stream pipeline (C3)")
    ap.add_argument("--overlap", action="store_true", help="velocity model of step i on a second HIP stream under the depth model "
                    "of step i + 1 (evfly_amd/pipeline.py) for every composite config")
    ap.add_argument("--no-stage-rates", action="store_true", help="skip the V / D / P-only timings")
    ap.add_argument("--pmc-pass", action="store_true", help="shape of a rocprofv3 --pmc counter pass: ONE stream, nothing but whole steps "
                    "(warm-up + one bracketed + K timed; no stage timings, no one-stream probe, no side configs), so that every kernel's "
                    "dispatch count is a whole multiple of `steps_executed`, which the line carries (tools/pmc_traffic_summary.py reads it)")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--side-cpu-seconds", type=float, default=4.0, help="CPU-oracle budget of each `other_configs` / `c4` object (one thread count)")
    ap.add_argument("--no-c4", action="store_true", help="N > 1 ranks: skip the compact C4 object (BASELINE's 8-GPU config) behind the C2 line")
    ap.add_argument("--c4-streams", type=int, default=None, help="streams per rank of that C4 object (default: the config's 256; tests shrink it)")
    a = ap.parse_args()
    if a.pmc_pass:
        a.no_overlap = a.no_stage_rates = a.no_cpu_baseline = a.no_alt = a.no_other_configs = True
    cfg = dict(CONFIGS[a.config])
    for k, v in (("streams", a.streams), ("windows", a.windows), ("epw", a.events_per_window), ("vit", a.vit), ("dtype", a.dtype)):
        if v is not None:
            cfg[k] = v
    a.cfg = cfg
    return a


def build_model(cfg):
    from evfly_amd import synthetic as syn
    import evfly_amd.learner_models as lm
    import evfly_amd.vitfly_models as vm
    kw = dict(num_in_channels=2, num_out_channels=1, num_recurrent=[1, 0], input_shape=[1, 1, H, W], velpred=0, form_BEV=2,
              evs_min_cutoff=0.15, skip_type="interp", logger=lambda *a: None)       # eval_config_real.txt
    if cfg["model"] == "unet":
        m = lm.OrigUNet(**kw)
    else:
        m = lm.OrigUNet_w_VITFLY_ViTLSTM(vit_trunk=vm.BASE if cfg["vit"] == "base" else None, **kw)
    sd = syn.fill_state_dict(m.state_dict())
    m.load_state_dict(sd)
    m.set_compute_dtype(cfg["dtype"])
    return m.to("cuda").float().eval(), sd


def cpu_baseline(sd, cfg, budget_s, only_mid=False):
    """The oracle (CPU port of the reference path) on a bounded sample of the same workload: streams of T windows through
    the C voxelizer port -> (crop +) conditioning -> model forward, at ONE thread and at every core torch sees (§8d)."""
    from evfly_amd import synthetic as syn
    import evfly_amd.vitfly_models as vm
    from oracle import accum as oaccum, conditioning as ocond, models as om, voxel as ovox
    T, epw, (hs, ws) = cfg["windows"], cfg["epw"], cfg["sensor"]
    if cfg["vit"] == "base":
        om.use_trunk(heads=vm.BASE["heads"], layers=vm.BASE["layers"], reductions=vm.BASE["reductions"])

    def one_stream(s):
        ev, edges = syn.make_stream(10_000 + s, T, hs, ws, epw)
        tt = time.perf_counter()
        c = oaccum.window_counts_c(ev["x"], ev["y"], ev["t"], ev["p"], edges, hs, ws, 0)
        fr = torch.from_numpy(ovox.signed_frame(c[:, 0], c[:, 1]).astype(np.float32)[:, None])
        if (hs, ws) != (H, W):
            fr = ocond.center_crop(fr)
        x, _ = ocond.q97_normalize(fr)
        with torch.no_grad():
            if cfg["model"] == "unet":
                om.origunet_forward(sd, x, None)
            else:
                om.composite_forward(sd, [x, torch.full((T, 1), 4.0), [None, None], None])
        return time.perf_counter() - tt

    nproc = torch.get_num_threads()
    res = {}
    try:
        if only_mid:
            torch.set_num_threads(max(1, min(16, nproc)))
        one_stream(0)                                          # warm-up (oneDNN primitive caches)
        # (oneDNN on a many-core host is slower with EVERY core on a T-frame batch than with a few: a middle count is timed too so
        # that `value` is the host's best, not a strawman)
        mid = max(1, min(16, nproc))
        plan = [("threads_1", 1, 0.35), ("threads_nproc", nproc, 0.35)] + ([(f"threads_{mid}", mid, 0.3)] if mid not in (1, nproc) else [])
        if only_mid:
            plan = [(f"threads_{mid}", mid, 1.0)]
        for label, th, share in plan:
            torch.set_num_threads(th)
            frames_done, dt, s = 0, 0.0, 0
            while dt < budget_s * share:
                dt += one_stream(1 + s)                        # synthetic-event generation is not charged
                frames_done += T
                s += 1
            res[label] = {"value": round(frames_done / dt, 3), "cores": th, "streams": s, "seconds": round(dt, 1)}
    finally:
        torch.set_num_threads(nproc)
        om.use_trunk()
    best = max(res.values(), key=lambda r: r["value"])
    what = "U-Net + ConvLSTM" if cfg["model"] == "unet" else f"composite ({cfg['vit']} ViT)"
    out = {"value": best["value"], "unit": "event-frames/s", "cores": best["cores"], "kind": "port"}
    out.update(res)
    if only_mid:
        r = best
        out["sample"] = (f"{r['streams']} stream(s) of {T} windows x {epw} events at {hs}x{ws} in {r['seconds']} s at {r['cores']} threads (the count that wins the "
                         f"headline's three-way comparison on these hosts): C voxelizer port + torch-CPU fp32 oracle forward of the {what}")
        return out
    out["sample"] = (f"`value` = the FASTEST of {len(res)} thread counts ({', '.join(str(r['cores']) for r in res.values())}; oneDNN on a many-core "
                     f"host is slower on a {T}-frame batch with every core than with a few); streams of {T} windows x {epw} events at {hs}x{ws}: C voxelizer port + torch-CPU fp32 oracle forward of the {what}, "
                     f"batch-as-time; " + ", ".join(f"{r['streams']} stream(s) at {r['cores']} thread(s) ({r['seconds']} s)" for r in res.values()) +
                     "")
    return out


def pmc_traffic(cname, dtype, dom_launches_per_step, files=None):
    """HBM bytes per launch of the dominant family (the 3x3 convs) from a committed PMC summary (tools/pmc_traffic_summary.py over
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `bench.py --pmc-pass`). The file's own bookkeeping is checked, not
    trusted: every family's launches per step must be a whole number (a pass divided by the wrong step count -- round 4's C2 file:
    196 / 3 -- is rejected) and the conv family's must equal what this run launched per step. -> (bytes per launch | None, note)."""
    files = PMC_TRAFFIC.get(cname, ()) if files is None else files
    pmc = next((q for q in (os.path.join(REPO, "profiles", f) for f in files) if os.path.exists(q)), None)
    if not pmc:
        return None, "no PMC summary for this shape/dtype"
    ks = json.load(open(pmc))["kernels"]
    base = os.path.basename(pmc)
    bad = [k for k, v in ks.items() if abs(v["launches_per_step"] - round(v["launches_per_step"])) > 1e-6]
    if bad:
        return None, f"profiles/{base} rejected: launches_per_step of {bad[0]} = {ks[bad[0]]['launches_per_step']:.2f} is not a whole number (divided by a wrong step count)"
    # fp32: the Winograd launches; bf16 pipeline: the 3x3 convs run on the direct kernel (C_in <= 64) and the patch-staged /
    # per-tap wide-tile kernels (deep layers), igemm16 for what is left
    fams = ["wino_conv3x3"] if dtype == "f32" else ["conv16_direct", "conv16w_deep", "conv16p_deep"]
    gs = [ks[f] for f in fams if f in ks]
    if not gs:
        return None, f"profiles/{base}: no conv3x3 kernels in it"
    lps = round(sum(x["launches_per_step"] for x in gs))
    if lps != round(dom_launches_per_step) and dtype != "f32" and "igemm16_conv" in ks:
        # (a 3x3 layer that none of the three direct kernels took runs on igemm16's convolution form -- whose family also holds the
        # velocity model's patch / reduction / head convolutions: counted only if that makes the launch counts agree)
        gs.append(ks["igemm16_conv"])
        lps = round(sum(x["launches_per_step"] for x in gs))
    if lps != round(dom_launches_per_step):
        return None, f"profiles/{base} rejected: {lps} conv3x3 launches per step in the PMC pass, {dom_launches_per_step:.2f} in this run"
    by = sum(x["fetch_bytes_per_step"] + x["write_bytes_per_step"] for x in gs)
    return round(by / lps), ("HBM bytes per launch, mean over the %d conv3x3 kernel launches of a step (17 layers): (2 x FETCH_SIZE + WRITE_SIZE) from "
                             "profiles/%s (%s steps profiled); algorithmic = algorithmic.bytes_per_launch" % (lps, base, json.load(open(pmc)).get("steps_profiled")))


def visible_gpus():
    """GPUs this process could use, WITHOUT initialising the runtime (torch.cuda.device_count() only enumerates on this image;
    /sys/class/kfd is the cross-check when torch reports none)."""
    n = torch.cuda.device_count()
    if n:
        return n
    import glob
    cnt = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            kv = dict(l.split() for l in open(prop) if len(l.split()) == 2)
            cnt += int(kv.get("simd_count", "0")) > 0
        except OSError:
            pass
    return cnt


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start one rank per GPU as CHILD processes (nothing in this process has touched the
        # GPU yet) and leave with their exit code -- never silently measure one GPU when N were asked for
        have = visible_gpus()
        if have < a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but this node exposes {have} GPU(s); refusing to start {a.gpus} ranks")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("EVFLY_BENCH_FORCE_DIST"):     # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))      # RCCL over xGMI

    out = run_config(a, a.config, a.cfg, rank, world, dist, detail=True)
    # BASELINE.json's other GPU configs in the SAME (driver-run) invocation: compact objects under `other_configs`; the ONE JSON
    # line and `value` stay the headline config's. Only for the plain default run (no shape overrides), on one GPU.
    if a.config == "C2" and a.cfg == CONFIGS["C2"] and world == 1 and not a.no_other_configs:
        others = {}
        for name in ("C5", "C3", "C4"):
            t0 = time.perf_counter()
            try:
                o = run_config(a, name, dict(CONFIGS[name]), rank, world, dist, detail=False)
                others[name] = compact(o, time.perf_counter() - t0)
            except Exception as e:                                       # never lose the headline line to a side config
                others[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            out["other_configs"] = others
    # N > 1 ranks (the driver's SCALE runs launch the default C2 line): BASELINE.json's 8-GPU config IS C4 -- 2048 streams as 256 per
    # rank, ViT-base, fp32, the all_gather of (256 * 5, 3) velocity rows per rank -- so the same invocation measures it too and
    # appends it as a compact object; `value` stays C2's (weak-scaled), `c4.value` is the whole job's C4 rate.
    if a.config == "C2" and dist is not None and not a.no_c4:
        c4 = dict(CONFIGS["C4"])
        if a.c4_streams:
            c4["streams"] = a.c4_streams
        t0 = time.perf_counter()
        try:
            o = run_config(a, "C4", c4, rank, world, dist, detail=False)
            o = compact(o, time.perf_counter() - t0)
        except Exception as e:
            o = {"error": f"{type(e).__name__}: {e}"[:300]}
        if rank == 0:
            out["c4"] = o
    # The ONE JSON line goes last: RCCL writes a banner (host name, library path) into the C stdio buffer, which a pipe only
    # flushes at exit -- behind everything Python printed, on every rank. Flush C stdio on all ranks, meet, then print.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def compact(o, wall_s):
    """One BASELINE config as a compact object of the headline line's `other_configs`."""
    r = o.get("roofline", {})
    c = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "step_ms": o.get("step_ms"), "dtype": o["dtype"],
         "steps": o["steps"], "warmup": o["warmup"], "workload": o["config"]["workload"],
         "roofline": {k: r.get(k) for k in ("kernel", "bound", "frac", "frac_useful", "achieved", "peak", "unit", "launches", "avg_launch_ms", "one_stream")},
         "pipeline": o.get("pipeline"),
         "step_mfma_util": o.get("step_mfma_util"),
         "top_kernels": [{k: q[k] for k in ("name", "ms_per_step", "tflops")} for q in o.get("kernels", [])[:6]],
         "stages": {k: o.get("stages", {}).get(k) for k in ("voxelize_ms", "condition_ms", "model_ms")},
         "wall_s_incl_setup": round(wall_s, 1)}
    if "hbm" in r:
        c["roofline"]["hbm_frac_algorithmic"] = r["hbm"]["frac"]
    if "convlstm" in o:
        c["convlstm"] = {k: o["convlstm"][k] for k in ("serial_critical_path_ms_per_step", "steps_in_series", "single_stream_sequence_ms", "single_stream_frames_per_s")}
    for k in ("ranks", "cpu_baseline", "n_gpus", "precision_check"):
        if k in o:
            c[k] = o[k]
    return c


def run_config(a, cname, cfg, rank, world, dist, detail):
    """Measure one BASELINE workload: W warm-up steps, exactly K timed steps between barrier + synchronize pairs (max over ranks).
    detail: the full line (stage rates, alt precision, CPU baseline); otherwise what `compact` keeps."""
    from evfly_amd import synthetic as syn, voxelizer
    from evfly_amd.distributed import gather_velocities, shard_row_counts
    B, T, epw, (Hs, Ws), dtype = cfg["streams"], cfg["windows"], cfg["epw"], cfg["sensor"], cfg["dtype"]
    composite = cfg["model"] == "composite"
    model, sd = build_model(cfg)
    # N > 1 ranks share one host's cores for the set-up: at most 32 streams per rank come from their seeds, the rest are rotated copies
    # ON THE DEVICE (voxelizer.tile_events == synthetic.make_batch's `distinct` layout: C4's 76.8 M events per rank cost 12 s alone and 48 s
    # eight at a time on an 8-core host -- most of it first-touch page faults of 1 GB of host arrays per rank); one rank generates every stream as before
    t_setup = time.perf_counter()
    distinct = 32 if dist is not None and B > 32 else B
    batch = syn.make_batch(distinct, T, Hs, Ws, epw, first_stream=rank * B)
    ev = voxelizer.upload_events(batch, prepare=distinct == B)
    del batch
    if distinct < B:
        ev = voxelizer.tile_events(ev, B, Hs, Ws)
    n_events = int(ev["offsets"][-1])
    torch.cuda.synchronize()
    setup_events_s = time.perf_counter() - t_setup
    desvel = torch.full((B * T, 1), 4.0, device="cuda")                            # run.py:255
    # sensor larger than the model's 260 x 346 (C3): the centre crop of run.py:345-350 is the voxelizer's region of interest
    roi = None if (Hs, Ws) == (H, W) else voxelizer.centre_crop_roi(Hs, Ws, (H, W))
    frames = torch.empty(B, T, H, W, device="cuda")
    counts = shard_row_counts(world * B, world, T)                                  # every rank: B streams (weak scaling)
    vel_host = torch.empty(world * B * T, 3).pin_memory() if composite else None    # §8d: "velocity rows on host-visible memory"
    # Two-stream pipeline (evfly_amd/pipeline.py): the velocity model of step i (ViT + LSTM, ~55 small launches) runs on a second HIP
    # stream behind the depth model's completion event while the first stream goes on with voxelize + condition + depth model of
    # step i + 1 -- the same kernels and bits as the back-to-back composite call, the small launches filling the large ones' gaps.
    # The timed region still closes with a device-wide synchronize: all K steps' velocities are in pinned host memory inside it.
    pipe = None
    if composite and not a.no_overlap and (a.overlap or cfg.get("overlap")):
        from evfly_amd.pipeline import StreamPipeline
        pipe = StreamPipeline(model)
    hips = [pipe.unet.hip(), pipe.vit.hip()] if pipe else [model.hip()]
    hip = hips[0]
    L = hip._L

    def publish(vel):
        vel_all = gather_velocities(vel, dist, counts=counts if dist is not None else None)
        vel_host.copy_(vel_all, non_blocking=True)                                    # lands before the closing synchronize
        return vel_all

    calls = [0]                      # every whole step this config executes (`steps_executed`: what a PMC pass divides by)

    def step():
        calls[0] += 1
        voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi)
        x = voxelizer.condition_frames(frames.view(B * T, H, W), out_hw=(H, W))
        if not composite:                                                             # C5: depth maps stay in HBM
            depth, _, _ = model.forward_streams(x, None, B, T)
            return depth
        if pipe:
            return pipe.step(x, desvel, B, T, after=publish)[2]
        vel, _ = model.forward_streams([x, desvel, [None, None], None], B, T)
        return publish(vel)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(a.warmup):
            out_dev = step()
        # An untimed, fully bracketed step first: the per-kernel breakdown (`kernels`, `conv_layers`, `stages`) and the
        # name of the dominant family. Bracketing EVERY launch with HIP events costs ~1 ms per step (two
        # hipEventRecord serialise each of the ~150 launches), so the timed region brackets only that family.
        for hh in hips:
            L.evfly_model_set_profile_filter(hh.h, None)
            L.evfly_model_profile_reset(hh.h)
            L.evfly_model_set_profiling(hh.h, 1)
        step(); sync()
        layers_all = []
        for hh in hips:
            L.evfly_model_set_profiling(hh.h, 0)
            layers_all += hh.profile()
        fam_ms = {}
        for p in layers_all:
            fam_ms[p["name"].split("/")[0]] = fam_ms.get(p["name"].split("/")[0], 0.0) + p["ms"]
        dom_name = max(fam_ms, key=fam_ms.get)

        def bracket(on):
            """bracket the dominant family's launches on EVERY handle (it may belong to the velocity model's one), or stop."""
            for hh in hips:
                if on:
                    L.evfly_model_profile_reset(hh.h)
                L.evfly_model_set_profile_filter(hh.h, dom_name.encode() if on else None)
                L.evfly_model_set_profiling(hh.h, 1 if on else 0)

        def bracketed():
            return [p for hh in hips for p in hh.profile()]
        bracket(True)
        # one event per step boundary on the launch stream (a record is ~1 us of host time, no synchronisation): the spread of
        # the K steps inside the one timed region
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        sync()
        t0 = time.perf_counter()
        marks[0].record()
        wall = []
        for i in range(a.steps):
            out_dev = step()
            marks[i + 1].record()
            wall.append(time.perf_counter())
        sync()
        dt = time.perf_counter() - t0
        if os.environ.get("BENCH_DEBUG_WALL"):      # developer switch: host time at which every step's launches were queued
            print("host ms from t0 after queueing step i:", [round(1e3 * (w - t0), 2) for w in wall], "end", round(1e3 * dt, 2), file=sys.stderr)
        bracket(False)
    timed_rec = bracketed()
    one_stream = None
    if pipe:
        # the same steps on ONE stream, outside the timed region: in the timed region kernels of the two streams share the chip, which
        # stretches the HIP-event durations of the dominant family; this pass times the family alone (3 steps)
        def step_serial():
            calls[0] += 1
            voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi)
            x = voxelizer.condition_frames(frames.view(B * T, H, W), out_hw=(H, W))
            depth, _, _ = pipe.unet.forward_streams(x, None, B, T)
            vel, _ = pipe.vit._run([depth, desvel, None], B, T, clip2x=1)
            return publish(vel)
        with torch.no_grad():
            step_serial(); sync()
            bracket(True)
            for _ in range(3):
                step_serial()
            sync()
            bracket(False)
        one_stream = bracketed()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    step_ms = {"min": round(per_step[0], 3), "median": round(per_step[len(per_step) // 2], 3), "max": round(per_step[-1], 3),
               "note": "GPU time between per-step events on the launch stream inside the timed region"}
    rank_ms, gather_us = None, None
    if dist is not None:
        mine = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = torch.empty(max(world, 1), device="cuda", dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        rank_ms = [round(1e3 * float(v) / a.steps, 3) for v in every.tolist()]   # where a scaling loss sits: per-rank ms/step
        tmax = mine.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if composite:
            # the one collective of the path, alone: 50 back-to-back all_gathers of this rank's velocity rows
            vel_mine = out_dev[rank * B * T:(rank + 1) * B * T].contiguous()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gather_velocities(vel_mine, dist, counts=counts); sync()
            g0.record()
            for _ in range(50):
                gather_velocities(vel_mine, dist, counts=counts)
            g1.record(); torch.cuda.synchronize()
            gather_us = round(1e3 * g0.elapsed_time(g1) / 50, 2)
    if composite:
        assert out_dev.shape == (world * B * T, 3), f"velocity rows {tuple(out_dev.shape)}"
        assert torch.isfinite(vel_host).all(), f"{int((~torch.isfinite(vel_host)).sum())} non-finite velocity values"
        assert torch.equal(vel_host, out_dev.cpu()), "published host copy differs from the device rows"
    else:
        assert out_dev.shape == (B * T, 1, H, W) and torch.isfinite(out_dev).all()

    # stage timings outside the timed region (torch events see the current stream, which is the one
    # every evfly_amd launch uses)
    def time_stage(fn, reps=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    vox_ms = cond_ms = vox1_ms = None
    if not a.pmc_pass:                                 # (a counter pass holds whole steps only)
        with torch.no_grad():
            vox_ms = time_stage(lambda: voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi))
            cond_ms = time_stage(lambda: voxelizer.condition_frames(frames.view(B * T, H, W), out_hw=(H, W)))
            # the same stage for a batch whose pass-1 tables do not exist yet (fresh events every call, SURVEY.md §8d's definition):
            # sortedness check + window ranges over the 8-B timestamps, then the accumulation
            ev_raw = {k: v for k, v in ev.items() if k not in ("starts", "unsorted", "skip_kernels")}
            vox1_ms = time_stage(lambda: voxelizer.voxelize_windows(ev_raw, Hs, Ws, out="f32", frames=frames, roi=roi))
    # bytes the timed kernel moves: x, y, p of every event once (5 B; the timestamps were consumed by pass 1 at upload) + the
    # (cropped) f32 frames once; with pass 1: + 8 B/event of timestamps = SURVEY.md §8d's 13 B/event
    vox_bytes = 5.0 * n_events + 4.0 * B * T * H * W
    vox1_bytes = 13.0 * n_events + 4.0 * B * T * H * W

    def families(recs):
        fam = {}
        for p in recs:
            f = fam.setdefault(p["name"].split("/")[0], dict(name=p["name"].split("/")[0], ms=0.0, flops=0.0, bytes=0.0, launches=0,
                                                              exec_flops=0.0, useful_flops=0.0))
            for k in ("ms", "flops", "bytes", "launches", "exec_flops", "useful_flops"):
                f[k] += p[k]
        return list(fam.values())
    timed = timed_rec                              # dominant family only, bracketed inside the timed region
    if not [p for p in timed if p["ms"] > 0]:      # (nothing bracketed: fall back to the untimed step's records of that family, x K)
        timed = [dict(p, **{k: p[k] * a.steps for k in ("ms", "flops", "bytes", "launches", "exec_flops", "useful_flops")})
                 for p in layers_all if p["name"].split("/")[0] == dom_name]
    dom = max(families(timed), key=lambda p: p["ms"])
    layers = layers_all                            # every launch site ("family/layer"), from the untimed step
    prof = families(layers)
    frames_per_step = world * B * T
    ms_per_step = 1e3 * dt / a.steps
    trunk = "reference-size ViT" if cfg["vit"] == "tiny" else "ViT-base trunk (widths 128/256, heads 4/8, 4+4 layers)"
    what = (f"OrigUNet+ConvLSTM -> LSTMNetVIT ({trunk})" if composite else "OrigUNet+ConvLSTM only (depth)")
    crop = "" if (Hs, Ws) == (H, W) else f" voxelized at {Hs}x{Ws}, centre-cropped to 260x346 (the voxelizer's region of interest),"
    out = {
        "metric": "event-frames/sec (260x346, 5 bins) event->depth->velocity fwd (voxelize + U-Net/ConvLSTM + ViT/LSTM)",
        "value": round(frames_per_step * a.steps / dt, 2), "unit": "event-frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"{cname}: {B} streams x {T} windows per GPU,{crop} 260x346, {epw} events/window, {what}, "
                               f"batch-as-time per stream; velocities end in pinned host memory" if composite else
                               f"{cname}: {B} streams x {T} windows per GPU (seq_len {T}), 260x346, {epw} events/window, {what}, "
                               f"batch-as-time per stream; depth maps stay in HBM",
                   "streams_per_gpu": B, "windows": T, "events_per_step_per_gpu": n_events, "sensor": [Hs, Ws], "vit_trunk": cfg["vit"] if composite else None,
                   "parallelism": f"streams sharded x{world}, all_gather of velocities" if world > 1 else "single GPU"},
        "step_ms": step_ms, "steps_executed": calls[0],
        "pipeline": ("two HIP streams: velocity model (ViT + LSTM) of step i under voxelize + depth model of step i + 1 "
                     "(evfly_amd/pipeline.py; --no-overlap = one stream)" if pipe else "one HIP stream"),
    }
    if dist is not None:
        out["ranks"] = {"ms_per_step_by_rank": rank_ms, "all_gather_us": gather_us, "setup_events_s": round(setup_events_s, 2),
                        "note": "ms/step of every rank between its own barrier + synchronize pairs (`ms_per_step` = their max); all_gather_us = "
                                "the velocity all_gather alone (RCCL, 50 back-to-back calls on this rank's rows)"}
    if rank == 0:
        peak = PEAK[dtype]
        sec = dom["ms"] * 1e-3
        # HBM bytes per launch of the dominant family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE, separate runs of this script at this config's default shape; FETCH doubled per the gfx950
        # calibration in profiles/README.md). PMC counters cannot be read inside this process, so the figure is null for other shapes.
        traffic, tnote = None, "no PMC summary for this shape/dtype"
        if cfg == CONFIGS[cname]:
            traffic, tnote = pmc_traffic(cname, dtype, dom["launches"] / a.steps)
        if dom["flops"]:
            # `achieved` = the flops the matrix cores EXECUTE per second in this kernel family (exec_flops, from the launch plans);
            # the algorithmic (direct-conv, SURVEY.md §8d: 2*M*N*K) rate is kept beside it as frac_algorithmic / `algorithmic`.
            ex_flops = dom["exec_flops"] if dom["exec_flops"] else dom["flops"]
            us_flops = dom["useful_flops"] if dom["useful_flops"] else dom["flops"]
            ex, us, tfl = ex_flops / sec / 1e12, us_flops / sec / 1e12, dom["flops"] / sec / 1e12
            gbs = dom["bytes"] / sec / 1e9
            wino = abs(dom["exec_flops"] - dom["flops"]) > 1e-6 * dom["flops"]
            out["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ex, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ex / peak, 4), "frac_useful": round(us / peak, 4), "frac_algorithmic": round(tfl / peak, 4),
                               "traffic": traffic, "traffic_note": tnote,
                               "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                               "flops_per_launch": ex_flops / dom["launches"],
                               "hbm": {"achieved_GBs_algorithmic": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4)},
                               "algorithmic": {"tflops": round(tfl, 2), "frac_of_peak": round(tfl / peak, 4),
                                               "flops_per_launch": dom["flops"] / dom["launches"],
                                               "bytes_per_launch": dom["bytes"] / dom["launches"]},
                               "note": ("achieved / frac = MFMA flops issued: Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32 issues 16/36 of the "
                                        "direct 3x3 convolution's multiplies, plus the padding of ragged tile rows / columns; frac_useful = "
                                        "without that padding; frac_algorithmic = direct-conv flops (SURVEY.md §8d) over the same time, "
                                        "above 1 because the convolution is computed with 2.25x fewer multiplies") if wino
                                       else "achieved = 2*M*N*K of the launches / their HIP-event time (direct implicit GEMM: issued = useful = algorithmic)"}
            # The same family against the roofline that bounds EACH layer: t_min(layer) = max(executed flops / MFMA peak, algorithmic bytes /
            # HBM peak) -- the shallow layers of this U-Net sit below the machine balance (e12: 131 flop/B against 312), the single MFMA
            # `frac` above prices them against a bound they cannot reach. From the bracketed untimed step (one launch per layer).
            lay = [p for p in layers if p["name"].startswith(dom["name"] + "/") and p["ms"] > 0]
            if lay:
                tmin = sum(max((p["exec_flops"] or p["flops"]) / (peak * 1e12), p["bytes"] / (HBM_PEAK_GBS * 1e9)) for p in lay)
                tact = sum(p["ms"] for p in lay) * 1e-3
                out["roofline"]["per_layer_bound"] = {
                    "frac": round(tmin / tact, 4), "hbm_bound_layers": sum(1 for p in lay if p["bytes"] / (HBM_PEAK_GBS * 1e9) > (p["exec_flops"] or p["flops"]) / (peak * 1e12)),
                    "layers": len(lay), "note": "sum over the family's layers of max(flops / MFMA peak, algorithmic bytes / 8 TB/s), divided by their measured time "
                                                "(bracketed untimed step); `frac` above stays flops / MFMA peak for the whole family"}
        else:
            gbs = dom["bytes"] / sec / 1e9
            out["roofline"] = {"kernel": dom["name"], "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                               "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4)}
        if one_stream and dom["flops"]:
            od = max(families(one_stream), key=lambda p: p["ms"])
            osec = od["ms"] * 1e-3
            oex = (od["exec_flops"] or od["flops"]) / osec / 1e12
            out["roofline"]["one_stream"] = {"achieved": round(oex, 2), "frac": round(oex / peak, 4), "launches": od["launches"],
                                             "avg_launch_ms": round(od["ms"] / od["launches"], 4),
                                             "note": "the same family over 3 one-stream steps outside the timed region (no kernels of "
                                                     "the velocity model beside it)"}
        model_ms = sum(p["ms"] for p in prof)
        # matrix-core utilisation of the WHOLE step: every MFMA kernel's issued flops (one bracketed step) over the timed ms/step
        step_exec = sum((p["exec_flops"] or p["flops"]) for p in prof if p["flops"] and not p["name"].startswith(("vit_attention", "vit_grouped", "lstm_rec", "e11_direct", "unet_out")))
        out["step_mfma_util"] = round(step_exec / (ms_per_step * 1e-3) / 1e12 / peak, 4)      # (this rank's kernels over the step time)
        # the same on USEFUL flops (Winograd tile padding not counted): what the north star's ">= 60 % MFMA utilisation" reads on
        step_useful = sum((p["useful_flops"] or p["exec_flops"] or p["flops"]) for p in prof if p["flops"] and not p["name"].startswith(("vit_attention", "vit_grouped", "lstm_rec", "e11_direct", "unet_out")))
        out["step_mfma_util_useful"] = round(step_useful / (ms_per_step * 1e-3) / 1e12 / peak, 4)
        out["breakdown_note"] = ("kernels / conv_layers / stages.model_ms: one untimed step with every launch bracketed by HIP "
                                 "events; roofline: the dominant family bracketed inside the timed region")
        out["kernels"] = [{"name": p["name"], "ms_per_step": round(p["ms"], 3), "launches_per_step": p["launches"],
                           "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2) if p["flops"] and p["ms"] else None,
                           "gbs_algorithmic": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1) if p["ms"] else None}
                          for p in sorted(prof, key=lambda p: -p["ms"])]
        out["model_ms_per_step"] = round(model_ms, 3)
        if os.environ.get("BENCH_DUMP_LAYERS"):      # developer switch: every launch site of the bracketed step, not only the families
            out["layers"] = [{"name": p["name"], "ms": round(p["ms"], 4), "launches": p["launches"],
                              "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 1) if p["flops"] and p["ms"] else None,
                              "gbs": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1) if p["ms"] else None,
                              "slot_fill": round(p["useful_flops"] / p["exec_flops"], 4) if p.get("exec_flops") and p.get("useful_flops") else None,
                              "issued_frac_of_peak": round(p["exec_flops"] / (p["ms"] * 1e-3) / 1e12 / PEAK[dtype], 3) if p.get("exec_flops") and p["ms"] else None}
                             for p in layers]
        out["conv_layers"] = [{"name": p["name"], "ms_per_step": round(p["ms"], 3),
                               "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 1)}
                              for p in layers if p["name"].startswith(dom["name"] + "/")]
        if vox_ms is None:
            out["stages"] = {"model_ms": round(model_ms, 3)}
        else:
            out["stages"] = {"voxelize_ms": round(vox_ms, 4), "voxelize_bytes_moved": vox_bytes,
                             "voxelize_GBs": round(vox_bytes / (vox_ms * 1e-3) / 1e9, 1),
                             "voxelize_frac_of_hbm_peak": round(vox_bytes / (vox_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "voxelize_with_pass1_ms": round(vox1_ms, 4), "voxelize_with_pass1_bytes": vox1_bytes,
                             "voxelize_with_pass1_GBs": round(vox1_bytes / (vox1_ms * 1e-3) / 1e9, 1),
                             "voxelize_with_pass1_frac_of_hbm_peak": round(vox1_bytes / (vox1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             "condition_ms": round(cond_ms, 4), "model_ms": round(model_ms, 3),
                             "voxelize_note": "voxelize_ms = what a timed step runs: the events are resident together with the voxelizer's pass-1 "
                                              "tables (evfly_voxel_prepare at upload: per-stream sortedness, window -> event ranges), so the step's "
                                              "kernel k_vox_band reads x, y, p (5 B/event) and writes the frames; voxelize_with_pass1 = the same "
                                              "call on a batch without tables (k_check_sorted + k_window_ranges read the 8-B timestamps first: "
                                              "13 B/event, SURVEY.md §8d's definition for fresh events); fractions = those bytes / ms / 8 TB/s"}
        if detail and not a.no_stage_rates:
            # labelled per-stage rates ("publish both P-only and V+D+P"): each stage alone on the same batch,
            # inputs resident, torch events on the launch stream. V = voxelize + condition, D = OrigUNet + ConvLSTM,
            # P = ViT + LSTM on 260x346 depth images (resize to 60x90 inside, like the composite's hand-off).
            fper = B * T
            rates = {"v_only": {"frames_per_s": round(fper / ((vox1_ms + cond_ms) * 1e-3), 1), "ms": round(vox1_ms + cond_ms, 4),
                                "what": "voxelize FRESH events (pass 1 over the timestamps included: 13 B/event, stages.voxelize_with_pass1_*) + crop/q97 "
                                        "conditioning; the timed step itself replays resident events whose pass-1 tables were prepared at upload "
                                        "(stages.voxelize_ms)",
                                "replayed_events": {"frames_per_s": round(fper / ((vox_ms + cond_ms) * 1e-3), 1), "ms": round(vox_ms + cond_ms, 4)}}}
            with torch.no_grad():
                x = voxelizer.condition_frames(frames.view(B * T, H, W), out_hw=(H, W))
                unet = model.origunet if composite else model
                unet.set_compute_dtype(dtype)
                d_ms = time_stage(lambda: unet.forward_streams(x, None, B, T), reps=3)
                rates["d_only"] = {"frames_per_s": round(fper / (d_ms * 1e-3), 1), "ms": round(d_ms, 3), "what": "OrigUNet + ConvLSTM (depth) on conditioned frames"}
                if composite:
                    depth, _, _ = unet.forward_streams(x, None, B, T)
                    vit = model.vitfly_vitlstm
                    vit.set_compute_dtype(dtype)
                    p_ms = time_stage(lambda: vit.forward_streams([depth, desvel, None], B, T), reps=3)
                    rates["p_only"] = {"frames_per_s": round(fper / (p_ms * 1e-3), 1), "ms": round(p_ms, 3),
                                       "what": f"LSTMNetVIT ({cfg['vit']} trunk) on 260x346 depth images (the metric string's 'ViT fwd')"}
                    del depth
            rates["v_d_p"] = {"frames_per_s": out["value"] / world, "ms": round(ms_per_step, 3), "what": "the headline: whole path per GPU"}
            out["stage_rates"] = rates
        if not composite:
            # C5: the serial part of the ConvLSTM -- T dependent (hidden-side GEMM + gate kernel) pairs per chunk --
            # and the latency of ONE stream's 16-frame sequence
            serial = sum(p["ms"] for p in prof if p["name"] in ("convlstm_h_gemm", "convlstm_gates", "convlstm_seq"))
            xg = sum(p["ms"] for p in prof if p["name"] == "convlstm_x_gemm")
            one_ms = None
            if not a.no_stage_rates:
                with torch.no_grad():
                    x1 = voxelizer.condition_frames(frames.view(B * T, H, W)[:T], out_hw=(H, W))
                    one_ms = time_stage(lambda: model.forward_streams(x1, None, 1, T), reps=5)
            out["convlstm"] = {"serial_critical_path_ms_per_step": round(serial, 3), "steps_in_series": T,
                               "batched_input_gemm_ms_per_step": round(xg, 3),
                               "single_stream_sequence_ms": round(one_ms, 3) if one_ms else None,
                               "single_stream_frames_per_s": round(T / (one_ms * 1e-3), 1) if one_ms else None,
                               "note": f"serial = the {T} dependent recurrence steps of one {B}-stream chunk (from ~1 k state rows: ONE launch, k_clstm16_coop, weights "
                                       f"resident in groups of 16 CUs, a group barrier per step; below: a GEMM launch per step with the cell update in its epilogue); single_stream = one stream's "
                                       f"{T}-frame sequence through the U-Net alone (latency-bound: {T} frames do not fill the chip)"}
        mf = sum(p["flops"] for p in prof if p["flops"])
        out["mfma_flops_per_frame"] = mf / (B * T)
        if detail and world == 1 and cname == "C2" and dtype == "f32" and not a.no_alt:
            # Informational second precision mode, NOT the headline `value`: the bf16 pipeline (bf16 activations in HBM, bf16
            # MFMA, fp32 accumulate). Same inputs, same weights; the velocity deviation from the exact-fp32 run is reported with it.
            vel_f32 = out_dev.clone()
            for mm in (model, model.origunet, model.vitfly_vitlstm):
                mm.set_compute_dtype("bf16")
            with torch.no_grad():
                for _ in range(max(1, a.warmup)):
                    v3 = step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    v3 = step()
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t1
            out["alt_precision"] = {"dtype": "bf16", "value": round(B * T * a.steps / dt3, 2), "unit": "event-frames/s",
                                    "ms_per_step": round(1e3 * dt3 / a.steps, 3),
                                    "max_rel_dev_velocity_vs_f32": float(((v3 - vel_f32).abs().max() / vel_f32.abs().max()).item()),
                                    "note": "bf16 pipeline (BASELINE configs C3 / C5 run in it); informational, not the headline"}
            for mm in (model, model.origunet, model.vitfly_vitlstm):
                mm.set_compute_dtype("f32")
        if composite and dtype == "bf16" and not a.pmc_pass:
            # the bf16 pipeline against the exact-fp32 one on the SAME inputs, 8 sampled streams (the first, evenly spaced, the last): the
            # velocity deviation a user of this config pays for the precision mode (C2's `alt_precision` is the same figure for C2)
            pick = sorted(set(int(round(i * (B - 1) / 7)) for i in range(8)))
            with torch.no_grad():
                voxelizer.voxelize_windows(ev, Hs, Ws, out="f32", frames=frames, roi=roi)
                xs = voxelizer.condition_frames(frames.view(B * T, H, W), out_hw=(H, W)).view(B, T, 1, H, W)[pick].reshape(len(pick) * T, 1, H, W).contiguous()
                dv = desvel[:len(pick) * T]
                v16, _ = model.forward_streams([xs, dv, [None, None], None], len(pick), T)
                for mm in (model, model.origunet, model.vitfly_vitlstm):
                    mm.set_compute_dtype("f32")
                v32, _ = model.forward_streams([xs, dv, [None, None], None], len(pick), T)
                for mm in (model, model.origunet, model.vitfly_vitlstm):
                    mm.set_compute_dtype(dtype)
            out["precision_check"] = {"streams": pick, "max_rel_dev_velocity_vs_f32": float(((v16 - v32).abs().max() / v32.abs().max()).item()),
                                      "rms_rel_dev_velocity_vs_f32": float(((v16 - v32).pow(2).mean().sqrt() / v32.pow(2).mean().sqrt()).item()),
                                      "note": f"bf16 pipeline vs the exact-fp32 pipeline, same inputs and weights, {len(pick)} of the {B} streams x {T} windows"}
            del xs, v16, v32
        if detail and not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd, cfg, a.cpu_seconds)
        elif not detail and not a.no_cpu_baseline and a.side_cpu_seconds > 0:
            # side configs: one thread count (the one that wins on these hosts), a few seconds -- a reported baseline, not a study
            out["cpu_baseline"] = cpu_baseline(sd, cfg, a.side_cpu_seconds, only_mid=True)
    # release this config's device memory (events, frames, the model handle's arena) before the next one is built
    del model, hip, hips, pipe, ev, frames, out_dev, desvel
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
