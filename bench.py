#!/usr/bin/env python3
"""bench.py -- Msamples/s of the render hot path on MI355X (BASELINE.json metric).

One step = one frame of the workload: caustics scene (scenes/caustics), 1920x1080, 256 spp, 200 000 photon indices emitted
and built into the photon map before the timed region (BASELINE.json configs[2], the configuration the metric is quoted
on).  1 sample = one radiance() evaluation: primary ray + the whole path (bounces, shadow rays, photon gathers).

N GPUs: one process per GPU (torch.distributed / RCCL).  The frame is cut into 16-row stripes dealt round-robin to the
ranks (strong scaling: the frame is fixed); every step ends with one RCCL gather of the float-RGB stripes to rank 0, which
is inside the timed region.  Timing: barrier + torch.cuda.synchronize() on both sides of exactly K steps, MAX over ranks.

Extra objects on the JSON line: "roofline" (algorithmic HBM bytes of the render kernel / its HIP-event duration, against
8 TB/s) and, at N=1, "cpu_baseline" (the CPU oracle, OpenMP on the box's host cores, on a bounded sample of the same
workload: full-width rows spread over the frame at the full 256 spp).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes_per_sample(counters, n_samples, spp):
    """B = 32 V + 36 T + 96 H + 36 P + 12/spp with V node visits (trace + visible), T ray-triangle tests,
    H shaded hits, P photon candidates per sample, counted by the CPU oracle on the same workload."""
    v = (counters[0] + counters[1]) / n_samples
    t = counters[2] / n_samples
    hh = counters[3] / n_samples
    p = counters[4] / n_samples
    mix = {"V": v, "T": t, "H": hh, "P": p, "V_trace": counters[0] / n_samples, "V_shadow": counters[1] / n_samples,
           "T_shadow": counters[8] / n_samples, "T_trace": (counters[2] - counters[8]) / n_samples}
    return 32 * v + 36 * t + 96 * hh + 36 * p + 12.0 / spp, mix


def measured_traffic(scene, w, h, spp, photons, world, mode):
    """HBM bytes of one frame from the TCC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), as committed under
    profiles/ for exactly this workload; bench.py cannot sit under the profiler itself.  FETCH_SIZE is doubled as
    MI355X_MICROARCH.md (HBM section) prescribes for gfx950."""
    import glob
    best = None
    import re
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]   # r01_v10 after r01_v9
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), key=natural):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        k = d.get("workload_key")
        if k == {"scene": scene, "frame": [w, h], "spp": spp, "photons": photons, "n_gpus": world, "mode": mode}:
            best = (path, d)
    if best is None:
        return None
    path, d = best
    per_kernel = {k: 2.0 * v["FETCH_SIZE_KB"] * 1024 + v["WRITE_SIZE_KB"] * 1024 for k, v in d.get("per_kernel", {}).items()}
    return {"traffic": 2.0 * d["frame_fetch_bytes_uncorrected"] + d["frame_write_bytes"], "traffic_unit": "bytes per frame (one pass of the pipeline)",
            "traffic_source": os.path.relpath(path, ROOT) + ": 2 x FETCH_SIZE + WRITE_SIZE", "_per_kernel": per_kernel}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="caustics")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--photons", type=int, default=200000)
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the CPU-baseline sample (-1: sized for ~20 s, 0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pool", type=int, default=0, help="path slots of the wavefront pool (0: library default)")
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "rounds", "megakernel"])
    args = ap.parse_args()

    import torch
    import gi_raytracer_amd as gi
    import parity_checks as pc
    from gi_raytracer_amd.sharding import STRIPE_H, FrameGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render hot path has no CPU fallback")
    # rehearsal of the N > 1 path on a one-GPU box: GI_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo (RCCL cannot
    # run two ranks on one device); numbers from such a run mean nothing, it only exercises the code path
    rehearsal = os.environ.get("GI_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- setup (untimed): scene tables, octree, photon emission on the device, photon octree
    scene = pc.load_scene(args.scene)
    rt = gi.RayTracer(local_rank).setScene(scene)
    if rehearsal and world > 1:
        rt.set_pool_slots(args.width * args.height * args.spp // world // 2 + 1)   # several ranks share one device's memory
    rt.set_stream(torch.cuda.current_stream().cuda_stream)
    rt.set_render_mode(args.mode)
    if args.pool > 0:
        rt.set_pool_slots(args.pool)
    t0 = time.time()
    n_photons = 0
    if args.photons > 0 and scene.desc().n_light > 0:
        ph, tries = rt.tracePhotons(args.photons)
        n_photons = len(ph)
    setup_s = time.time() - t0

    w, h, spp = args.width, args.height, args.spp
    stripe_h = STRIPE_H if world > 1 else h
    p = rt.params(w, h, stripe_h=stripe_h, rank=rank, world=world, min_samples=spp, max_samples=spp)
    rows = rt.local_rows(p)
    fg = FrameGather(torch, dist, w, h, stripe_h, rank, world, dev, torch.float32)
    assert rows == len(fg.rows[rank])

    kernel_ms = []
    stage_ms = []

    def step(record):
        rt.run_device(p, fg.local.data_ptr(), f64=False)
        fg.gather()                                    # N > 1: one RCCL gather of the stripes to rank 0 (inside the timed region)
        if record:
            kernel_ms.append(rt.last_render_ms()[0])   # HIP events on the launch stream (synchronises on the second event)
            stage_ms.append(rt.last_stage_ms())        # HIP events around every launch, summed per pipeline stage

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        km = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms_avg = float(km.item())
    else:
        kernel_ms_avg = float(np.mean(kernel_ms))

    if rank == 0:
        samples_per_step = w * h * spp
        value = samples_per_step * args.steps / elapsed / 1e6
        img = fg.frame.cpu().numpy()
        out = {
            "metric": "Msamples/sec (primary+path rays) at 1080p", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"scenes/{args.scene} {w}x{h} {spp} spp, {args.photons} photon indices ({n_photons} photons stored) + gather",
                       "frame": [w, h], "spp": spp, "photons_stored": n_photons, "sharding": f"{STRIPE_H}-row stripes round-robin over {world} GPU(s), RCCL gather to rank 0" if world > 1 else "single GPU",
                       "setup_s_untimed": round(setup_s, 3), "mean_radiance": float(img.mean())},
        }
        # ---- CPU baseline + algorithmic bytes (oracle = the checker, timed on the host cores; N=1 only)
        cpu = None
        bytes_per_sample, mix = None, None
        if world == 1 and not args.no_cpu and args.cpu_rows != 0:
            import oracle_lib as ol
            o = pc.oracle_for(scene)
            o.set_photons(scene.photon_tables()["photons"]).build_photon_map()
            cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("GI_CPU_THREADS", "16"))))   # the box's CPU share for one GPU
            n_rows = args.cpu_rows
            if n_rows < 0:
                # calibrate on one row per core, then size the sample for ~20 s of CPU work
                probe = np.unique(np.linspace(0, h - 1, cores).round().astype(np.int32))
                tc = time.perf_counter()
                o.render_rows(w, h, probe, spp, rt.seed, cores)
                t_probe = time.perf_counter() - tc
                n_rows = int(min(h, max(cores, len(probe) * 20.0 / max(t_probe, 1e-6))))
            rows_sel = np.unique(np.linspace(0, h - 1, n_rows).round().astype(np.int32))
            tc = time.perf_counter()
            lin, cnt = o.render_rows(w, h, rows_sel, spp, rt.seed, cores)
            cpu_s = time.perf_counter() - tc
            n_s = len(rows_sel) * w * spp
            cpu = {"value": n_s / cpu_s / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": f"{len(rows_sel)} full-width rows spread evenly over the {w}x{h} frame at {spp} spp = {n_s} samples in {cpu_s:.1f} s (OpenMP oracle, rows dealt dynamically to {cores} threads)"}
            bytes_per_sample, mix = algorithmic_bytes_per_sample(cnt, n_s, spp)
            # parity of the timed frame against the oracle on exactly those rows
            rmse = float(np.sqrt(((img[rows_sel].astype(np.float64) - lin[rows_sel]) ** 2).mean()))
            out["config"]["rmse_vs_oracle_on_cpu_rows"] = rmse
        if bytes_per_sample is None:
            # per-sample mix measured by the oracle on this workload (profiles/*_bench.json: per_sample_mix); used when the CPU leg is skipped
            bytes_per_sample, mix = 32 * 99.1 + 36 * 56.1 + 96 * 0.90 + 36 * 71.9 + 12.0 / spp, None
        local_samples = rows * w * spp
        stages = {k: float(np.mean([s[k] for s in stage_ms])) for k in stage_ms[0]} if stage_ms else {}
        pipeline_ms = sum(stages.values()) if sum(stages.values()) > 0 else kernel_ms_avg   # megakernel mode: one launch
        achieved = bytes_per_sample * local_samples / (pipeline_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None,   # filled below from the committed PMC measurement of this very workload, if there is one
                           "kernel": "streaming wavefront pipeline of one frame: k_st_regen, k_st_trace, k_st_shade, k_st_gkeys + 2 radix sorts, k_st_gather, k_st_finish, k_st_accum"
                                     if args.mode == "wavefront" else args.mode,
                           "kernel_ms": pipeline_ms, "frame_ms_event_to_event": kernel_ms_avg, "stage_ms": stages,
                           "algorithmic_bytes_per_sample": bytes_per_sample, "per_sample_mix": mix}
        if stages and mix is not None and sum(stages.values()) > 0:
            dom = max(stages, key=stages.get)
            # algorithmic bytes of the stages that walk the scene octree (SURVEY 8(d) terms restricted to that stage)
            stage_bytes = {"trace": 32 * mix["V_trace"] + 36 * mix["T_trace"], "shade": 32 * mix["V_shadow"] + 36 * mix["T_shadow"] + 96 * mix["H"],
                           "gather": 36 * mix["P"]}
            if dom in stage_bytes:
                a = stage_bytes[dom] * local_samples / (stages[dom] * 1e-3) / 1e9
                out["roofline"]["dominant"] = {"kernel": "k_st_" + dom, "ms_per_frame": stages[dom], "algorithmic_bytes_per_sample": stage_bytes[dom],
                                               "achieved": a, "frac": a / HBM_PEAK_GBS}
        tr = measured_traffic(args.scene, w, h, spp, args.photons, world, args.mode)
        if tr is not None:
            per_kernel = tr.pop("_per_kernel")
            out["roofline"].update(tr)
            dom = out["roofline"].get("dominant")
            if dom and dom["kernel"] in per_kernel:      # the dominant kernel's own HBM bytes per frame (all its launches)
                dom["traffic"] = per_kernel[dom["kernel"]]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
