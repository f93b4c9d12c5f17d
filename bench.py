#!/usr/bin/env python3
"""bench.py -- Msamples/s of the render hot path on MI355X (BASELINE.json metric).

One step = one frame of the workload: caustics scene (scenes/caustics), 1920x1080, 256 spp, 200 000 photon indices emitted
and built into the photon map before the timed region (BASELINE.json configs[2], the configuration the metric is quoted
on).  1 sample = one radiance() evaluation: primary ray + the whole path (bounces, shadow rays, photon gathers).

N GPUs: one process per GPU (torch.distributed / RCCL).  `python bench.py --gpus N` starts the N ranks itself (fresh child
processes through torch.distributed.run, before this process touches a GPU); under torch.distributed.run it is one of the
ranks.  The frame is cut into 16-row stripes dealt round-robin to the ranks (strong scaling: the frame is fixed); every step
ends with one RCCL gather of the float-RGB stripes to rank 0, inside the timed region.  Timing: barrier +
torch.cuda.synchronize() on both sides of exactly K steps, MAX over ranks.

Extra objects on the JSON line:
  "roofline"      algorithmic HBM-class bytes of the pipeline / its HIP-event duration against 8 TB/s
                  ("algorithmic_frac"; those bytes are served from LDS / L2, so this is a work rate, not a bus load), the PMC-measured
                  HBM traffic of the same frame and what share of the HBM peak it is ("hbm_measured_frac"), per stage and for the dominant kernel;
  "cpu_baseline"  at N=1: the CPU oracle, OpenMP on the box's host cores, on a bounded sample of the same workload;
  "other_configs" at N=1: BASELINE configs 2 (cornell 512x512x64 spp, closed box) and 4 (cornell + glass teapot 1080p x 256 spp, 200 k photons).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
SCN = {"caustics": "scenes/caustics/caustics.scn", "cornell": "scenes/cornell/test.scn", "teapot": "scenes/cornell/teapot.scn", "test_scene": "scenes/test_scene/test.scn",
       "caustics_02": "scenes/caustics_02/caustics.scn", "fog": "scenes/fog/fog.scn", "spheres": "scenes/spheres/spheres.scn", "textures": "scenes/textures/tex.scn"}
MIX_FILE = os.path.join(ROOT, "profiles", "workload_mix.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="caustics")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--photons", type=int, default=200000)
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the CPU-baseline sample (-1: sized for ~12 s, 0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the other_configs lines (BASELINE configs 2 and 4)")
    ap.add_argument("--pool", type=int, default=0, help="path slots of the wavefront pool (0: library default)")
    ap.add_argument("--mode", default="wavefront", choices=["wavefront", "rounds", "megakernel"])
    ap.add_argument("--write-mix", action="store_true", help="record the oracle-counted per-sample mix of this workload in profiles/workload_mix.json")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` typed as is: start the N ranks as fresh child processes (torch.distributed.run) and relay rank 0's JSON
    line.  Nothing in this process has touched a GPU (torch is not even imported yet), and nothing is exec'ed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def algorithmic_bytes_per_sample(mix, spp):
    """B = 32 V + 36 T + 96 H + 36 P + 12/spp with V node visits (trace + visible), T ray-triangle tests,
    H shaded hits, P photon candidates per sample of the REFERENCE algorithm, counted by the CPU oracle on the same workload."""
    return 32 * mix["V"] + 36 * mix["T"] + 96 * mix["H"] + 36 * mix["P"] + 12.0 / spp


def mix_from_counters(counters, n_samples):
    c = [float(v) for v in counters]
    return {"V": (c[0] + c[1]) / n_samples, "T": c[2] / n_samples, "H": c[3] / n_samples, "P": c[4] / n_samples, "V_trace": c[0] / n_samples,
            "V_shadow": c[1] / n_samples, "T_shadow": c[8] / n_samples, "T_trace": (c[2] - c[8]) / n_samples}


def workload_key(scene, w, h, spp, photons):
    return f"{scene} {w}x{h} {spp}spp {photons}ph"


def stored_mix(key):
    """Per-sample mix of a workload as the oracle counted it in an earlier CPU leg (profiles/workload_mix.json); None when this workload was never counted."""
    try:
        with open(MIX_FILE) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def measured_traffic(scene, w, h, spp, photons, world, mode):
    """HBM bytes of one frame from the TCC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), as committed under
    profiles/ for exactly this workload; bench.py cannot sit under the profiler itself.  FETCH_SIZE is doubled as
    MI355X_MICROARCH.md (HBM section) prescribes for gfx950."""
    import glob
    import re
    best = None
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]   # r01_v10 after r01_v9, r02 after r01
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), key=natural):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("workload_key") == {"scene": scene, "frame": [w, h], "spp": spp, "photons": photons, "n_gpus": world, "mode": mode}:
            best = (path, d)
    if best is None:
        return None
    path, d = best
    per_kernel = {k: 2.0 * v["FETCH_SIZE_KB"] * 1024 + v["WRITE_SIZE_KB"] * 1024 for k, v in d.get("per_kernel", {}).items()}
    return {"traffic": 2.0 * d["frame_fetch_bytes_uncorrected"] + d["frame_write_bytes"], "traffic_unit": "bytes per frame (one pass of the pipeline)",
            "traffic_source": os.path.relpath(path, ROOT) + ": 2 x FETCH_SIZE + WRITE_SIZE", "_per_kernel": per_kernel}


def roofline_of(scene_name, w, h, spp, photons, world, mode, local_samples, stages, kernel_ms_avg, mix):
    """The roofline object for one workload; mix = per-sample counts of the reference algorithm (oracle), or None when unknown."""
    pipeline_ms = sum(stages.values()) if stages and sum(stages.values()) > 0 else kernel_ms_avg   # megakernel mode: one launch
    r = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "kernel": "streaming wavefront pipeline of one frame: k_st_trace (+ path start), k_st_shade, k_st_shadow, k_st_compact (+ gather keys), sorts, k_st_gather, k_st_finish, k_st_accum"
                   if mode == "wavefront" else mode,
         "kernel_ms": pipeline_ms, "frame_ms_event_to_event": kernel_ms_avg, "stage_ms": stages}
    if mix is not None:
        b = algorithmic_bytes_per_sample(mix, spp)
        a = b * local_samples / (pipeline_ms * 1e-3) / 1e9
        r.update({"achieved": a, "frac": a / HBM_PEAK_GBS, "algorithmic_frac": a / HBM_PEAK_GBS, "algorithmic_bytes_per_sample": b, "per_sample_mix": mix,
                  "note": "achieved / frac follow SURVEY 8(d): bytes the reference algorithm touches per sample / pipeline time.  The scene and photon "
                          "tables are LDS / L2 resident, so these bytes do not cross the HBM bus and the fraction can exceed 1 (a work rate, not a bus "
                          "load; content-box culling also skips node visits the reference makes): hbm_measured_frac (PMC) is the bus load; the kernels "
                          "are latency / issue bound (profiles/*_sq_pmc.json)."})
        if stages and sum(stages.values()) > 0:
            stage_bytes = {"trace": 32 * mix["V_trace"] + 36 * mix["T_trace"], "shade": 32 * mix["V_shadow"] + 36 * mix["T_shadow"] + 96 * mix["H"], "gather": 36 * mix["P"]}
            r["stage_algorithmic_frac"] = {k: (v * local_samples / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBS if stages.get(k, 0) > 0 else None) for k, v in stage_bytes.items()}
            dom = max(stage_bytes, key=lambda k: stages.get(k, 0.0))
            if stages.get(dom, 0) > 0:
                a = stage_bytes[dom] * local_samples / (stages[dom] * 1e-3) / 1e9
                r["dominant"] = {"kernel": "k_st_" + dom, "ms_per_frame": stages[dom], "algorithmic_bytes_per_sample": stage_bytes[dom], "achieved": a, "frac": a / HBM_PEAK_GBS}
    tr = measured_traffic(scene_name, w, h, spp, photons, world, mode)
    if tr is not None:
        per_kernel = tr.pop("_per_kernel")
        r.update(tr)
        r["hbm_measured_frac"] = tr["traffic"] / (pipeline_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        dom = r.get("dominant")
        if dom and dom["kernel"] in per_kernel:      # the dominant kernel's own HBM bytes per frame (all its launches)
            dom["traffic"] = per_kernel[dom["kernel"]]
            dom["hbm_measured_frac"] = dom["traffic"] / (dom["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return r


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import gi_raytracer_amd as gi
    from gi_raytracer_amd.sharding import STRIPE_H, FrameGather

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_workload(scene_name, w, h, spp, photons, steps, warmup):
        """Setup (untimed: scene tables, octree, photon emission on the device, photon octree), W warm-up frames, K timed frames."""
        scene = gi.Scene.load(os.path.join(ROOT, SCN.get(scene_name, scene_name))).rebuild()
        rt = gi.RayTracer(local_rank).setScene(scene)
        if rehearsal and world > 1:
            rt.set_pool_slots(w * h * spp // world // 2 + 1)   # several ranks share one device's memory
        rt.set_stream(torch.cuda.current_stream().cuda_stream)
        rt.set_render_mode(args.mode)
        if args.pool > 0:
            rt.set_pool_slots(args.pool)
        t0 = time.time()
        n_photons = 0
        if photons > 0 and scene.desc().n_light > 0:
            n_photons, _ = rt.tracePhotonsOnDevice(photons)    # emission, octree and candidate ranges on the device: the photons never visit the host
        setup_s = time.time() - t0
        stripe_h = STRIPE_H if world > 1 else h
        p = rt.params(w, h, stripe_h=stripe_h, rank=rank, world=world, min_samples=spp, max_samples=spp)
        rows = rt.local_rows(p)
        fg = FrameGather(torch, dist, w, h, stripe_h, rank, world, dev, torch.float32)
        assert rows == len(fg.rows[rank])
        kernel_ms, stage_ms = [], []

        def step(record):
            rt.run_device(p, fg.local.data_ptr(), f64=False)
            fg.gather()                                    # N > 1: one RCCL gather of the stripes to rank 0 (inside the timed region)
            if record:
                kernel_ms.append(rt.last_render_ms()[0])   # HIP events on the launch stream (synchronises on the second event)
                stage_ms.append(rt.last_stage_ms())        # HIP events around every launch, summed per pipeline stage

        for _ in range(warmup):
            step(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        sync()
        elapsed = time.perf_counter() - t0
        kernel_ms_avg = float(np.mean(kernel_ms))
        if world > 1:
            tt = torch.tensor([elapsed, kernel_ms_avg], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed, kernel_ms_avg = float(tt[0].item()), float(tt[1].item())
        stages = {k: float(np.mean([s[k] for s in stage_ms])) for k in stage_ms[0]} if stage_ms else {}
        img = fg.frame.cpu().numpy() if rank == 0 else None
        return {"scene": scene, "rt": rt, "elapsed": elapsed, "kernel_ms": kernel_ms_avg, "stages": stages, "img": img, "rows": rows, "n_photons": n_photons, "photons_asked": photons, "setup_s": setup_s}

    def cpu_leg(res, scene_name, w, h, spp, budget_s, cores):
        """The oracle (the checker) timed on the host cores on full-width rows spread over the frame; returns (cpu_baseline, mix, rmse)."""
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import parity_checks as pc
        o = pc.oracle_for(res["scene"])
        rt = res["rt"]
        ph = np.zeros((0, 9))
        if res["n_photons"] > 0:
            ph, _ = rt.tracePhotons(res["photons_asked"])      # the same keyed photon set, this time brought to the host for the oracle
            assert len(ph) == res["n_photons"]
        o.set_photons(ph).build_photon_map()
        n_rows = args.cpu_rows
        if n_rows < 0 or budget_s != CPU_BUDGET_S:
            probe = np.unique(np.linspace(0, h - 1, cores).round().astype(np.int32))   # calibrate on one row per core, then size the sample
            tc = time.perf_counter()
            o.render_rows(w, h, probe, spp, rt.seed, cores)
            t_probe = time.perf_counter() - tc
            n_rows = int(min(h, max(cores, len(probe) * budget_s / max(t_probe, 1e-6))))
        rows_sel = np.unique(np.linspace(0, h - 1, n_rows).round().astype(np.int32))
        tc = time.perf_counter()
        lin, cnt = o.render_rows(w, h, rows_sel, spp, rt.seed, cores)
        cpu_s = time.perf_counter() - tc
        n_s = len(rows_sel) * w * spp
        cpu = {"value": n_s / cpu_s / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": f"{len(rows_sel)} full-width rows spread evenly over the {w}x{h} frame at {spp} spp = {n_s} samples in {cpu_s:.1f} s (OpenMP oracle, rows dealt dynamically to {cores} threads)"}
        rmse = float(np.sqrt(((res["img"][rows_sel].astype(np.float64) - lin[rows_sel]) ** 2).mean()))
        return cpu, mix_from_counters(cnt, n_s), rmse

    CPU_BUDGET_S = 12.0
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("GI_CPU_THREADS", "16"))))   # the box's CPU share for one GPU
    w, h, spp = args.width, args.height, args.spp
    res = run_workload(args.scene, w, h, spp, args.photons, args.steps, args.warmup)

    out = None
    if rank == 0:
        value = w * h * spp * args.steps / res["elapsed"] / 1e6
        out = {
            "metric": "Msamples/sec (primary+path rays) at 1080p", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"scenes/{args.scene} {w}x{h} {spp} spp, {args.photons} photon indices ({res['n_photons']} photons stored) + gather",
                       "frame": [w, h], "spp": spp, "photons_stored": res["n_photons"],
                       "sharding": f"{STRIPE_H}-row stripes round-robin over {world} GPU(s), RCCL gather to rank 0" if world > 1 else "single GPU",
                       "setup_s_untimed": round(res["setup_s"], 3), "mean_radiance": float(res["img"].mean())},
        }
        key = workload_key(args.scene, w, h, spp, args.photons)
        mix, cpu = stored_mix(key), None
        if world == 1 and not args.no_cpu and args.cpu_rows != 0:
            cpu, mix, rmse = cpu_leg(res, args.scene, w, h, spp, CPU_BUDGET_S, cores)
            out["config"]["rmse_vs_oracle_on_cpu_rows"] = rmse
            if args.write_mix:
                try:
                    with open(MIX_FILE) as f:
                        allmix = json.load(f)
                except (OSError, ValueError):
                    allmix = {}
                allmix[key] = mix
                with open(MIX_FILE, "w") as f:
                    json.dump(allmix, f, indent=1, sort_keys=True)
        out["roofline"] = roofline_of(args.scene, w, h, spp, args.photons, world, args.mode, res["rows"] * w * spp, res["stages"], res["kernel_ms"], mix)
        if mix is None:
            out["roofline"]["note"] = "no oracle-counted per-sample mix for this workload (run the N=1 CPU leg with --write-mix): algorithmic bytes unknown, achieved / frac omitted"
        if cpu is not None:
            out["cpu_baseline"] = cpu
    del res

    # ---- the closed-box configurations (BASELINE configs 2 and 4), N=1 only, inside the default run's time budget
    if world == 1 and not args.no_others and args.scene == "caustics":
        others = []
        for name, (ow, oh, ospp, oph, osteps, label) in {"cornell": (512, 512, 64, 0, 3, "config 2: scenes/cornell 512x512 64 spp, no photon map"),
                                                            "teapot": (1920, 1080, 256, 200000, 1, "config 4: scenes/cornell + glass teapot 1920x1080 256 spp, 200000 photon indices + gather")}.items():
            r = run_workload(name, ow, oh, ospp, oph, osteps, 1)
            o = {"workload": label, "value": ow * oh * ospp * osteps / r["elapsed"] / 1e6, "unit": "Msamples/s", "steps": osteps, "warmup": 1,
                 "ms_per_step": r["elapsed"] / osteps * 1e3, "photons_stored": r["n_photons"]}
            key = workload_key(name, ow, oh, ospp, oph)
            mix = stored_mix(key)
            if not args.no_cpu and args.cpu_rows != 0:
                c, mix, rmse = cpu_leg(r, name, ow, oh, ospp, 4.0, cores)
                o["cpu_baseline"], o["rmse_vs_oracle_on_cpu_rows"] = c, rmse
            o["roofline"] = roofline_of(name, ow, oh, ospp, oph, 1, args.mode, ow * oh * ospp, r["stages"], r["kernel_ms"], mix)
            others.append(o)
            del r
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
