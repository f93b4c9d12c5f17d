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
  "roofline"      for the dominant kernel (by live HIP-event time) and for every hot kernel: the resource that binds it and the fraction of that
                  resource's peak it reaches -- VALU issue (vector-ALU busy quad-cycles of the committed SQ counter profile / the live kernel
                  time against 1024 SIMDs at 2.4 GHz) or HBM (TCC bytes of the committed profile / the live kernel time against 8 TB/s); every
                  fraction is <= 1 by construction.  Next to it "reference_work": the SURVEY 8(d) bytes of the REFERENCE algorithm per second
                  (a work rate -- those tables live in LDS / L2, so it is not a bus load and carries no "frac"), and "executed_work": what the
                  streaming kernels executed (their own counters, one extra untimed frame) against the reference's visits;
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
    ap.add_argument("--no-executed", action="store_true", help="skip the extra counted frame (executed_work); the profiler passes use this so that every launch they see belongs to a timed frame")
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


N_SIMD = 256 * 4           # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # peak engine clock (MI355X_MICROARCH.md); the chip may hold less under load, which only lowers a busy fraction computed with it
VALU_PEAK_GQC = N_SIMD * CLOCK_HZ / 4 / 1e9   # VALU-busy quad-cycles per second the chip can deliver: one 64-wide VALU instruction = one quad-cycle of a SIMD
HBM_ACHIEVABLE_GBS = 6290.0                   # measured float4 copy (MI355X_MICROARCH.md, HBM section)
KERNEL_OF = {"trace": "k_st_trace", "shade": "k_st_shade", "shadow": "k_st_shadow", "gather": "k_st_gather", "finish": "k_st_finish", "other": "k_st_compact", "accum": "k_st_accum"}


def _natural(p):
    import re
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]   # r01_v10 after r01_v9, r03 after r02


def committed_profile(suffix, scene, w, h, spp, photons, world, mode):
    """The latest profiles/*<suffix> recorded for exactly this workload (rocprofv3 --pmc passes of `bench.py`, folded by tools/fold_profile.py);
    bench.py cannot sit under the profiler itself.  Returns (relative path, dict) or (None, None)."""
    import glob
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), key=_natural):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        wk = d.get("workload_key") or dict(d.get("workload", {}), n_gpus=1, mode="wavefront")
        if wk == {"scene": scene, "frame": [w, h], "spp": spp, "photons": photons, "n_gpus": world, "mode": mode}:
            best = (os.path.relpath(path, ROOT), d)
    return best


def _sum_family(per_kernel, family, field):
    """Sum a counter over the instances of one kernel family ('k_st_trace' matches 'k_st_trace<0, 1, false>')."""
    tot, hit = 0.0, False
    for name, d in per_kernel.items():
        base = name.split("<")[0].replace("void ", "")
        if (base == family or base.startswith(family + "_")) and field(d) is not None:     # k_st_gather_wave belongs to k_st_gather
            tot += field(d)
            hit = True
    return tot if hit else None


def roofline_of(scene_name, w, h, spp, photons, world, mode, local_samples, kernels, kernel_ms_avg, mix, executed):
    """kernels: live HIP-event ms per kernel family of one frame; mix: per-sample visits of the reference algorithm (oracle) or None;
    executed: per-sample counters of the streaming kernels or None."""
    pipeline_ms = sum(kernels.values()) if kernels and sum(kernels.values()) > 0 else kernel_ms_avg   # megakernel mode: one launch
    sq_path, sq = committed_profile("_sq_pmc.json", scene_name, w, h, spp, photons, world, mode)
    hb_path, hb = committed_profile("_hbm_traffic.json", scene_name, w, h, spp, photons, world, mode)
    per = {}
    for fam_key, fam in KERNEL_OF.items():
        ms = kernels.get(fam_key, 0.0)
        if ms <= 0:
            continue
        e = {"ms_per_frame": ms}
        if sq:
            act = _sum_family(sq["per_kernel"], fam, lambda d: d.get("SQ_ACTIVE_INST_VALU", {}).get("sum"))
            thr = _sum_family(sq["per_kernel"], fam, lambda d: d.get("SQ_THREAD_CYCLES_VALU", {}).get("sum"))
            wav = _sum_family(sq["per_kernel"], fam, lambda d: d.get("SQ_WAVE_CYCLES", {}).get("sum"))
            if act:
                gqc = act / (ms * 1e-3) / 1e9                      # VALU-busy quad-cycles per second, live time
                e["valu"] = {"achieved": gqc, "peak": VALU_PEAK_GQC, "unit": "G VALU-busy quad-cycles/s", "frac": gqc / VALU_PEAK_GQC,
                             "lanes_per_valu": (thr / act) if thr else None, "valu_active_share_of_wave_cycles": (act / wav) if wav else None}
        if hb:
            byt = _sum_family(hb.get("per_kernel", {}), fam, lambda d: 2.0 * d["FETCH_SIZE_KB"] * 1024 + d["WRITE_SIZE_KB"] * 1024)
            if byt:
                gbs = byt / (ms * 1e-3) / 1e9
                e["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frac_of_achievable": gbs / HBM_ACHIEVABLE_GBS,
                            "traffic": byt, "traffic_per_sample": byt / local_samples}
        # the binding resource: the one closest to its peak
        cands = [(k, e[k]["frac"]) for k in ("valu", "hbm") if k in e]
        if cands:
            b = max(cands, key=lambda kv: kv[1])
            e["bound"], e["frac"] = ("valu_issue" if b[0] == "valu" else "hbm"), b[1]
        per[fam] = e
    dom_key = max(kernels, key=lambda k: kernels.get(k, 0.0)) if kernels else None
    dom = per.get(KERNEL_OF.get(dom_key, ""), {}) if dom_key else {}
    res = dom.get("valu" if dom.get("bound") == "valu_issue" else "hbm", {})
    r = {"kernel": KERNEL_OF.get(dom_key, mode), "bound": dom.get("bound"), "achieved": res.get("achieved"), "peak": res.get("peak"), "unit": res.get("unit"),
         "frac": dom.get("frac"), "traffic": dom.get("hbm", {}).get("traffic"), "traffic_unit": "HBM bytes of this kernel per frame (all its launches): 2 x FETCH_SIZE + WRITE_SIZE",
         "lanes_per_valu": dom.get("valu", {}).get("lanes_per_valu"), "hbm_frac": dom.get("hbm", {}).get("frac"),
         "kernel_ms": dom.get("ms_per_frame"), "pipeline_ms": pipeline_ms, "frame_ms_event_to_event": kernel_ms_avg, "stage_ms": kernels,
         "per_kernel": per,
         "evidence": {"sq_counters": sq_path, "hbm_traffic": hb_path,
                      "how": "instruction / byte counts per frame from the committed rocprofv3 --pmc passes of this command (a property of binary + workload), "
                             "divided by the kernels' live HIP-event times of this run; tools/check_profile_agreement.py re-derives every fraction"}}
    if hb:
        tot = 2.0 * hb["frame_fetch_bytes_uncorrected"] + hb["frame_write_bytes"]
        r["pipeline_hbm"] = {"traffic": tot, "traffic_per_sample": tot / local_samples, "achieved": tot / (pipeline_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": tot / (pipeline_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if mix is not None:
        b = algorithmic_bytes_per_sample(mix, spp)
        r["reference_work"] = {"bytes_per_sample": b, "per_sample_mix": mix, "rate_GBps": b * local_samples / (pipeline_ms * 1e-3) / 1e9,
                               "note": "SURVEY 8(d): bytes the REFERENCE algorithm touches per sample (32 V + 36 T + 96 H + 36 P + 12/spp, counted by the oracle) x samples / pipeline time.  "
                                       "A work rate: the scene and photon tables are LDS / L2 resident and part of the reference's visits is never made (executed_work), so it is "
                                       "not bounded by the HBM peak and is not a roofline fraction."}
    if executed is not None:
        ex = dict(executed)
        ex["boxes"] = ex["trace_walks"] + ex["trace_child_boxes"] + ex["shadow_walks"] + ex["shadow_child_boxes"]
        ex["entity_tests"] = ex["trace_tris"] + ex["shadow_tris"]
        ex["entity_boxes"] = ex.get("trace_entity_boxes", 0.0) + ex.get("shadow_entity_boxes", 0.0)   # references sorted by their own box before Entity::intersect
        r["executed_work"] = {"per_sample": ex, "source": "gi_set_counters(ctx, 2): per-lane counters of k_st_trace / k_st_shadow / k_st_gather over one extra, untimed frame"}
        if mix is not None:
            r["executed_work"]["vs_reference"] = {"boxes": ex["boxes"] / mix["V"] if mix["V"] else None, "entity_tests": ex["entity_tests"] / mix["T"] if mix["T"] else None,
                                                  "photon_candidates": ex["gather_candidates"] / mix["P"] if mix["P"] else None, "shaded_hits": ex["shaded"] / mix["H"] if mix["H"] else None}
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
                stage_ms.append(rt.last_kernel_ms())       # HIP events around every launch, summed per kernel family

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
        executed = None
        if rank == 0 and world == 1 and args.mode == "wavefront" and not args.no_executed:      # what the kernels executed: one more frame, untimed, with their counters on
            try:
                rt.set_counters("stream")
                rt.run_device(p, fg.local.data_ptr(), f64=False)
                torch.cuda.synchronize()
                executed = {k: v / float(rows * w * spp) for k, v in rt.stream_counters().items()}
            except gi.GiError:
                executed = None                                       # a scene the counting instances do not cover
            finally:
                rt.set_counters(0)
        return {"executed": executed, "scene": scene, "rt": rt, "elapsed": elapsed, "kernel_ms": kernel_ms_avg, "stages": stages, "img": img, "rows": rows, "n_photons": n_photons, "photons_asked": photons, "setup_s": setup_s}

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
        cpu = {"value": n_s / cpu_s / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "sampled": True,
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
        out["roofline"] = roofline_of(args.scene, w, h, spp, args.photons, world, args.mode, res["rows"] * w * spp, res["stages"], res["kernel_ms"], mix, res["executed"])
        out["config"]["library"] = os.path.relpath(gi.LIB_PATH, ROOT)
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
            o["roofline"] = roofline_of(name, ow, oh, ospp, oph, 1, args.mode, ow * oh * ospp, r["stages"], r["kernel_ms"], mix, r["executed"])
            others.append(o)
            del r
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
