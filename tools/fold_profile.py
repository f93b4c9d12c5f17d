#!/usr/bin/env python3
"""Fold one tools/profile_r03.sh output directory (gpurun_out/prof_TAG) into the committed summaries under profiles/:

  <prefix>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
  <prefix>_bench.json                the bench line of the same command without the profiler
  <prefix>_bench_under_rocprof.json  ... and under it
  <prefix>_sq_pmc.json               SQ / TA / TCP counters per kernel (separate --pmc passes, one frame each)
  <prefix>_hbm_traffic.json          TCC FETCH_SIZE / WRITE_SIZE per kernel (tools/hbm_traffic.py)

usage: tools/fold_profile.py gpurun_out/prof_v6 profiles/r02_v6_streaming [scene w h spp photons]
"""
import collections, csv, json, os, shutil, subprocess, sys

src, prefix = sys.argv[1:3]
wl = sys.argv[3:8] or ["caustics", "1920", "1080", "256", "200000"]
samples = int(wl[1]) * int(wl[2]) * int(wl[3])
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), prefix + "_kernel_stats.csv")
for a, b in (("bench.json", "_bench.json"), ("bench_under_rocprof.json", "_bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, a), prefix + b)


def kernel_name(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:48]


per = collections.defaultdict(dict)
sets = {}
for name in ("sq_a", "sq_b", "sq_c", "ta", "tcp"):
    path = os.path.join(src, name, name + "_counter_collection.csv")
    if not os.path.exists(path):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[kernel_name(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    sets[name] = sorted({c for d in acc.values() for c in d})
    for k, d in acc.items():
        if not k.startswith("k_"):
            continue
        wc = d.get("SQ_WAVE_CYCLES")
        for c, v in d.items():
            if c == "SQ_WAVE_CYCLES" and c in per[k]:
                continue
            e = {"sum": v, "per_64_samples": v / (samples / 64.0)}
            if wc and c != "SQ_WAVE_CYCLES":
                e["per_wave_cycle"] = round(v / wc, 4)
            per[k][c] = e
for k, d in per.items():
    if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"]["sum"] > 0:
        d["lanes_per_valu"] = d["SQ_THREAD_CYCLES_VALU"]["sum"] / d["SQ_ACTIVE_INST_VALU"]["sum"]
json.dump({"command": "tools/profile_r03.sh: rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-others (one pass per set)",
           "workload": {"scene": wl[0], "frame": [int(wl[1]), int(wl[2])], "spp": int(wl[3]), "photons": int(wl[4])},
           "sets": sets, "per_kernel": per,
           "note": "sums over all launches of one frame; SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); per_wave_cycle = counter / SQ_WAVE_CYCLES of the same pass; "
                   "per_64_samples = counter / (samples of the frame / 64); lanes_per_valu = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)"},
          open(prefix + "_sq_pmc.json", "w"), indent=1)
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hbm_traffic.py"), prefix + "_hbm_traffic.json",
                       os.path.join(src, "fetch", "fetch_counter_collection.csv"), os.path.join(src, "write", "write_counter_collection.csv")] + wl)
for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", {}).get("sum", 0)):
    g = lambda c: d.get(c, {}).get("per_wave_cycle")
    print("%-28s wait %s valu %s lanes %.1f valu/64samples %.0f" % (k, g("SQ_WAIT_ANY"), g("SQ_ACTIVE_INST_VALU"), d.get("lanes_per_valu", 0.0), d.get("SQ_INSTS_VALU", {}).get("per_64_samples", 0.0)))
