"""rocprofv3 --kernel-trace --stats summary vs the HIP-event times bench.py reports for the same run (profiles/<tag>_kernel_stats.csv and
<tag>_bench_under_rocprof.json): per-frame sum of the pipeline's kernel durations, and the dominant kernel.  For bench lines of round 3 on
(roofline.per_kernel) it also re-derives every roofline fraction of the line from <tag>_sq_pmc.json / <tag>_hbm_traffic.json and the line's own
kernel times, and fails when one of them is above 1 or differs from what the line says.  usage: ... profiles/r03_v1_streaming [bench line .json]"""
import csv, json, os, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(tag + "_kernel_stats.csv")))
d = json.loads(open(sys.argv[2] if len(sys.argv) > 2 else tag + "_bench_under_rocprof.json").readline())
frames = d["steps"] + d["warmup"]
tot, parts = 0.0, {}
for r in rows:
    n = r["Name"]
    if n.startswith(("k_emit", "k_pb_", "k_pleaf", "k_pcand", "k_pjump", "k_pdescent")) or "at::native" in n:      # photon emission / map build (setup) and torch's own fills are not part of a frame
        continue
    t = float(r["TotalDurationNs"]) / 1e6 / frames
    tot += t
    key = n.split("(")[0].replace("void ", "")[:32]
    parts[key] = parts.get(key, 0.0) + t
roof = d["roofline"]
pipeline_ms = roof.get("pipeline_ms", roof["kernel_ms"])
print("rocprof kernel durations per frame: %.1f ms; bench.py HIP events: %.1f ms (ratio %.4f)" % (tot, pipeline_ms, tot / pipeline_ms))
stages = roof["stage_ms"]
name = max(stages, key=stages.get)                        # the dominant stage (trace / shade / shadow / gather: one kernel each)
k = [v for n, v in parts.items() if n.split("<")[0].split("(")[0] in ("k_st_" + name, "k_st_" + name + "_wave") or (name == "shade" and "shadow" not in stages and n.startswith("k_st_shadow"))]
print("dominant k_st_%s: rocprof %.1f ms, bench.py %.1f ms" % (name, sum(k), stages[name]))
if "per_kernel" in roof:
    # the counter summaries the line says it used (relative to the repository root), else this tag's
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ev = roof.get("evidence", {})
    sq_path = os.path.join(root, ev["sq_counters"]) if ev.get("sq_counters") and os.path.exists(os.path.join(root, ev["sq_counters"])) else tag + "_sq_pmc.json"
    hb_path = os.path.join(root, ev["hbm_traffic"]) if ev.get("hbm_traffic") and os.path.exists(os.path.join(root, ev["hbm_traffic"])) else tag + "_hbm_traffic.json"
    print("counters: %s, %s" % (os.path.relpath(sq_path, root), os.path.relpath(hb_path, root)))
    sq = json.load(open(sq_path))["per_kernel"]
    hb = json.load(open(hb_path))["per_kernel"]
    base = lambda n: n.split("<")[0].replace("void ", "")
    fam = lambda table, f, get: sum(get(v) for n, v in table.items() if (base(n) == f or base(n).startswith(f + "_")) and get(v) is not None)
    bad = 0
    for f, e in roof["per_kernel"].items():
        ms = e["ms_per_frame"]
        valu = fam(sq, f, lambda v: v.get("SQ_ACTIVE_INST_VALU", {}).get("sum")) / (ms * 1e-3) / 1e9 / (1024 * 2.4e9 / 4 / 1e9)
        hbm = fam(hb, f, lambda v: 2.0 * v["FETCH_SIZE_KB"] * 1024 + v["WRITE_SIZE_KB"] * 1024) / (ms * 1e-3) / 1e9 / 8000.0
        said_v, said_h = e.get("valu", {}).get("frac"), e.get("hbm", {}).get("frac")
        ok = all(x is None or (x <= 1.0 and abs(x - y) <= 1e-6 + 1e-6 * y) for x, y in ((said_v, valu), (said_h, hbm))) and (e.get("frac") is None or e["frac"] <= 1.0)
        bad += 0 if ok else 1
        print("%-14s %7.2f ms  bound %-10s frac %.3f | valu %.3f (line %s) hbm %.3f (line %s)%s" % (f, ms, e.get("bound"), e.get("frac") or 0.0, valu, "%.3f" % said_v if said_v is not None else "-",
                                                                                                  hbm, "%.3f" % said_h if said_h is not None else "-", "" if ok else "   <-- MISMATCH"))
    print("roofline fractions re-derived from the committed counters: %s" % ("all agree, all <= 1" if bad == 0 else "%d DISAGREE" % bad))
    if bad:
        sys.exit(1)
