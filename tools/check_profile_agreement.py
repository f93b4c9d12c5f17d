"""rocprofv3 --kernel-trace --stats summary vs the HIP-event times bench.py reports for the same run (profiles/<tag>_kernel_stats.csv and
<tag>_bench_under_rocprof.json): per-frame sum of the pipeline's kernel durations, and the dominant kernel.  usage: ... profiles/r01_v9_streaming"""
import csv, json, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(tag + "_kernel_stats.csv")))
d = json.loads(open(tag + "_bench_under_rocprof.json").readline())
frames = d["steps"] + d["warmup"]
tot, parts = 0.0, {}
for r in rows:
    n = r["Name"]
    if n.startswith("k_emit") or "at::native" in n:      # photon emission (setup) and torch's own fills are not part of a frame
        continue
    t = float(r["TotalDurationNs"]) / 1e6 / frames
    tot += t
    key = n.split("(")[0].replace("void ", "")[:32]
    parts[key] = parts.get(key, 0.0) + t
print("rocprof kernel durations per frame: %.1f ms; bench.py HIP events: %.1f ms (ratio %.4f)" % (tot, d["roofline"]["kernel_ms"], tot / d["roofline"]["kernel_ms"]))
stages = d["roofline"]["stage_ms"]
name = max(stages, key=stages.get)                        # the dominant stage (trace / shade / gather: one kernel each)
k = [v for n, v in parts.items() if n.startswith("k_st_" + name)]
print("dominant k_st_%s: rocprof %.1f ms, bench.py %.1f ms" % (name, sum(k), stages[name]))
