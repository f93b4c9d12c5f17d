"""Sum rocprofv3 PMC counters per kernel from a counter_collection.csv (profiling aid)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k, {c: (f"{v:.3g}", f"{v / wc:.2f}") for c, v in d.items()})
