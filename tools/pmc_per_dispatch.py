"""Per-dispatch PMC rows of one kernel from a rocprofv3 counter_collection.csv (profiling aid)."""
import csv, sys, collections
path, kern = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if kern not in r["Kernel_Name"]:
        continue
    rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in list(rows.items())[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]:
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: ("%.3g" % v, "%.2f" % (v / wc)) for c, v in d.items()})
