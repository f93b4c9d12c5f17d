"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of one bench.py frame into profiles/<name>_hbm_traffic.json.

usage: python tools/hbm_traffic.py OUT.json FETCH_counter_collection.csv WRITE_counter_collection.csv [scene w h spp photons]
"""
import csv, json, sys, collections
out, fcsv, wcsv = sys.argv[1:4]
scene, w, h, spp, photons = (sys.argv[4:9] + ["caustics", 1920, 1080, 256, 200000][len(sys.argv[4:9]):])
per = collections.defaultdict(lambda: {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "launches": 0})
for path, name in ((fcsv, "FETCH_SIZE"), (wcsv, "WRITE_SIZE")):
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")[:48]
            per[k][name + "_KB"] += float(r["Counter_Value"])
            if name == "FETCH_SIZE":
                per[k]["launches"] += 1
fetch = sum(v["FETCH_SIZE_KB"] for v in per.values()) * 1024
write = sum(v["WRITE_SIZE_KB"] for v in per.values()) * 1024
d = {"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python3 bench.py --steps 1 --warmup 0 --cpu-rows 0",
     "unit": "KB as reported by rocprofv3, summed over all launches of one frame",
     "workload_key": {"scene": scene, "frame": [int(w), int(h)], "spp": int(spp), "photons": int(photons), "n_gpus": 1, "mode": "wavefront"},
     "per_kernel": per, "frame_fetch_bytes_uncorrected": fetch, "frame_write_bytes": write,
     "frame_hbm_bytes_corrected": 2 * fetch + write,
     "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> doubled in the corrected figure (an upper bound for narrow accesses)"}
json.dump(d, open(out, "w"), indent=1)
print(out, "fetch %.1f GB (x2 = %.1f) write %.1f GB" % (fetch / 1e9, 2 * fetch / 1e9, write / 1e9))
