"""Re-derive the roofline object of saved bench lines (profiles/<tag>_bench.json, <tag>_bench_under_rocprof.json) against the counter summaries
committed under the SAME tag.  A profile run computes its lines while only the previous tag's counters exist; once the new summaries are
copied into profiles/ this makes each line cite (and agree with) the counters collected in its own run.  Times in the line are untouched.
usage: python tools/rebase_roofline.py profiles/r03_v3_streaming"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1]
for suffix in ("_bench.json", "_bench_under_rocprof.json"):
    path = tag + suffix
    d = json.loads(open(path).readline())
    r = d["roofline"]
    k = json.load(open(tag + "_sq_pmc.json"))
    k = k.get("workload_key") or dict(k["workload"], n_gpus=1, mode="wavefront")       # the workload the counters were collected on = the line's
    (w, h), world = k["frame"], k["n_gpus"]
    assert d["config"]["frame"] == [w, h] and d["config"]["spp"] == k["spp"] and d["n_gpus"] == world
    mix = r.get("reference_work", {}).get("per_sample_mix")
    ex = r.get("executed_work", {}).get("per_sample")
    new = bench.roofline_of(k["scene"], w, h, k["spp"], k["photons"], world, k["mode"], w * h * k["spp"] / world, r["stage_ms"], r["frame_ms_event_to_event"], mix, ex)
    assert os.path.basename(new["evidence"]["sq_counters"]).startswith(os.path.basename(tag)), new["evidence"]
    d["roofline"] = new
    with open(path, "w") as f:
        f.write(json.dumps(d) + "\n")
    print(path, "->", new["evidence"]["sq_counters"], "frac %.3f" % new["frac"])
