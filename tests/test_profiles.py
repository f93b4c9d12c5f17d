"""The committed rocprofv3 summaries under profiles/ belong to the bench lines committed next to them: the kernel durations rocprofv3 traced
per frame agree with the HIP-event times bench.py reported in the same run (tools/check_profile_agreement.py), for the benchmark configuration
and for BASELINE configs 2 and 4, and the HBM-traffic summaries carry the workload key bench.py looks them up by."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest(kind):
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(p))]
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % kind)), key=natural)
    return found[-1][: -len("_kernel_stats.csv")] if found else None


@pytest.mark.parametrize("kind", ["streaming", "config2_cornell", "config4_teapot"])
def test_kernel_trace_agrees_with_the_bench_line(kind):
    prefix = latest(kind)
    assert prefix, "no committed profile of kind " + kind
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_profile_agreement.py"), prefix], check=True, capture_output=True, text=True).stdout
    m = re.search(r"\(ratio ([0-9.]+)\)", out)
    assert m and 0.97 < float(m.group(1)) < 1.03, out
    m = re.search(r"dominant (\S+): rocprof ([0-9.]+) ms, bench.py ([0-9.]+) ms", out)
    assert m and abs(float(m.group(2)) / float(m.group(3)) - 1.0) < 0.03, out
    with open(prefix + "_hbm_traffic.json") as f:
        t = json.load(f)
    assert set(t["workload_key"]) == {"scene", "frame", "spp", "photons", "n_gpus", "mode"} and t["frame_hbm_bytes_corrected"] > 0
    with open(prefix + "_sq_pmc.json") as f:
        pmc = json.load(f)["per_kernel"]
    assert any(k.startswith("k_st_trace") and "SQ_WAIT_ANY" in v and "lanes_per_valu" in v for k, v in pmc.items())
