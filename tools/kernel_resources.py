#!/usr/bin/env python3
"""Tuning aid: registers, scratch (spills) and LDS of every kernel in a built library, from the code object's metadata.
usage: tools/kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gi_raytracer_amd", "libgi_raytracer_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else "k_st_|k_render|k_emit"
with tempfile.TemporaryDirectory() as td:
    co = os.path.join(td, "co")
    # the fat binary sits in section .hip_fatbin: unbundle the gfx950 code object
    sec = os.path.join(td, "fat")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, sec], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={sec}", f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    filt = subprocess.run(["c++filt"], input=notes, capture_output=True, text=True).stdout
blocks = re.split(r"\n\s+- \.agpr_count:", filt)
rows = []
for b in blocks[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
    name = (re.search(r"\.name:\s+(.*)", b) or [None, "?"])[1].strip("'\" ")
    name = re.sub(r"\(.*", "", name)
    if re.search(flt, name):
        rows.append((name, b.split()[0], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
print("%-46s %5s %5s %5s %8s %8s %6s" % ("kernel", "agpr", "vgpr", "sgpr", "scratch", "lds", "spills"))
for r in sorted(rows):
    print("%-46s %5s %5s %5s %8s %8s %6s" % r)
