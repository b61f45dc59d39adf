#!/usr/bin/env python3
"""Turns the module+offset call sites of a DPH_SAMPLE_PROF=1 report (stderr lines starting with [sample]) into file:line with
addr2line on the in-tree libraries (the same files the GPU box ran).  usage: sample_resolve.py gpurun_out/samp.err"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"libdownpore_hip.so": os.path.join(ROOT, "downpore_amd", "lib", "libdownpore_hip.so"),
        "libdownpore_host.so": os.path.join(ROOT, "downpore_amd", "lib", "libdownpore_host.so")}
for ln in open(sys.argv[1]):
    if not ln.startswith("[sample]"):
        continue
    m = re.search(r"(libdownpore_\w+\.so)\+0x([0-9a-f]+)( own)?", ln)
    if not m:
        print(ln.rstrip())
        continue
    off = int(m.group(2), 16) - (0 if m.group(3) else 1)  # a return address: the call is the instruction before it
    out = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", libs[m.group(1)], hex(off)], capture_output=True, text=True).stdout.split("\n")
    where = " <- ".join("%s %s" % (out[i][:60], os.path.basename(out[i + 1])) for i in range(0, len(out) - 1, 2))
    print(ln.rstrip().replace(m.group(0), m.group(0) + "  " + where))
