#!/bin/bash
# `map` config 3: the host's profile marks of a run (DPH_PROFILE=1), third run of a process
python3 - <<'PY' 2>&1 | tail -60
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.synth import gen_genome, gen_reads
from downpore_amd.mapping import map_reads
from downpore_amd.overlap import Reads
g=json.load(open("tests/golden_full/config3_map.json"))["generator"]
genome = np.frombuffer(gen_genome(g["seed"], g["genome"]), dtype=np.uint8)
goff = np.array([0, g["genome"]], dtype=np.int64)
bases, off = gen_reads(g["seed"], g["genome"], g["reads"], g["read_len"], g["error"], False)
ref = Reads(genome, goff, min_len=0, himem=False); reads = Reads(bases, off, min_len=500, himem=False)
for i in range(4):
    if i == 3: os.environ["DPH_PROFILE"] = "1"
    t0=time.perf_counter(); paf, err, st = map_reads(ref, reads, circular=True, k=11); dt=time.perf_counter()-t0
    print("run %d: %.1f ms"%(i, dt*1e3), {k: round(v,4) for k,v in st.items() if k.startswith("t_")})
PY
