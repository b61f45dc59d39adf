#!/bin/bash
# `map` config 3: mapper threads x reads in flight, median of 10 runs each (after 2 warm runs), one process per setting, alternating
for rep in 1 2 3; do for cfg in "6 2730" "8 2730" "8 2048"; do set -- $cfg
DP_MAP_THREADS=$1 DP_MAP_INFLIGHT=$2 python3 - <<'PY'
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
ts=[]
for i in range(12):
    t0=time.perf_counter(); paf, err, st = map_reads(ref, reads, circular=True, k=11); dt=time.perf_counter()-t0
    if i >= 2: ts.append(dt)
ts.sort()
print("threads %s inflight %s: median %.1f ms = %.0f k reads/s, best %.1f, worst %.1f" % (os.environ["DP_MAP_THREADS"], os.environ["DP_MAP_INFLIGHT"], 1e3*ts[len(ts)//2], g["reads"]/ts[len(ts)//2]/1e3, 1e3*ts[0], 1e3*ts[-1]), flush=True)
PY
done; done 2>&1 | grep threads
