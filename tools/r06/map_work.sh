#!/bin/bash
# `map` config 3: tests of the mapper, then the host's profile marks of a run and the spread of 12 runs of one process
R=gpurun_out/r06; mkdir -p $R; TAG=${TAG:-x}
if [ -z "$NOTESTS" ]; then timeout 1500 python3 -m pytest tests/test_gpu_map.py tests/test_host_coroutines.py -x -q > $R/map_tests_$TAG.log 2>&1; echo "tests rc $?"; tail -3 $R/map_tests_$TAG.log; fi
python3 - > $R/map_marks_$TAG.txt 2>&1 <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.synth import gen_genome, gen_reads
from downpore_amd.mapping import map_reads
from downpore_amd.overlap import Reads
import hashlib
fx=json.load(open("tests/golden_full/config3_map.json")); g=fx["generator"]
genome = np.frombuffer(gen_genome(g["seed"], g["genome"]), dtype=np.uint8)
goff = np.array([0, g["genome"]], dtype=np.int64)
bases, off = gen_reads(g["seed"], g["genome"], g["reads"], g["read_len"], g["error"], False)
ref = Reads(genome, goff, min_len=0, himem=False); reads = Reads(bases, off, min_len=500, himem=False)
ts=[]
for i in range(14):
    if i == 3: os.environ["DPH_PROFILE"] = "1"
    else: os.environ.pop("DPH_PROFILE", None)
    t0=time.perf_counter(); paf, err, st = map_reads(ref, reads, circular=True, k=11); dt=time.perf_counter()-t0
    if i >= 2 and i != 3: ts.append(dt)
    if i < 4: print("run %d: %.1f ms"%(i, dt*1e3), {k: round(v,4) for k,v in st.items() if k.startswith("t_")}, flush=True)
    if i == 0: print("sha ok", hashlib.sha256(paf if isinstance(paf,bytes) else paf.encode()).hexdigest() == fx.get("paf_sha256"), flush=True)
ts.sort()
print("runs (ms):", [round(1e3*t,1) for t in ts])
print("median %.1f ms = %.0f k reads/s; best %.1f ms = %.0f k reads/s" % (1e3*ts[len(ts)//2], g["reads"]/ts[len(ts)//2]/1e3, 1e3*ts[0], g["reads"]/ts[0]/1e3))
PY
tail -40 $R/map_marks_$TAG.txt
