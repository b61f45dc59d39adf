#!/bin/bash
# config 5's share of the reference (375 Mb): the mapper's set-up with the single-seed walk on the host / candidates from the device
mkdir -p gpurun_out/r04
for v in 1 0; do
DP_MAP_SEEDS_HOST=$v DPH_PROFILE=1 timeout 900 python3 - > gpurun_out/r04/c5_setup_host$v.log 2>&1 <<'PY'
import numpy as np, time
from tests import oracle_lib as O
from downpore_amd.mapping import map_reads
from downpore_amd.overlap import Reads
G, N, L, e, seed = 375000000, 200, 15000, 0.1, 5
genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
goff = np.array([0, G], dtype=np.int64)
bases, off = O.gen_reads(seed, G, N, L, e, False)
t0 = time.time()
got, gerr, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=500, himem=False), circular=True, k=13)
print("map_reads %.2f s, chunks %d seeds %d, paf lines %d" % (time.time() - t0, st["n_chunks"], st["n_seeds"], got.count("\n")))
import hashlib; print("paf sha", hashlib.sha256(got.encode()).hexdigest()[:16])
PY
echo "host_walk=$v"; grep -E "map setup|map_reads|paf sha" gpurun_out/r04/c5_setup_host$v.log | tail -14
done
