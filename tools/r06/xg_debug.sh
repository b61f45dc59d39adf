#!/bin/bash
# dense walk with read ranges per XCD: PAF hashes of the first rounds, the k = 13 case of the variants test
for dbg in "" kx_bins; do for slots in 1 5; do for xg in 1 3 8; do
DP_SCAN_INDEX=1 DP_KX_DENSE=1 DP_DEBUG=$dbg DP_TUNE=kx_lps=64,kx_xgroups=$xg SLOTS=$slots python3 - 2>/tmp/err.txt <<'PY'
import os, sys, hashlib
sys.path.insert(0, os.getcwd())
from tests import oracle_lib as O
from downpore_amd.overlap import OverlapPipeline, Reads
bases, off = O.gen_reads(114, 1200000, 2000, 12000, 0.002, True)
reads = Reads(bases, off, min_len=1000, himem=True)
pipe = OverlapPipeline(reads, k=13, himem=True, slots=int(os.environ["SLOTS"]))
h = hashlib.sha256(); n = 0; rounds = 0
while rounds < 4:
    c = pipe.step()
    if c == 0: break
    rounds += c
    t = pipe.round_paf(); n += t.count("\n"); h.update(t.encode())
print("debug", os.environ.get("DP_DEBUG") or "-", "slots", os.environ["SLOTS"], os.environ["DP_TUNE"], "rounds", rounds, "lines", n, h.hexdigest()[:16])
pipe.close()
PY
grep "kx bins" /tmp/err.txt | head -2
done; done; done
