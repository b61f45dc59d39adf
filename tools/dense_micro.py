#!/usr/bin/env python3
"""Per-round kernel times of the dense-seed regime (k=10) on BASELINE config 2's reads with ONE executor slot (no other
round competes for the GPU): the numbers to optimise the index query / index-mode write kernels against."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import gen_reads  # noqa: E402
from downpore_amd.overlap import OverlapPipeline, Reads  # noqa: E402
from downpore_amd import hip
hip.load_library().dp_set_kernel_timing(1)  # every round carries its timing events here

N = int(os.environ.get("READS", "100000"))
k = int(os.environ.get("K", "10"))
bases, off = gen_reads(2, N * 10000 // 20, N, 10000, 0.0, False)
reads = Reads(bases, off, min_len=1000)
pipe = OverlapPipeline(reads, k=k, slots=int(os.environ.get("SLOTS", "1")))
rounds = 0
acc = {}
for i in range(int(os.environ.get("ROUNDS", "8"))):
    if pipe.step() == 0:
        break
    st = pipe.stats()
    if i >= 2:
        rounds += 1
        for kk, v in st.items():
            acc[kk] = acc.get(kk, 0.0) + v
out = {kk: acc[kk] / max(1, rounds) for kk in ("k_count_ms", "k_write_ms", "k_query_ms", "k_chain_ms", "k_cons_ms", "query_bytes", "n_indexed", "n_matches", "n_hits", "idx_hits")}
out["query_GBs"] = out["query_bytes"] / 1e9 / (out["k_query_ms"] / 1e3) if out["k_query_ms"] else 0
print(json.dumps(out))
pipe.close()
