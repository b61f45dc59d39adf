#!/usr/bin/env python3
"""Micro-benchmark of the scan kernels alone on config-2-shaped data (timing experiments; DP_SCAN_DEBUG toggles)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import gen_reads
import downpore_amd

N, L, k = int(os.environ.get("N", 100000)), 10000, 13
bases, off = gen_reads(2, N * L // 20, N, L, 0.0, False)
ctx = downpore_amd.Context(0)
ctx.upload_reads(bases, off)
rng = np.random.default_rng(1)
# seeds drawn from the reads themselves (like the real rounds): k-mers at random positions + their reverse complements
pos = rng.integers(0, len(bases) - k, 5000)
code = ((bases >> 1) ^ ((bases & 4) >> 2)) & 3
kms = []
for p in pos:
    v = 0
    for j in range(k):
        v = (v << 2) | int(code[p + j])
    rc = 0
    t = v
    for j in range(k):
        rc = (rc << 2) | ((t ^ 3) & 3)
        t >>= 2
    kms += [v, rc]
seeds = np.unique(np.array(kms, dtype=np.uint32))
ctx.round_begin(k, seeds)
items = np.zeros((N, 4), dtype=np.uint32)
items[:, 0] = np.arange(N)
items[:, 2] = L - k + 1
items[:, 3] = 15
for rep in range(4):
    r = ctx.scan(items)
print("count_ms %.4f write_ms %.4f total_ms %.4f hits %d  GB/s %.1f" % (r["count_kernel_ms"], r["write_kernel_ms"], r["kernel_ms"],
      int(r["n_seeds"].sum()), (N * L / 4 + 4 ** k / 8) / 1e9 / (r["count_kernel_ms"] / 1e3)))
