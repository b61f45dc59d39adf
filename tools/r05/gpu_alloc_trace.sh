#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
DP_ALLOC_TRACE=1 timeout 300 python3 - 2> $R/alloc_trace.txt <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
from tools.synth import gen_genome, gen_reads
from downpore_amd.mapping import map_reads
from downpore_amd.overlap import Reads
genome = np.frombuffer(gen_genome(3, 4600000), dtype=np.uint8); goff = np.array([0, 4600000], dtype=np.int64)
bases, off = gen_reads(3, 4600000, 50000, 8000, 0.1, False)
ref = Reads(genome, goff, min_len=0, himem=False); reads = Reads(bases, off, min_len=500, himem=False)
for i in range(7):
    t0 = time.perf_counter(); paf, err, st = map_reads(ref, reads, circular=True, k=11); dt = time.perf_counter() - t0
    sys.stderr.write("=== run %d: %.3f s\n" % (i, dt)); sys.stderr.flush()
PY
python3 - <<'PY'
import re
run = 0; acc = {}
for l in open('gpurun_out/r05/alloc_trace.txt'):
    m = re.match(r"=== run (\d+): ([\d.]+) s", l)
    if m:
        print('run', m.group(1), m.group(2), 's | driver allocations >= 1 MB: %d blocks, %.1f MB' % (acc.get('n', 0), acc.get('b', 0) / 1e6), '| sizes (MB):', sorted(acc.get('sz', []), reverse=True)[:12], '| parked at last alloc: %.0f MB' % (acc.get('parked', 0) / 1e6)); acc = {}; continue
    m = re.search(r"device block (\d+) bytes .*parked now: (\d+) bytes", l)
    if m:
        b = int(m.group(1)); acc['n'] = acc.get('n', 0) + 1; acc['b'] = acc.get('b', 0) + b; acc.setdefault('sz', []).append(round(b / 1e6, 1)); acc['parked'] = int(m.group(2))
PY
