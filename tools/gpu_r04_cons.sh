#!/bin/bash
# consensus kernel: phase timers of a 40-round one-slot job (stderr digest) + the parity flag of the bench line
mkdir -p gpurun_out/r04
DP_CONS_DEBUG=1 python3 bench.py --steps 1 --warmup 0 --max-rounds 40 --slots 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 2> gpurun_out/r04/cons_debug.txt > gpurun_out/r04/cons_debug.json
python3 - <<'PY'
import re, statistics as st
L = open("gpurun_out/r04/cons_debug.txt").read().splitlines()
def med(pat, n):
    rows = [list(map(float, re.findall(pat, l)[0])) for l in L if re.findall(pat, l)]
    rows = rows[5:]
    return [st.median(r[i] for r in rows) for i in range(n)] if rows else None
print("mean/max per phase (median over rounds): gather %s/%s trim %s/%s shared+reduce %s/%s align %s/%s contig+paf %s/%s total %s/%s" % tuple(
    med(r"gather\+query ([\d.]+)/([\d.]+) trim ([\d.]+)/([\d.]+) shared\+reduce ([\d.]+)/([\d.]+) align ([\d.]+)/([\d.]+) contig\+paf ([\d.]+)/([\d.]+) \| group total ([\d.]+)/([\d.]+)", 12)))
print("general steps per window: before scan %s scan %s selection %s update %s us | runs %s" % tuple(
    med(r"before the scan ([\d.]+) us, scan ([\d.]+), selection ([\d.]+), update ([\d.]+) \| scan runs ([\d.]+)", 5)))
PY
DP_CONS_DEBUG= python3 bench.py --steps 2 --warmup 1 --slots 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('one slot: ms/round', j['rounds_only']['ms_per_round'], 'parity', j['parity']['paf_sha256_matches_oracle_fixture'], 'k_cons_ms per job', j['per_rank'][0]['kernel_ms_per_job']['k_cons_ms'])"
