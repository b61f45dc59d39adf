#!/bin/bash
# per-phase wave timers of map_kernel (PROF build) over config 3
R=gpurun_out/r05; mkdir -p $R
DP_LIB_DIR=$PWD/downpore_amd/lib_prof DP_MAP_PROF=1 DP_MAP_THREADS=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 1 --map-cpu-baseline 0 > $R/map_prof.json 2> $R/map_prof.err
grep "map prof" $R/map_prof.err | head -3
python3 - <<'PY'
import re
rows=[]
for ln in open('gpurun_out/r05/map_prof.err'):
    m=re.search(r"\[map prof\] (\d+) window pairs on (\d+) waves, kernel ([\d.]+) us, slowest wave ([\d.]+) us, mean wave ([\d.]+) us \| us per wave: prefilter ([\d.]+) reduce target ([\d.]+) reduce query ([\d.]+) dynamicMatch ([\d.]+) out ([\d.]+) rest ([\d.]+) \| candidates chained per wave ([\d.]+)", ln)
    if m: rows.append([float(x) for x in m.groups()])
n=len(rows)
if n:
    mean=[sum(r[i] for r in rows)/n for i in range(12)]
    print("launches %d (one host thread): pairs %.0f waves %.0f | kernel %.1f us, slowest wave %.1f, mean wave %.1f | per wave: prefilter %.1f reduce target %.1f reduce query %.1f dynamicMatch %.1f out %.1f rest %.1f | chained candidates per wave %.2f" % tuple([n]+mean))
PY
rm -f $R/map_prof.err $R/map_prof.json
