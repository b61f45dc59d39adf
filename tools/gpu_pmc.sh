#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  echo "$c rc=$?"
done
find gpurun_out/pmc_FETCH_SIZE -name "*.csv" | head
python3 - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv"%c)
    if not f: print(c,"no csv"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"]==c: acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(c,k,len(v),"mean",sum(v)/len(v))
PY
