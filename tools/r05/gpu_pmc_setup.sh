#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the set-up kernels on the round's last code state (one --pmc pass each, --kernel-trace only), 3 builds
mkdir -p gpurun_out/r05; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r05
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/pmc_setup_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/pmc_setup_$c -- python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --max-rounds 4 > /dev/null 2> $R/pmc_setup_$c.err; echo "$c rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob('gpurun_out/r05/pmc_setup_%s/*/*counter_collection.csv' % c))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n.startswith('kb_') or n.startswith('values_'):
            acc[n].append(float(r['Counter_Value']))
    for n, v in acc.items():
        out.setdefault(n, {})[c + '_KiB_mean'] = sum(v) / len(v)
        out[n]['dispatches'] = len(v)
for n, d in sorted(out.items()):
    f, w = d.get('FETCH_SIZE_KiB_mean', 0), d.get('WRITE_SIZE_KiB_mean', 0)
    d['hbm_MB_per_launch_fetch_x2_plus_write'] = (2 * f + w) * 1024 / 1e6
    print(n, d['dispatches'], 'fetch %.1f MB (x2: %.1f) write %.1f MB' % (f * 1024 / 1e6, 2 * f * 1024 / 1e6, w * 1024 / 1e6))
json.dump(out, open('gpurun_out/r05/pmc_setup_kernels.json', 'w'), indent=1)
PY
find $R -name "*kernel_trace.csv" -path "*pmc_setup*" -delete; rm -rf $R/pmc_setup_FETCH_SIZE $R/pmc_setup_WRITE_SIZE
