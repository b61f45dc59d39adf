#!/bin/bash
# SQ counters of the consensus kernel after the contig + PAF rewrite (the pass of tools/gpu_profile_r05.sh's main_SQ run alone)
mkdir -p gpurun_out/r05; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r05
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
rm -rf $R/pmc_cons_SQ
timeout 900 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $R/pmc_cons_SQ -- python3 bench.py --steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --max-rounds 40 > $R/pmc_cons_SQ.json 2> $R/pmc_cons_SQ.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/r05/pmc_cons_SQ/*/*counter_collection.csv'))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'consensus_full' in n or 'match_anchor' in n:
        key = 'consensus_full_kernel<0>' if 'consensus_full_kernel<0>' in n else ('match_anchor' if 'match_anchor' in n else 'consensus_full_kernel<other>')
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k, 'dispatches', len(next(iter(v.values()))), {c: round(x) for c, x in m.items()}, 'conflict cycles per LDS instruction %.2f' % (m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, m.get('SQ_INSTS_LDS', 1))))
PY
find $R -name "*kernel_trace.csv" -path "*pmc_cons*" -delete; rm -rf $R/pmc_cons_SQ
