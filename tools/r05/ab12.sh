#!/bin/bash
# alternating 12-job runs of the default workload, one environment setting per argument (e.g. "DP_KX_FUSE=0" "DP_KX_FUSE=1"), REPS passes
R=gpurun_out/r05; mkdir -p $R
C="--gpus 1 --steps 12 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for rep in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
  env $v timeout 600 python bench.py $C > $R/ab12.json 2>/dev/null
  python3 - "$v" <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/ab12.json') if l.startswith('{')][-1])
pj=d['per_rank'][0]['per_job']
print('%-22s value %.2fM rounds_only %.4f | slots waited for plans %.1f ms | parity %s' % (sys.argv[1], d['value']/1e6, d['rounds_only']['ms_per_round'], pj['slot_wait_for_plan_us']/1e3, d['parity']['paf_sha256_matches_oracle_fixture']))
PY
done; done
rm -f $R/ab12.json
