#!/bin/bash
# the text pool's cap against the driver-style run (20 jobs): 256 MB (what the first version of the byte cap was) against 512 and 2048
R=gpurun_out/r05; mkdir -p $R
for mb in 256 512 2048 256 512; do
  DPH_TEXT_POOL_MB=$mb timeout 600 python bench.py --gpus 1 --steps 12 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 > $R/pool.json 2>/dev/null
  python3 - $mb <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/pool.json') if l.startswith('{')][-1])
pj=d['per_rank'][0]['per_job']
print('pool %4s MB: value %.2fM rounds_only %.4f | commit text %.1f ms (waits for formatter %.1f) formatter busy %.1f plan computes %.1f' % (sys.argv[1], d['value']/1e6, d['rounds_only']['ms_per_round'], pj['commit_text_us']/1e3, pj['commit_wait_for_formatter_us']/1e3, pj['formatter_busy_us']/1e3, pj['plan_compute_us']/1e3))
PY
done
for v in "DP_KX_BINS=0" "DP_KX_BINS=1" "DP_KX_BINS=0" "DP_KX_BINS=1"; do
  export $v
  timeout 600 python bench.py --gpus 1 --steps 12 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 > $R/pool.json 2>/dev/null
  unset DP_KX_BINS
  python3 - $v <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/pool.json') if l.startswith('{')][-1])
pj=d['per_rank'][0]['per_job']
print('%s (pool 512): value %.2fM rounds_only %.4f | commit text %.1f ms (waits %.1f)' % (sys.argv[1], d['value']/1e6, d['rounds_only']['ms_per_round'], pj['commit_text_us']/1e3, pj['commit_wait_for_formatter_us']/1e3))
PY
done
rm -f $R/pool.json
