#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
C="--gpus 1 --steps 12 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for t in 2 3 4 2 3 4 6; do
  DPH_TEXT_THREADS=$t timeout 600 python bench.py $C > $R/tt.json 2>/dev/null
  python3 - $t <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/tt.json') if l.startswith('{')][-1])
pj=d['per_rank'][0]['per_job']
print('text threads %s: value %.2fM rounds_only %.4f | commit: waits %.1f text %.1f (for formatter %.1f) | formatter busy %.1f | slots waited for plans %.1f' % (sys.argv[1], d['value']/1e6, d['rounds_only']['ms_per_round'], pj['commit_thread_wait_us']/1e3, pj['commit_text_us']/1e3, pj['commit_wait_for_formatter_us']/1e3, pj['formatter_busy_us']/1e3, pj['slot_wait_for_plan_us']/1e3))
PY
done
rm -f $R/tt.json
