#!/bin/bash
# the driver-style line alone (10 steps), with what says how fast this box's host is
mkdir -p gpurun_out/r04
python3 bench.py --gpus 1 --steps 10 --warmup 3 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); pj=d['per_rank'][0]['per_job']
print('value %.2fM ms/round %.4f | oracle 2 rounds: %s | slots waited for plans %.1f ms, plan computes %.1f ms per job | lanes default' % (d['value']/1e6, d['rounds_only']['ms_per_round'], d['cpu_baseline']['sample'], pj['slot_wait_for_plan_us']/1e3, pj['plan_compute_us']/1e3))"
