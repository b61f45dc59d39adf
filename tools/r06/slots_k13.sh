#!/bin/bash
# k = 13 config-2 jobs with 4 .. 7 executor slots, alternating, three times (after the flags' pinned shadow)
for rep in 1 2 3; do for s in 5 6 4 7; do
timeout 300 python3 bench.py --steps 3 --warmup 1 --slots $s --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
pj=(j.get('per_rank') or [{}])[0].get('per_job',{})
print('slots $s: %.2f M, job %.4f s, rounds %.4f ms, slots waited for plans %.1f ms, lanes %s'%(j['value']/1e6, j['job_breakdown_s']['whole_job'], j['rounds_only']['ms_per_round'], pj.get('slot_wait_for_plan_us',0)/1e3, (j.get('per_rank') or [{}])[0].get('planner_lanes_at_the_end')))"
done; done
