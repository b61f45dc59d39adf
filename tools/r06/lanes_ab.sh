#!/bin/bash
# planner lanes fixed at 3 / fixed at 4 / adaptive (default): 12 timed k = 13 jobs each, alternating twice; job times' spread
for rep in 1 2; do for lanes in adaptive 3 4; do
if [ $lanes = adaptive ]; then unset DPH_PLAN_LANES; else export DPH_PLAN_LANES=$lanes; fi
timeout 300 python3 bench.py --steps 12 --warmup 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
pj=sorted(j['job_breakdown_s']['per_job']); pr=(j.get('per_rank') or [{}])[0]
print('lanes $lanes: %.2f M, jobs ms min %.1f med %.1f max %.1f, slots waited for plans %.1f ms per job, lanes at the end %s'%(j['value']/1e6, 1e3*pj[0], 1e3*pj[len(pj)//2], 1e3*pj[-1], pr['per_job']['slot_wait_for_plan_us']/1e3, pr.get('planner_lanes_at_the_end')))"
done; done
