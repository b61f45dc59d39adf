#!/bin/bash
# stream waits: the runtime's spinning wait against the 20 us poll, slots, CPU used
for cfg in "1 8" "1 6" "1 10" "1 12" "0 8"; do set -- $cfg
  DP_SPIN_SYNC=$1 timeout 300 python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('spin $1 slots $2: %.2f M/s, job %.3f s, %.3f ms/round, cpu %.1f s over %.2f s wall (%.1f cores), throttled %.2f s, parity %s' % (d['value']/1e6, d['job_breakdown_s']['whole_job'], d['rounds_only']['ms_per_round'], d['host_cpu']['cpu_s'], d['host_cpu']['wall_s'], d['host_cpu']['cpu_s']/d['host_cpu']['wall_s'], d['host_cpu']['throttled_s'], d['parity']['paf_sha256_matches_oracle_fixture']))"
done
