#!/bin/bash
# A/B on one box: the previous commit's build (_ab/prev, a git worktree) against the working tree, alternating
run() { # dir label
  ( cd $1 && timeout 600 python3 bench.py --steps ${STEPS:-3} --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots ${SLOTS:-8} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 slots ${SLOTS:-8}: %.2f M/s, job %.3f s, setup %.3f s, %.3f ms/round, parity %s' % (d['value']/1e6, d['job_breakdown_s']['whole_job'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['rounds_only']['ms_per_round'], d['parity']['paf_sha256_matches_oracle_fixture']))" )
}
for i in ${REPS-1 2}; do run _ab/prev prev; run . new; done
