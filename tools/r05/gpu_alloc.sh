#!/bin/bash
# contexts park their blocks (device from 256 KiB, pinned) instead of freeing them; mapper defaults 4 threads x 2 730 reads in flight
R=gpurun_out/r05; mkdir -p $R
timeout 1200 python -m pytest tests -x -q -m gpu -k "map or release or cache or trim or ctx" > $R/alloc_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/alloc_tests.log
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 4 --map-cpu-baseline 0 > $R/map_marks.json 2> $R/map_marks.err
grep -E "^\[map (thread|end|setup)" $R/map_marks.err | tail -14
for cfg in "0 0" "3 1365" "0 0" "3 1365"; do
set -- $cfg
if [ $1 = 0 ]; then unset DP_MAP_THREADS DP_MAP_INFLIGHT; else export DP_MAP_THREADS=$1 DP_MAP_INFLIGHT=$2; fi
timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 6 --map-cpu-baseline 0 > $R/map_marks.json 2> /dev/null
python3 -c "
import json
d=json.loads([l for l in open('$R/map_marks.json') if l.startswith('{')][-1])['map_config3']; print('threads $1 inflight $2: %.0f reads/s' % d['value'], [round(x,4) for x in d['wall_s_runs']], {k:round(v,4) for k,v in d['breakdown_s'].items()}, 'kernel ms', round(d['map_kernel']['ms_total'],1), 'launches', d['map_kernel']['launches'], d['paf_sha256_matches_oracle_fixture'])"
done
unset DP_MAP_THREADS DP_MAP_INFLIGHT
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('overlap: value %.2fM ms/step %.1f rounds_only %.4f setup %.4f parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity']['paf_sha256_matches_oracle_fixture']))"
