#!/bin/bash
R=gpurun_out/r06; mkdir -p $R
timeout 1200 python3 -m pytest tests/test_gpu_overlap_e2e.py -m gpu -x -q -k "chunks_made_on_the_device or paf_bit_exact or config1_k10" 2>&1 | tail -3
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0"
for s in 1 5 5; do
    timeout 300 python3 bench.py --steps 2 --warmup 1 --slots $s $OFF --dense-job 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
d=j['overlap_default_k10_job']
print('slots $s | k13 %.2f M %.4f ms/round parity %s | k10 job %.4f s setup %.3f ms/round %.4f parity %s'%(j['value']/1e6,j['rounds_only']['ms_per_round'],j['parity']['paf_sha256_matches_oracle_fixture'],d['wall_s'],d['setup_s'],d['ms_per_round'],d['parity']),{k:round(v,3) for k,v in d['kernel_ms_per_round'].items()})" | tee -a $R/k10_step3.txt
done
TAG=step3 SLOTS="1 5" bash tools/r06/k10_trace.sh
