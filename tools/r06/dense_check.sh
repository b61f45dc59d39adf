#!/bin/bash
R=gpurun_out/r06; mkdir -p $R
timeout 1200 python3 -m pytest tests/test_gpu_overlap_e2e.py -m gpu -x -q -k "counting_step_variants or kmer_index_mode or identical_and_repetitive or paf_bit_exact or slots_with_ignores" 2>&1 | tail -5
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0"
for v in 0 1 0 1; do
  for s in 1 5; do
    [ $v = 0 ] && export DP_KX_DENSE=0 || unset DP_KX_DENSE
    timeout 300 python3 bench.py --steps 1 --warmup 1 --slots $s $OFF --dense-job 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
d=j['overlap_default_k10_job']
print('dense $v slots $s | k13 %.2f M | k10 job %.4f s ms/round %.4f parity %s'%(j['value']/1e6,d['wall_s'],d['ms_per_round'],d['parity']),{k:round(v,3) for k,v in d['kernel_ms_per_round'].items()})" | tee -a $R/dense_ab.txt
  done
done
