#!/bin/bash
R=gpurun_out/r06; mkdir -p $R
timeout 1200 python3 -m pytest tests/test_gpu_overlap_e2e.py tests/test_hand_known_answers.py -m gpu -x -q -k "chunks_made_on_the_device or chunk_worker or config1_k10 or paf_bit_exact" 2>&1 | tail -3
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for rep in 1 2; do
for s in 3 4 5 6 7; do
  timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('slots $s job %.4f s ms/round %.4f'%(j['job_breakdown_s']['whole_job'],j['rounds_only']['ms_per_round']),{k:round(v,3) for k,v in j['kernel_ms_per_round'].items()})" | tee -a $R/k10_slots_step4.txt
done
done
