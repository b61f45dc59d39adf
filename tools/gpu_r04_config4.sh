#!/bin/bash
# config-4 scale on ONE GPU (1 M reads x 10 kb resident, 80 GB k-mer index): set-up and the first ROUNDS rounds of a job
mkdir -p gpurun_out/r04
DPH_PROFILE=1 timeout 1500 python3 bench.py --reads 1000000 --steps 1 --warmup 0 --max-rounds ${ROUNDS:-300} --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > gpurun_out/r04/config4_one_gpu.json 2> gpurun_out/r04/config4_one_gpu.err; echo "rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/config4_one_gpu.json') if l.startswith('{')][-1])
print('value %.2fM ms/round %.4f setup %.3f s rounds %d' % (d['value']/1e6, d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['config']['rounds_per_step']))
print(d['kernel_ms_per_round'])
print(d['per_rank'][0]['per_job'])
PY
grep "\[setup\]" gpurun_out/r04/config4_one_gpu.err | tail -6
