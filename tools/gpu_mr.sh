#!/bin/bash
mkdir -p gpurun_out
export DP_BENCH_BACKEND=gloo DP_BENCH_SAME_DEVICE=1
for n in 2 4; do
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 120 --warmup 8 --cpu-rounds 0 ${MR_EXTRA} > gpurun_out/bench_mr$n.json 2> gpurun_out/bench_mr$n.err; echo "mr$n rc=$?"
tail -3 gpurun_out/bench_mr$n.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/bench_mr$n.json').read().strip().split('\n')[-1])
    print('N=$n value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'steps',d['steps'],'lines',d['paf_lines'])
except Exception as e: print('parse fail',e)
PY
done
unset DP_BENCH_BACKEND DP_BENCH_SAME_DEVICE
timeout 600 python bench.py --steps 120 --warmup 8 --cpu-rounds 0 ${MR_EXTRA} > gpurun_out/bench_mr1.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_mr1.json').read().strip().split('\n')[-1])
print('N=1 value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'steps',d['steps'],'lines',d['paf_lines'])
PY
