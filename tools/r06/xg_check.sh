#!/bin/bash
# (record of a dropped experiment: the kernels and DP_TUNE keys this script switches were taken out again - profiles/r06/k10_walk_read_ranges_per_xcd.txt)
# read ranges per XCD in the dense walk: variants on small inputs, then the k = 10 job with and without (alternating)
R=gpurun_out/r06; mkdir -p $R
timeout 1500 python3 -m pytest tests/test_gpu_overlap_e2e.py -x -q -k "counting_step_variants" 2>&1 | tail -4
for rep in 1 2; do for xg in 1 0; do
DP_TUNE=kx_no_xgroups=$xg timeout 600 python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > $R/xg_$xg.json 2> $R/xg_$xg.err
python3 - <<PY
import json
j=json.loads([l for l in open('$R/xg_$xg.json') if l.startswith('{')][-1]); d=j['overlap_default_k10_job']
print('no_xgroups $xg | k13 %.2f M | k10 job %.4f s ms/round %.4f parity %s'%(j['value']/1e6, d['wall_s'], d['ms_per_round'], d['parity']), {k:round(v,3) for k,v in d['kernel_ms_per_round'].items()})
PY
done; done | tee $R/k10_xgroups_ab.txt
