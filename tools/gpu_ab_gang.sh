#!/bin/bash
# A/B of the gang size (rounds per launch) on the default bench workload: DPH_GANG x slots
mkdir -p gpurun_out/r03
for cfg in ${CFGS:-1:5 4:4 4:8 2:8 4:12 8:8 8:16}; do
  G=${cfg%%:*}; S=${cfg##*:}
  DPH_GANG=$G timeout 600 python3 bench.py --steps ${STEPS:-4} --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots $S > gpurun_out/r03/ab_gang${G}_slots${S}.json 2> gpurun_out/r03/ab_gang${G}_slots${S}.err
  echo "gang=$G slots=$S rc=$? $(python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r03/ab_gang${G}_slots${S}.json'))
    print('value %.2fM ms/job %.1f rounds_only %.4f ms/round parity %s kernel_ms %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['parity']['paf_sha256_matches_oracle_fixture'], {k: round(v,3) for k,v in d['kernel_ms_per_round'].items()}))
except Exception as e:
    print('failed', e); print(open('gpurun_out/r03/ab_gang${G}_slots${S}.err').read()[-1500:])
")"
done
