#!/bin/bash
# a five-slot job beside a second process that keeps a fixed amount of scattered traffic going: what do the rounds lose?
mkdir -p gpurun_out/r04
runb() { python3 bench.py --steps 6 --warmup 2 --slots 5 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   rounds: ms/round %.4f' % j['rounds_only']['ms_per_round'])"; }
echo "alone"; runb
for spec in ${SPECS:-"3 64 8" "0 4 8" "0 16 8" "0 64 8" "2 4 8" "2 16 8" "2 64 8"}; do
  spec=${spec//,/ }
  set -- $spec
  tools/micro/gather_rate bg $1 $2 ${BG_SECONDS:-45} $3 > gpurun_out/r04/bg.tmp &
  BP=$!
  sleep 3
  echo "beside mode $1 blocks $2 footprint $3 GiB"; runb
  kill -0 $BP 2>/dev/null && echo '   (background still running when the job ended: it covered all of it)'; wait $BP; cat gpurun_out/r04/bg.tmp
done 2>&1 | tee gpurun_out/r04/background_load.txt
