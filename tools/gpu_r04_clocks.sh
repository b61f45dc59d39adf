#!/bin/bash
# GPU clocks and power while one slot / five slots run (rocm-smi sampled twice a second)
mkdir -p gpurun_out/r04
for S in 1 5; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/r04/clocks_s$S.txt &
  SP=$!
  python3 bench.py --steps 12 --warmup 2 --slots $S --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('slots $S ms/round', j['rounds_only']['ms_per_round'])"
  kill $SP
  echo "--- slots $S"; sort gpurun_out/r04/clocks_s$S.txt | uniq -c | sort -rn | head -8
done
