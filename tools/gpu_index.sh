#!/bin/bash
# Compare scan mode vs resident k-mer index mode (DP_SCAN_INDEX) on config 2 and the config-4 scale.
mkdir -p gpurun_out/index
for m in 0 1; do
  DP_SCAN_INDEX=$m timeout 300 python bench.py --cpu-rounds 0 > gpurun_out/index/c2_idx$m.json 2> gpurun_out/index/c2_idx$m.err
  DP_SCAN_INDEX=$m timeout 600 python bench.py --reads 1000000 --seed 4 --steps 60 --cpu-rounds 0 > gpurun_out/index/c4_idx$m.json 2> gpurun_out/index/c4_idx$m.err
done
for f in gpurun_out/index/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d.get("roofline"), d.get("other_kernels"))
except Exception as e:
    print("ERR", e)
PY
done
tail -3 gpurun_out/index/*.err
