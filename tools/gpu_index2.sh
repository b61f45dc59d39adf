#!/bin/bash
# index-mode validation: new tests, the bench with its index_mode leg, 60 full-size rounds against the oracle, whole job timing
mkdir -p gpurun_out/index
timeout 600 python -m pytest tests/test_gpu_overlap_e2e.py -m gpu -x -q -k "kmer_index" --timeout=300 --timeout-method=thread 2>&1 | tail -3
timeout 400 python bench.py --cpu-rounds 0 > gpurun_out/index/bench_default.json 2> gpurun_out/index/bench_default.err
DP_SCAN_INDEX=1 timeout 600 python tools/full_parity.py --max-rounds 60 --slots 5 --out gpurun_out/index/parity60_index.json 2>&1 | tail -3
DP_SCAN_INDEX=1 timeout 600 python bench.py --steps 1000000 --warmup 0 --cpu-rounds 0 > gpurun_out/index/full_idx1.json 2> gpurun_out/index/full_idx1.err
python - <<'PY'
import json
for f in ("bench_default","full_idx1"):
    try:
        d=json.loads(open('gpurun_out/index/%s.json'%f).read().strip().split('\n')[-1])
        print(f,'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'steps',d['steps'],'paf',d['paf_lines'])
        print('  roofline',d['roofline'])
        print('  index_mode',d.get('index_mode'))
    except Exception as e:
        print(f,'ERR',e)
PY
tail -n 3 gpurun_out/index/bench_default.err gpurun_out/index/full_idx1.err
cat gpurun_out/index/parity60_index.json
