#!/bin/bash
# round 4: the huge consensus layout - layouts test, the dense leg (fixture hash), copy sites after the change
mkdir -p gpurun_out/r04; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "consensus_layouts or paf_bit_exact or config1" > gpurun_out/r04/cons_tests.log 2>&1; echo "cons tests rc $?"; tail -3 gpurun_out/r04/cons_tests.log
python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "k10 or dense" > gpurun_out/r04/dense_tests.log 2>&1; echo "dense tests rc $?"; tail -3 gpurun_out/r04/dense_tests.log
DP_LIB_DIR=$PWD/downpore_amd/lib_copylog timeout 600 python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds 12 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots 1 > gpurun_out/r04/dense_copylog2.json 2> gpurun_out/r04/dense_copylog2.err; echo "copylog rc $?"
grep copylog gpurun_out/r04/dense_copylog2.err | sort -t' ' -k7 -n -r | head -12
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04/dense_copylog2.json'))
print('dense 1 slot: ms/round', d['rounds_only']['ms_per_round'], d['kernel_ms_per_round'])
PY
timeout 900 python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --map-leg-repeats 0 > gpurun_out/r04/bench_dense2.json 2> gpurun_out/r04/bench_dense2.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_dense2.json'))
print('value %.2fM ms/round %.4f' % (d['value']/1e6, d['rounds_only']['ms_per_round']), 'dense', json.dumps(d['index_query_dense'])[:700])
PY
