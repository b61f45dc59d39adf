#!/bin/bash
# where a k = 13 job's set-up goes: the host's profile marks (DPH_PROFILE=1) of the third job of a process, then five plain jobs
R=gpurun_out/r06; mkdir -p $R
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 3 --warmup 2 --map-leg-repeats 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --cpu-rounds 0 > $R/setup_marks.json 2> $R/setup_marks.err
grep -E "^\[(setup|init|kb|values|overlap)" $R/setup_marks.err | tail -40
python3 - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r06/setup_marks.json") if l.startswith("{")][-1])
print("value %.2f M  ms/step %.2f  setup %.2f ms"%(j["value"]/1e6, j["ms_per_step"], 1e3*j["job_breakdown_s"]["setup_value_table_kmer_index_slots"]))
PY
