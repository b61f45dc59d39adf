#!/bin/bash
# parity subset for the chaining stage + the k = 10 and k = 13 jobs (one and five slots)
R=gpurun_out/r06; mkdir -p $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -m gpu -x -q -k "scan_index_query_chain or chain_shortcuts or paf_bit_exact or config1_k10 or other_query_types or identical_and_repetitive or flag" 2>&1 | tail -5
timeout 600 python3 -m pytest tests/test_gpu_flag_matrix.py -m gpu -x -q 2>&1 | tail -3
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for s in 1 5; do
  timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF > $R/k10_chainfix_slots$s.json 2> $R/k10_chainfix_slots$s.err; echo "k10 slots $s rc=$?"
done
for v in 1 9; do
  DP_CHAIN_FULL_FROM=$v timeout 300 python3 bench.py --steps 5 --warmup 2 $OFF > $R/k13_fullfrom$v.json 2> $R/k13_fullfrom$v.err; echo "k13 full_from $v rc=$?"
  DP_CHAIN_FULL_FROM=$v timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 $OFF > $R/k10_fullfrom$v.json 2> $R/k10_fullfrom$v.err; echo "k10 full_from $v rc=$?"
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/k10_chainfix_slots*.json")+glob.glob("gpurun_out/r06/k1?_fullfrom*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1],"job %.4f s"%j["job_breakdown_s"]["whole_job"],"ms/round %.4f"%j["rounds_only"]["ms_per_round"],"value %.2f M"%(j["value"]/1e6),"parity",j["parity"]["paf_sha256_matches_oracle_fixture"],{k:round(v,3) for k,v in j["kernel_ms_per_round"].items()})
    except Exception as e: print(f,e)
PY
