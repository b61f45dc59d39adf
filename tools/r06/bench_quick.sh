#!/bin/bash
R=gpurun_out/r06; mkdir -p $R
timeout 900 python3 bench.py --steps 3 --warmup 1 --map-leg-repeats 0 --scan-leg-rounds 0 --cpu-rounds 0 > $R/bench_quick.json 2> $R/bench_quick.err; echo rc=$?
python3 - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r06/bench_quick.json") if l.startswith("{")][-1])
print("k13 value %.2f M ms/round %.4f parity %s"%(j["value"]/1e6,j["rounds_only"]["ms_per_round"],j["parity"]))
for k,v in j["kernels_per_round"].items(): print("  k13 %-22s %.4f ms %10.0f B %8.1f GB/s %.4f"%(k,v["ms"],v["algorithmic_bytes"],v["GBs"],v["frac_of_hbm_peak"]))
d=j["overlap_default_k10_job"]
print("k10 job %.4f s setup %.3f ms/round %.4f parity %s"%(d["wall_s"],d["setup_s"],d["ms_per_round"],d["parity"]))
for k,v in d["kernels_per_round"].items(): print("  k10 %-22s %.4f ms %12.0f B %8.1f GB/s %.4f"%(k,v["ms"],v["algorithmic_bytes"],v["GBs"],v["frac_of_hbm_peak"]))
print("dense leg 1 slot:", j["index_query_dense"]["query_kernel"], j["index_query_dense"]["parity"])
print("dense leg 5 slots:", j["index_query_dense_slots"]["query_kernel"])
PY
tail -3 $R/bench_quick.err
