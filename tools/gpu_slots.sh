#!/bin/bash
# whole-job bench for several executor-slot counts
mkdir -p gpurun_out
for s in ${SLOTS_LIST:-6 8 12 16}; do
  timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots $s > gpurun_out/slots_$s.json 2> gpurun_out/slots_$s.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/slots_$s.json"))
print("slots $s: value %.2f M/s, job %.3f s, setup %.3f s, rounds %.3f s (%.3f ms/round), parity %s, cpu %.1f s/%.2f s" % (d["value"]/1e6, d["job_breakdown_s"]["whole_job"], d["job_breakdown_s"]["setup_value_table_kmer_index_slots"], d["job_breakdown_s"]["rounds"], d["rounds_only"]["ms_per_round"], d["parity"]["paf_sha256_matches_oracle_fixture"], d["host_cpu"]["cpu_s"], d["host_cpu"]["wall_s"]))
PY
done
