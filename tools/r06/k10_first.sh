#!/bin/bash
# Round 6, first call: the command's default regime (k = 10) - slot sweep 1..6 on whole config-2 jobs, then a kernel trace of the
# same job with five slots and with one slot (per-kernel durations under load and alone).
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r06
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for s in 1 2 3 4 5 6; do
  timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF > $R/k10_slots$s.json 2> $R/k10_slots$s.err; echo "slots $s rc=$?"
done
python3 - <<'PY'
import json
for s in range(1,7):
    try:
        j=json.loads([l for l in open("gpurun_out/r06/k10_slots%d.json"%s) if l.startswith("{")][-1])
        print("slots",s,"job %.3f s"%j["job_breakdown_s"]["whole_job"],"setup %.3f"%j["job_breakdown_s"]["setup_value_table_kmer_index_slots"],"ms/round %.3f"%j["rounds_only"]["ms_per_round"],{k:round(v,3) for k,v in j["kernel_ms_per_round"].items()})
    except Exception as e: print(s,e)
PY
for s in 5 1; do
  rm -rf $R/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --k 10 --steps 1 --warmup 1 --slots $s $OFF > $R/k10_ktrace_slots$s.json 2> $R/k10_ktrace_slots$s.err; echo "ktrace $s rc=$?"
  t=$(find $R/kt -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/ktrace_digest.py $t > $R/k10_kernel_trace_digest_slots$s.txt
  f=$(find $R/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/k10_kernel_stats_slots$s.csv
  rm -rf $R/kt
done
head -40 $R/k10_kernel_trace_digest_slots5.txt
