#!/bin/bash
# rocprofv3 --kernel-trace --stats of the timed workload ALONE (the k = 13 config-2 jobs of the default bench command, its other legs
# switched off): per-kernel averages and the kernel groups' time per round, to be held against bench.py's own HIP-event figures
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r06
rm -rf $R/kt_main
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_main -- python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 > $R/bench_main_only_under_rocprof.json 2> $R/kt_main.err; echo "rc=$?"
f=$(find $R/kt_main -name "*kernel_stats.csv" | head -1); cp $f $R/bench_main_only_kernel_stats.csv
python3 - $R/bench_main_only_kernel_stats.csv $R/bench_main_only_under_rocprof.json <<'PY' | tee $R/bench_main_only_kernel_groups.txt
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
jobs = j["steps"] + j["warmup"]
rounds = jobs * j["config"]["rounds_per_step"]
groups = {"chain kernels (pair_scan + chain_walk + chain_spec + chain_resolve)": ["pair_scan_kernel", "chain_walk_kernel", "chain_spec_kernel", "chain_resolve_kernel"],
          "query_kernel": ["query_kernel"], "consensus (match_anchor + consensus_full_kernel)": ["match_anchor_kernel", "consensus_full_kernel"],
          "index build (chunk + index_fill + posting_transpose + posting_meta + zero_regions)": ["chunk_kernel", "index_fill", "posting_transpose", "posting_meta", "zero_regions"],
          "index counting step (kidx_prepare + kidx_walk_bin + kidx_bin_count + kidx_offsets)": ["kidx_prepare", "kidx_walk_bin", "kidx_bin_count", "kidx_offsets"],
          "index write step (kidx_bin_fill + kidx_sortwrite)": ["kidx_bin_fill", "kidx_sortwrite", "kidx_bin_sort_dense"]}
print("rocprofv3 --kernel-trace --stats of %d config-2 jobs (%d rounds), five slots; kernel time per round by group vs bench.py's HIP events of the same run" % (jobs, rounds))
ev = j["kernels_per_round"]
names = {"chain": "chain_kernels", "query": "query_kernel", "consensus": "consensus_kernel", "index build": "index_build", "index counting": "index_counting_step", "index write": "index_write_step"}
for g, keys in groups.items():
    tot = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in keys))
    calls = sum(int(r["Calls"]) for r in rows if any(k in r["Name"] for k in keys))
    mine = [v for k, v in names.items() if g.startswith(k)][0]
    print("%-86s %8d launches  %.4f ms per round (rocprof)   %.4f ms per round (bench.py, HIP events on the slots' streams)" % (g, calls, tot / 1e6 / rounds, ev[mine]["ms"]))
print("bench line of this run: %.2f M overlaps/s, %.2f ms per job" % (j["value"] / 1e6, j["ms_per_step"]))
PY
rm -rf $R/kt_main
