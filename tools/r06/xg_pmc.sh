#!/bin/bash
# (record of a dropped experiment: the kernels and DP_TUNE keys this script switches were taken out again - profiles/r06/k10_walk_read_ranges_per_xcd.txt)
# walk_bin's HBM bytes with and without read ranges per XCD (k = 10, first 6 rounds)
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r06
COMMON="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --k 10 --max-rounds 6"
for no in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  export DP_TUNE=kx_no_xgroups=$no
  rm -rf $R/xgpmc_${no}_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/xgpmc_${no}_$c -- python3 bench.py $COMMON > /dev/null 2> $R/xgpmc.err
  f=$(find $R/xgpmc_${no}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $no $c <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for key in ("kidx_walk_bin<16>", "kidx_bin_sort_dense", "kidx_bin_count"):
        if key in n: acc[key].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("no_xgroups", sys.argv[2], sys.argv[3], k, "launches", len(v), "mean MiB %.1f" % (sum(v) / len(v) / 1024))
PY
  rm -rf $R/xgpmc_${no}_$c
done; done | tee $R/k10_xgroups_pmc.txt
