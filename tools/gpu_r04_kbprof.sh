#!/bin/bash
# per-kernel times of the index build, narrow against eight-byte entries (rocprofv3 kernel stats of a short job)
R=gpurun_out/r04
mkdir -p $R
export TMPDIR=/tmp
for V in default streams original ${EXTRA_VARIANTS}; do
  rm -rf $R/kbprof_$V
  unset DP_KINDEX_WIDE DP_KB_B1 DP_KB_STREAMS
  case $V in
    streams) export DP_KB_STREAMS=1;;
    original) export DP_KINDEX_WIDE=1;;
    b1_8) export DP_KB_B1=8;;
  esac
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kbprof_$V -- python3 bench.py --steps 2 --warmup 1 --max-rounds 10 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > $R/kbprof_$V.json 2> $R/kbprof_$V.err; echo "== $V rc=$?"
  python3 - <<PY
import csv, glob
fs = glob.glob("$R/kbprof_$V/**/*kernel_stats.csv", recursive=True)
if not fs:
    print(open("$R/kbprof_$V.err").read()[-600:])
else:
    for r in csv.DictReader(open(fs[0])):
        n = r["Name"]
        if n.startswith("kb_") or "values" in n or n.startswith("hist"):
            print("%-28s calls %3s avg %9.1f us" % (n.split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $R/kbprof_$V
done 2>&1 | tee $R/kbuild_kernels.txt
