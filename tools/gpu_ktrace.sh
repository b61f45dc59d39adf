#!/bin/bash
# rocprofv3 kernel trace of a short bench run (first ROUNDS rounds of one job): per-kernel call counts and durations
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/ktrace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktrace -- python3 bench.py --steps 1 --warmup 0 --max-rounds ${ROUNDS:-150} --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots ${SLOTS:-6} ${BENCH_ARGS} > gpurun_out/ktrace_bench.json 2> gpurun_out/ktrace_bench.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ktrace/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
tot = collections.defaultdict(int); cnt = collections.Counter(); seq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:44]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[n] += d; cnt[n] += 1; seq[n].append(d)
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:24]:
    s = sorted(seq[n])
    print("  %-46s %6d calls  %9.3f ms total  %8.1f us avg  %8.1f us median  %8.1f us max" % (n, cnt[n], v/1e6, v/1e3/cnt[n], s[len(s)//2]/1e3, s[-1]/1e3))
# chain_walk_kernel is launched 3x per round (modes 0,1,2): split by position
w = seq.get("chain_walk_kernel", [])
for m in range(3):
    x = sorted(w[m::3])
    if x: print("  chain_walk mode %d: avg %.1f us median %.1f us" % (m, sum(x)/len(x)/1e3, x[len(x)//2]/1e3))
sp = seq.get("chain_spec_kernel", [])
for m in range(2):
    x = sorted(sp[m::2])
    if x: print("  chain_spec pass %d: avg %.1f us median %.1f us" % (m, sum(x)/len(x)/1e3, x[len(x)//2]/1e3))
PY
