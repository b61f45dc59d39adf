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
# GPU busy time (union of kernel intervals) over the rounds (from the first to the last chain_walk_kernel)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
cw = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("chain_walk_kernel"))
if cw:
    t0, t1 = cw[0][0], cw[-1][1]
    busy = 0; cs = ce = None; ksum = 0
    for s_, e_ in ev:
        if e_ < t0 or s_ > t1: continue
        ksum += e_ - s_
        if ce is None or s_ > ce:
            if ce is not None: busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    nr = len(cw) // 2
    print("  rounds window %.1f ms, %d rounds (%.3f ms/round): GPU busy (union) %.1f ms = %.0f%%, sum of kernel times %.1f ms (%.3f ms/round)" % ((t1-t0)/1e6, nr, (t1-t0)/1e6/max(1,nr), busy/1e6, 100.0*busy/(t1-t0), ksum/1e6, ksum/1e6/max(1,nr)))
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
