#!/bin/bash
# kernel trace of a short bench run: per-kernel totals, GPU busy time (union of intervals) vs wall of the traced region
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace -- python3 bench.py --steps ${STEPS:-200} --warmup 8 --cpu-rounds 0 --index-steps 0 > gpurun_out/trace_bench.json 2> gpurun_out/trace_bench.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/trace/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]) for r in rows]
ev.sort()
# restrict to the steady pipeline: from the 1st chain_kernel after 30% of events to the last
names = collections.Counter(e[2] for e in ev)
chain = [e for e in ev if e[2].startswith("chain_kernel")]
t0, t1 = chain[len(chain)//4][0], chain[-1][1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy = 0; cur_s = cur_e = None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.defaultdict(int); cnt = collections.Counter()
for s, e, n in sel: tot[n] += e - s; cnt[n] += 1
rounds = cnt[[n for n in cnt if n.startswith("chain_kernel")][0]]
print("window %.1f ms, %d rounds, GPU busy (union) %.1f ms = %.0f%%, sum of kernel times %.1f ms" % ((t1-t0)/1e6, rounds, busy/1e6, 100*busy/(t1-t0), sum(tot.values())/1e6))
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:12]:
    print("  %-42s %6d calls  %8.3f ms total  %7.1f us avg  %6.3f ms/round" % (n, cnt[n], v/1e6, v/1e3/cnt[n], v/1e6/rounds))
print(open("gpurun_out/trace_bench.json").read()[:200])
PY
ls gpurun_out/trace/*/ | head
