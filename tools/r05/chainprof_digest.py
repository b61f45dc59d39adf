#!/usr/bin/env python3
"""Means over a job's rounds of the chaining kernels' per-wave phase timers ([chain prof] lines of a PROF=1 build, DP_CHAIN_PROF=1):
us per pair of every phase of every pass.  Usage: chainprof_digest.py stderr_file [skip_first_n_rounds]"""
import collections
import re
import sys

skip = int(sys.argv[2]) if len(sys.argv) > 2 else 20
acc = collections.defaultdict(lambda: collections.defaultdict(list))
seen = collections.Counter()
for ln in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[chain prof\] pass (-?\d+) .*?: (\d+) pairs \((\d+) chained\) on (\d+) waves .*?kernel span ([\d.]+) us, busiest wave ([\d.]+) us \| us per pair: "
                 r"records ([\d.]+) stage a ([\d.]+) prefilter ([\d.]+) chain_pair ([\d.]+) \| per chained pair: stage b \+ flags ([\d.]+) initial ([\d.]+) "
                 r"events \+ walk ([\d.]+) \(b events ([\d.]+) us; ([\d.]+) events; ([\d.]+) % perfect chains\) out ([\d.]+)", ln)
    if not m:
        continue
    ps = int(m.group(1))
    seen[ps] += 1
    if seen[ps] <= skip:
        continue
    names = ["pairs", "chained", "waves", "kernel_span_us", "busiest_wave_us", "records", "stage_a", "prefilter", "chain_pair", "stage_b_flags", "initial",
             "events_walk", "b_events_us", "events", "perfect_pct", "out"]
    for nm, v in zip(names, m.groups()[1:]):
        acc[ps][nm].append(float(v))
for ps in sorted(acc):
    d = {k: sum(v) / len(v) for k, v in acc[ps].items()}
    print("pass %2d (%4d rounds): pairs %7.0f chained %6.0f waves %5.0f | span %6.1f us busiest wave %6.1f us | us per pair: records %.3f stage_a %.3f prefilter %.3f "
          "chain_pair %.3f | per chained pair: stage_b %.3f initial %.3f events+walk %.3f out %.3f" %
          (ps, len(acc[ps]["pairs"]), d["pairs"], d["chained"], d["waves"], d["kernel_span_us"], d["busiest_wave_us"], d["records"], d["stage_a"], d["prefilter"],
           d["chain_pair"], d["stage_b_flags"], d["initial"], d["events_walk"], d["out"]))
