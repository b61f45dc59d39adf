#!/usr/bin/env python3
"""A/B runner for one GPU box: variants are 'label:dir:ENV=V,ENV=V' (dir '.' = working tree, '_ab/prev' = the worktree of the
previous commit); runs them in alternation REPS times and prints min / median of the rounds-only ms per round and of the whole job.
Boxes differ by 10 % and some are erratic: only numbers from one call compare."""
import json, os, statistics, subprocess, sys
reps = int(os.environ.get("REPS", "4"))
slots = os.environ.get("SLOTS", "5")
extra = os.environ.get("BENCH_ARGS", "").split()
variants = []
for v in sys.argv[1:]:
    label, d, envs = (v.split(":") + ["", ""])[:3]
    variants.append((label, d or ".", dict(e.split("=") for e in envs.split(",") if e)))
res = {v[0]: [] for v in variants}
for r in range(reps):
    for label, d, env in variants:
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-rounds", "0", "--scan-leg-rounds", "0",
                            "--dense-leg-rounds", "0", "--dense-job", "0", "--map-leg-repeats", "0", "--slots", slots] + extra, cwd=d, env=e, capture_output=True, text=True, timeout=900)
        try:
            j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
            res[label].append((j["rounds_only"]["ms_per_round"], j["job_breakdown_s"]["whole_job"], j["job_breakdown_s"]["setup_value_table_kmer_index_slots"],
                               j["parity"]["paf_sha256_matches_oracle_fixture"],
                               (j.get("per_rank") or [{}])[0].get("per_job", {}).get("slot_wait_for_plan_us", 0) / 1e3,
                               (j.get("per_rank") or [{}])[0].get("per_job", {}).get("plan_compute_us", 0) / 1e3,
                               tuple((j.get("per_rank") or [{}])[0].get("per_job", {}).get(k, 0) / 1e3 for k in
                                     ("commit_thread_wait_us", "commit_text_us", "commit_state_us", "commit_keep_text_us", "formatter_busy_us", "commit_wait_for_formatter_us")),
                               (j.get("host_cpu") or {}).get("cpu_s", 0) / max(1e-9, (j.get("host_cpu") or {}).get("wall_s", 1)), (j.get("host_cpu") or {}).get("throttled_s", 0),
                               (j.get("per_rank") or [{}])[0].get("planner_lanes_at_the_end")))
        except Exception as ex:
            print(label, "failed:", ex, p.stderr[-300:])
for label, v in res.items():
    if not v: continue
    ms = sorted(x[0] for x in v); job = sorted(x[1] for x in v)
    print("%-14s ms/round min %.3f med %.3f | job min %.3f med %.3f | setup med %.3f | parity %s | slots waited for plans %.1f ms, plan computes %.1f ms per job | all %s" %
          (label, ms[0], statistics.median(ms), job[0], statistics.median(job), statistics.median(x[2] for x in v), all(x[3] for x in v),
           statistics.median(x[4] for x in v), statistics.median(x[5] for x in v), " ".join("%.3f" % x for x in ms)))
    print("   committing thread, ms per job: waits %.1f, text %.1f, state %.1f, keeps text %.1f | formatter threads busy %.1f, commit waited for them %.1f" % tuple(statistics.median(x[6][i] for x in v) for i in range(6)))
    print("   host: %.1f cores busy on average, throttled %.2f s, planner lanes at the end %s" % (statistics.median(x[7] for x in v), max(x[8] for x in v), [x[9] for x in v]))
print("host threads:", os.cpu_count())
