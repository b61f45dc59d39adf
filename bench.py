#!/usr/bin/env python3
"""bench.py — overlaps/sec of the all-vs-all overlap hot path on synthetic long reads (BASELINE.json metric).

A "step" is ONE WHOLE JOB: one pass of the hot path over the batch of synthetic input, i.e. everything `downpore overlap`
does (commands/overlap.go:39-195) for BASELINE config 2 — 100 000 synthetic reads x 10 kb, genome 50 Mb (20x), k=13,
error-free (the error-free set is the throughput default at k=13) — from "packed reads resident in HBM" to
"last PAF line formatted": the k-mer value table (dp_kmer_values), the resident k-mer position index (dp_scan_prepare),
executor slots + planner, then EVERY round (599): seed selection, seed occurrences of every non-ignored read, survivor
exchange (N>1), index build, index query + chaining, consensus, PAF.  Nothing is carried over between steps except the
packed reads; the timed region is K consecutive jobs between two barriers.  `value` = PAF lines / wall time.
The PCIe-inclusive rate (upload + packing of the ASCII reads added to every job) is reported beside it as
`value_incl_upload`; the steady-state rate of the rounds alone as `rounds_only`.
The PAF of the first job is checked against the committed oracle fixture (tests/golden_full/config2.json: SHA-256 over all
3.9 M lines); every timed job must print the same number of lines.
`roofline` describes the kernel with the largest accumulated time of a job.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run with N ranks on
    127.0.0.1 as a child process and hand its exit code back."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


def main():
    if os.environ.get("DP_BENCH_AFFINITY", "").startswith("node"):  # experiment: every thread of the job on one NUMA node's CPUs
        try:
            os.sched_setaffinity(0, parse_cpulist(open("/sys/devices/system/node/%s/cpulist" % os.environ["DP_BENCH_AFFINITY"]).read()))
        except Exception as ex:
            print("DP_BENCH_AFFINITY:", ex, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed whole jobs")
    ap.add_argument("--warmup", type=int, default=1, help="untimed whole jobs before them")
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--error", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--seed-batch-size", type=int, default=10000)
    ap.add_argument("--max-rounds", type=int, default=-1, help="cut every job after this many rounds (experiments only)")
    ap.add_argument("--cpu-rounds", type=int, default=2, help="oracle rounds timed for cpu_baseline (0 = skip)")
    ap.add_argument("--slots", type=int, default=5, help="rounds executed concurrently per GPU (executor slots)")
    ap.add_argument("--scan-leg-rounds", type=int, default=60,
                    help="N=1: after the timed region, this many rounds with the scan kernels instead of the k-mer index "
                         "(reported as scan_kernels_leg with the count pass's achieved GB/s; 0 = skip)")
    ap.add_argument("--dense-leg-rounds", type=int, default=12,
                    help="N=1: after the timed region, this many rounds of the dense-seed regime (k=10) where the index query "
                         "carries real traffic (reported as index_query_dense; 0 = skip)")
    ap.add_argument("--dense-job", type=int, default=1,
                    help="N=1: after the timed region, ONE whole job with the command's default k = 10 on the same reads (every read indexed "
                         "in every round), reported as overlap_default_k10_job with ground truth and first-16-round fixture parity (0 = skip)")
    ap.add_argument("--map-cpu-baseline", type=int, default=1,
                    help="map_config3: also time the oracle's mapper on the host (about 5 s, one core) as its cpu_baseline (0 = skip)")
    ap.add_argument("--map-leg-repeats", type=int, default=8,
                    help="N=1: after the timed region, BASELINE config 3 (`downpore map`: 50k reads x 8 kb against a 4.6 Mb circular "
                         "reference, k=11) this many times (value = the median of the runs after the first), reported as map_config3 with its PAF held to the oracle's fixture (0 = skip)")
    ap.add_argument("--mode", default="auto", choices=["auto", "round", "round-batch", "scan-shard"],
                    help="multi-GPU decomposition (N > 1).  scan-shard: every rank runs every round on its own read range and the "
                         "survivors' seed index is all-gathered (RCCL, inside the library) - the layout for read sets that do not fit "
                         "one GPU.  round: the query batches (rounds) are dealt to the ranks, each with the whole read set resident, "
                         "results all-gathered and committed in order - rounds are the unit that shards without touching the per-round "
                         "latency, so this is what scales while a GPU holds the reads and their k-mer index (9 B per base).  auto: "
                         "round if that fits in half of the GPU's memory, else scan-shard")
    args = ap.parse_args()

    # --gpus N is the contract: N ranks, one per GPU.  Started under torch.distributed.run the ranks are there already
    # (WORLD_SIZE must then equal N); started plainly with N > 1 this process starts them itself - fresh child processes,
    # before anything here has touched the GPU or torch - relays their output (rank 0 prints the JSON line) and exits with
    # their exit code.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, env_world))
        sys.exit(2)
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    # the executor slots' streams (plus the planner's and the window cache's) want one hardware queue each: the runtime's
    # default of 4 makes streams share queues (measured: 8 slots 9.0 M overlaps/s with 8 queues against 8.7 M with 4)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and "DP_HOST_THREADS" not in os.environ:
        # one process per GPU on ONE host: the ranks share the container's CPU quota, so each gets its share of worker threads
        os.environ["DP_HOST_THREADS"] = str(max(2, cpu_budget() // world - (1 if cpu_budget() // world > 3 else 0)))
    if world > 1 and not any(a == "--slots" or a.startswith("--slots=") for a in sys.argv[1:]):
        # an executor slot is a thread that spins on its stream; a rank short of cores runs fewer of them (measured with two
        # ranks on a 16-core quota: 4 slots each 9.4 M overlaps/s, 6 slots each 4.4 M)
        args.slots = max(2, min(args.slots, int(os.environ["DP_HOST_THREADS"]) - 3))
    import torch
    torch_device = None
    if world > 1:
        import torch.distributed as dist
        # test hooks for a 1-GPU box: DP_BENCH_SAME_DEVICE=1 puts every rank on GPU 0, DP_BENCH_BACKEND=gloo exchanges on
        # the host (RCCL refuses two ranks on one GPU).  The driver's multi-GPU runs use neither.
        if os.environ.get("DP_BENCH_SAME_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if os.environ.get("DP_BENCH_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo")
            torch_device = None
        else:
            torch_device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=torch_device)
    else:
        dist = None

    if os.environ.get("DP_BENCH_DEBUG"):  # hung-run diagnosis: Python stacks of every rank every 40 s
        import faulthandler
        faulthandler.dump_traceback_later(40, repeat=True)
    from tools.synth import gen_reads
    from downpore_amd.overlap import OverlapPipeline, Reads

    N, L = args.reads, args.read_len
    alt_mode_name = None
    if args.mode == "auto":
        hbm = torch.cuda.get_device_properties(local_rank).total_memory if torch.cuda.is_available() else 0
        if world == 1:
            args.mode = "round"  # (one rank: the plain executor pipeline)
        else:
            # N > 1.  north_star's sentence has two halves - "the query batch shards naturally across reads, so partition across the
            # GPUs" and "RCCL all-gather of the seed index" - and the repo has a layout for each:
            #   round       the query batches (rounds) are dealt to the ranks, every rank holds the reads and their k-mer position
            #               index (9 B per base), finished rounds are all-gathered over RCCL and committed in order: the unit that
            #               shards without touching a round's latency - what scales while a GPU holds the read set;
            #   scan-shard  the reads are partitioned, every rank runs every round on its own range and the round's survivors - its
            #               seed index - are all-gathered over RCCL: the layout for read sets whose index does not fit one GPU.
            # The headline is the one predicted to scale (HISTORY.md 7.3: round 3.4 x / scan-shard < 2 x at 8 GPUs for config 2) where it
            # fits in half of the GPU's memory, the other one runs after it on the same reads and is reported as `alt_mode`.
            if 9 * N * L < hbm // 2:
                args.mode = "round"
                alt_mode_name = "scan-shard"
            else:
                args.mode = "scan-shard"
    G = N * L // 20
    t0 = time.time()
    from tools.synth import gen_reads_truth
    bases, off, truth_starts, truth_strands = gen_reads_truth(args.seed, G, N, L, args.error, False)  # (the same reads as gen_reads')
    reads = Reads(bases, off, min_len=1000)
    t_gen = time.time() - t0
    # N > 1, scan-shard (default): the survivor exchange runs inside the library on an RCCL communicator (dp_comm_init +
    # dp_allgather_survivors, device to device); DP_BENCH_BACKEND=gloo (1-GPU test hook) keeps it on host copies
    # (DP_BENCH_FORCE_SHARD=1, test hook: the sharded batch pipeline with a communicator of one rank on a 1-GPU box)
    force_shard = world == 1 and os.environ.get("DP_BENCH_FORCE_SHARD") == "1" and args.mode == "scan-shard"
    comm = "rccl" if ((world > 1 and torch_device is not None) or force_shard) else None  # (both layouts exchange inside the library)
    pipe = OverlapPipeline(reads, device=local_rank, k=args.k, seed_batch_size=args.seed_batch_size, rank=rank, world=world,
                           torch_device=torch_device, mode=args.mode, slots=args.slots, defer_init=True, comm=comm)
    upload = pipe.setup_times()
    if world > 1 and rank != 0:
        pipe.keep_text(False)  # (rank 0 verifies and would print the PAF; the others commit the gathered rounds without their text)
    if world > 1 and comm is not None and getattr(pipe, "mode", "") == "round" and os.environ.get("DP_BENCH_TEXT_ROOT") == "1":
        # (opt-in until a run with two GPUs has shown parity: the text gathered to rank 0 alone uses ncclSend / ncclRecv, which no run of
        # this repository has executed with a peer yet; the default exchange is the all-gather of the rounds' blobs, as in round 4)
        pipe.text_root(0)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    golden = golden_fixture(args)
    checks = {"fixture": golden["case"] if golden else None, "paf_sha256_matches_oracle_fixture": None, "jobs_with_equal_line_count": 0}

    def job(verify=False):
        """One whole job on the resident reads.  Returns (PAF lines, rounds, per-job kernel/phase totals, seconds of init)."""
        pipe.init()
        lines = rounds = 0
        t_prev = time.perf_counter()
        gaps = res["step_gaps"] = []   # (the longest waits for a step of this job: a stall shows as one long gap, a slow job as many)
        while args.max_rounds < 0 or rounds < args.max_rounds:
            c = pipe.step()
            t_now = time.perf_counter()
            if t_now - t_prev > 1e-3:
                gaps.append((round((t_now - t_prev) * 1e3, 2), rounds))
            t_prev = t_now
            if c == 0:
                break
            rounds += c
            lines += pipe.step_lines()
        tot = pipe.stats_total()
        t_init = pipe.setup_times()["init_s"]
        if verify and rank == 0:
            paf = pipe.all_paf()
            res["ground_truth"] = ground_truth(paf, off, truth_starts, truth_strands, args.k)
            res["first_job_paf"] = paf if args.cpu_rounds > 0 and world == 1 else None
            res["values"] = pipe.values() if args.cpu_rounds > 0 and world == 1 else None
            if golden is not None and args.max_rounds < 0:
                checks["paf_sha256_matches_oracle_fixture"] = bool(hashlib.sha256(paf.encode()).hexdigest() == golden["paf_sha256"] and
                                                                   lines == golden["paf_lines"] and rounds == golden["rounds"])
        t_r = time.perf_counter()
        pipe.reset()
        res["reset_s"] = res.get("reset_s", 0.0) + (time.perf_counter() - t_r) if not verify else res.get("reset_s", 0.0)
        return lines, rounds, tot, t_init

    res = {}
    verified = False
    for _ in range(args.warmup):
        job(verify=not verified)
        verified = True
    sync()

    def cpu_stat():
        out = {}
        try:
            for ln in open("/sys/fs/cgroup/cpu.stat"):
                k_, v_ = ln.split()
                out[k_] = int(v_)
        except Exception:
            pass
        return out

    def pipe_counters():
        H = pipe.H
        H.dph_planner_counter.restype = __import__("ctypes").c_int64
        names = ["plans_computed", "plans_thrown_away", "plans_erased_by_flags", "rounds_executed", "rounds_rejected", "rounds_committed",
                 "plan_compute_us", "slot_wait_for_plan_us", "commit_thread_wait_us", "commit_text_us", "commit_state_us", "commit_keep_text_us", "formatter_busy_us", "commit_wait_for_formatter_us", "planner_lanes_now"]
        return {nm: int(H.dph_planner_counter(i)) for i, nm in enumerate(names)}

    cs0 = cpu_stat()
    pc0 = pipe_counters()
    acc, lines, rounds, t_init_sum, per_job, per_job_parts = {}, 0, 0, 0.0, [], []
    t_start = time.perf_counter()
    pcj = pc0
    for _ in range(args.steps):
        tj = time.perf_counter()
        jl, jr, tot, ti = job()
        per_job.append(time.perf_counter() - tj)
        # (where a slow job lost its time: set-up, slots waiting for plans, the commit thread waiting for the formatters - ms)
        pcn = pipe_counters()
        per_job_parts.append([round(ti * 1e3, 2), round((pcn["slot_wait_for_plan_us"] - pcj["slot_wait_for_plan_us"]) / 1e3, 2),
                              round((pcn["commit_wait_for_formatter_us"] - pcj["commit_wait_for_formatter_us"]) / 1e3, 2),
                              round((pcn["commit_thread_wait_us"] - pcj["commit_thread_wait_us"]) / 1e3, 2)])
        per_job_parts[-1].append(sorted(res.get("step_gaps", []), reverse=True)[:4])
        pcj = pcn
        lines += jl
        rounds += jr
        t_init_sum += ti
        for key, v in tot.items():
            acc[key] = acc.get(key, 0.0) + v
        if lines == jl * len(per_job):
            checks["jobs_with_equal_line_count"] = len(per_job)
    t_local = time.perf_counter() - t_start  # (this rank's own: before the barrier)
    sync()
    elapsed = time.perf_counter() - t_start
    cs1 = cpu_stat()
    # per-rank stage times of the timed jobs, gathered on rank 0 (so that a first multi-GPU run explains itself): set-up, rounds,
    # what this rank's planner and executor slots did
    pc1 = pipe_counters()
    nj = max(1, args.steps)
    mine = {"rank": rank, "jobs_s": t_local / nj, "setup_s": t_init_sum / nj, "rounds_s": (t_local - t_init_sum) / nj,
            "kernel_ms_per_job": {k_: acc.get(k_, 0.0) / nj for k_ in ("k_count_ms", "k_write_ms", "k_query_ms", "k_chain_ms", "k_cons_ms")},
            "phase_s_per_job": {k_: acc.get(k_, 0.0) / nj for k_ in ("t_prepare", "t_scan", "t_index", "t_query", "t_consensus")},
            "per_job": {k_: (pc1[k_] - pc0[k_]) / nj for k_ in pc1 if k_ != "planner_lanes_now"}, "slots": args.slots,
            "planner_lanes_at_the_end": pc1.get("planner_lanes_now"),
            "host_threads": int(os.environ.get("DP_HOST_THREADS", "0")) or None}
    per_rank = [mine]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch_device if torch_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not verified:  # --warmup 0: the fixture check runs on an extra, untimed job
        job(verify=True)

    # ---- secondary legs (N=1, after the timed region, never part of `value`)
    scan_leg = dense_leg = None
    if world == 1 and args.scan_leg_rounds > 0 and "DP_SCAN_INDEX" not in os.environ and acc.get("idx_rounds"):
        os.environ["DP_SCAN_INDEX"] = "0"
        scan_leg = rounds_leg(pipe, args.scan_leg_rounds, torch)
        del os.environ["DP_SCAN_INDEX"]
        m = max(1.0, scan_leg.pop("_rounds"))
        cms, cb = scan_leg["kernel_ms_per_round"]["k_count_ms"], scan_leg.pop("_count_bytes") / m
        scan_leg["scan_count_pass"] = {"kernel": "scan_kernel<0> (count pass of the packed k-mer scan, A2/A10)", "launch_ms": cms,
                                       "algorithmic_bytes_per_launch": cb, "achieved_GBs": (cb / 1e9) / (cms / 1e3) if cms > 0 else 0.0,
                                       "frac_of_hbm_peak": ((cb / 1e9) / (cms / 1e3)) / HBM_PEAK_GBS if cms > 0 else 0.0}
    pipe.close()
    alt = None
    if alt_mode_name is not None:  # collective: every rank takes part
        alt = alt_mode_jobs(alt_mode_name, reads, args, rank, world, local_rank, torch_device, comm, golden, torch, dist)
    dense_leg5 = dense_job = None
    if world == 1 and args.dense_leg_rounds > 0:
        dense_leg = dense_regime_leg(reads, args, torch)
        if args.slots > 1:
            dense_leg5 = dense_regime_leg(reads, args, torch, slots=args.slots)
    if world == 1 and args.dense_job:
        dense_job = dense_job_leg(reads, args, torch, off, truth_starts, truth_strands)

    map_leg = None
    if world == 1 and args.map_leg_repeats > 0:
        map_leg = map_config3_leg(args.map_leg_repeats, cpu=bool(args.map_cpu_baseline))

    stream_gbs = None
    if rank == 0:
        try:  # measured HBM stream rate of this device (device-to-device copy, read + write counted), SURVEY 8(d)
            nb = 1 << 30
            a_ = torch.empty(nb, dtype=torch.uint8, device="cuda")
            b_ = torch.empty(nb, dtype=torch.uint8, device="cuda")
            b_.copy_(a_)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b_.copy_(a_)
            e1.record()
            torch.cuda.synchronize()
            stream_gbs = 2.0 * nb * 10 / (e0.elapsed_time(e1) / 1e3) / 1e9
            del a_, b_
        except Exception:
            stream_gbs = None
    if rank == 0:
        n_jobs = max(1, args.steps)
        n = max(1.0, float(rounds))  # per-round averages over every round of every timed job
        main_index = bool(acc.get("idx_rounds"))

        nt = max(1.0, float(acc.get("timed_rounds", 0.0)))  # rounds whose kernels were bracketed by HIP events (every 8th of a slot)

        def per_round(key):
            return acc.get(key, 0.0) / (nt if key.startswith("k_") else n)

        def pmc_traffic(name):
            tpath = os.path.join(ROOT, "profiles", name)
            try:
                return json.load(open(tpath)).get("hbm_bytes_per_launch") if os.path.exists(tpath) else None
            except Exception:
                return None

        # kernels of a job that are launched once per round, with their algorithmic bytes and HIP-event times
        kern = {
            "chain_kernel": (per_round("k_chain_ms"), per_round("chain_bytes"), "chain_traffic.json",
                             "chain kernels (A6 prefilter + A7 chaining + A8 ratchet of every (query, candidate) pair)"),
            "query_kernel": (per_round("k_query_ms"), per_round("query_bytes"), "query_traffic.json",
                             "query_kernel (A14 + A5: soft union of the posting bitsets)"),
            # (round 4: the consensus stage is a candidate like the others - its algorithmic bytes are summed by the kernel itself, per
            # window: records, chains, anchors, trimmed segments, query segments in; PAF records, ignore ids, group record out)
            "consensus_kernel": (per_round("k_cons_ms"), per_round("cons_bytes"), "consensus_traffic.json",
                                 "consensus_full_kernel (A15 + A16 + A17 numbers: anchors + seed-space consensus + PAF fields of every query window)"),
        }
        if main_index:
            kern["index_counting_step"] = (per_round("k_count_ms"), per_round("count_bytes"), "kindex_traffic.json",
                                           "k-mer position index lookup of the round's seeds (A2/A10 without a scan)")
        else:
            kern["scan_kernel<0>"] = (per_round("k_count_ms"), per_round("count_bytes"), "scan_traffic.json",
                                      "scan_kernel<0> (count pass of the packed k-mer scan, A2/A10)")
        dom = max(kern, key=lambda kk: kern[kk][0])
        rl_ms, rl_bytes, rl_file, rl_desc = kern[dom]
        achieved = (rl_bytes / 1e9) / (rl_ms / 1e3) if rl_ms > 0 else 0.0
        job_s = elapsed / n_jobs
        out = {
            "metric": "overlaps/sec (all-vs-all PAF)", "value": lines / elapsed if elapsed > 0 else 0.0, "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * job_s,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "whole `overlap` job per step: %d synthetic reads x %d bp, genome %d bp (20x), error %.3g, k=%d, all %d "
                                   "rounds from packed reads resident in HBM to the last PAF line (BASELINE config 2)"
                                   % (N, L, G, args.error, args.k, rounds // n_jobs),
                       "reads": N, "read_len": L, "k": args.k, "seed_batch_size": args.seed_batch_size, "executor_slots_per_gpu": args.slots,
                       "rounds_per_step": rounds / n_jobs, "paf_lines_per_step": lines / n_jobs,
                       "parallelism": ("single GPU" if world == 1 else
                                       "north_star layout: reads partitioned over %d GPUs, every round's survivors (seed index) all-gathered (RCCL, "
                                       "device to device inside the library), identical index built on every rank" % world
                                       if args.mode == "scan-shard" else
                                       "query batches dealt to %d GPUs (north_star: 'the query batch shards naturally across reads'): rank r's executor "
                                       "pipeline runs the rounds r, r+N, ... on the whole read set and plans only those (the starts of the rounds in "
                                       "between are guessed and checked at the commit); finished rounds all-gathered over RCCL inside the library per "
                                       "superstep and committed in order on every rank" % world)},
            "roofline": {"bound": "hbm", "kernel": rl_desc + " - largest accumulated kernel time of a job",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(rl_file), "algorithmic_bytes_per_launch": rl_bytes, "launch_ms": rl_ms,
                         "launches_per_step": rounds / n_jobs, "measured_stream_GBs": stream_gbs},
            "parity": checks,
            # the PAF held to the synthetic genome itself (independent of the oracle): tests/test_ground_truth.py's figures on a sample
            "ground_truth": res.get("ground_truth"),
            "per_rank": per_rank,
            # the PCIe-inclusive rate: the ASCII reads cross PCIe and are packed on the device once per job (never `value`)
            "value_incl_upload": lines / (elapsed + n_jobs * upload["upload_pack_s"]) if elapsed > 0 else 0.0,
            "rounds_only": {"value": lines / max(1e-9, elapsed - t_init_sum), "unit": "overlaps/s",
                            "ms_per_round": 1e3 * (elapsed - t_init_sum) / n,
                            "note": "the same timed jobs without their set-up (value table, k-mer index, slots, planner)"},
            "job_breakdown_s": {"whole_job": job_s, "setup_value_table_kmer_index_slots": t_init_sum / n_jobs, "reset_end_of_job": res.get("reset_s", 0.0) / n_jobs,
                                "rounds": (elapsed - t_init_sum) / n_jobs, "upload_pack_once": upload["upload_pack_s"],
                                "context_once": upload["context_s"], "per_job": per_job,
                                "per_job_ms_setup_waitplan_waitfmt_commitidle_longest_step_waits": per_job_parts},
            "kernels_per_round": kernels_table(acc, rounds, acc.get("timed_rounds", 0.0), index_mode=main_index),
            "scan_mode": "resident k-mer position index" if main_index else "scan kernels",
            "scan_kernels_leg": scan_leg, "index_query_dense": dense_leg, "index_query_dense_slots": dense_leg5, "overlap_default_k10_job": dense_job, "map_config3": map_leg, "alt_mode": alt,
            "paf_lines": lines, "rounds_per_s": rounds / elapsed if elapsed > 0 else 0.0,
            "phase_ms_per_round": {kk: 1e3 * per_round(kk) for kk in ("t_prepare", "t_scan", "t_index", "t_query", "t_consensus")},
            "kernel_ms_per_round": {kk: per_round(kk) for kk in ("k_count_ms", "k_write_ms", "k_scan_ms", "k_index_ms", "k_query_ms", "k_chain_ms", "k_cons_ms")},
            "kernel_event_sampling": {"rounds_with_events": acc.get("timed_rounds", 0.0), "rounds": float(rounds),
                                      "note": "HIP events bracket the kernels of every 8th round of an executor slot in the timed jobs (every event is a packet of its own: 7 % of a job when every round carries them); the legs time every round"},
            "setup_s": {"generate": t_gen},
            # host side of the timed region: CPU seconds used by this container and time it spent throttled by its CPU quota
            "host": host_info(),
            "host_cpu": {"cpu_s": (cs1.get("usage_usec", 0) - cs0.get("usage_usec", 0)) / 1e6,
                         "throttled_s": (cs1.get("throttled_usec", 0) - cs0.get("throttled_usec", 0)) / 1e6,
                         "wall_s": elapsed},
        }
        if world == 1 and args.cpu_rounds > 0 and res.get("values") is not None:
            out["cpu_baseline"] = cpu_baseline(bases, off, args, res["values"], threads=1, check_against=res["first_job_paf"])
            # the same port with its per-read scans spread over the host cores this container may use
            out["cpu_baseline_all_cores"] = cpu_baseline(bases, off, args, res["values"], threads=cpu_budget())
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def kernels_table(tot, rounds, timed_rounds, index_mode=True):
    """Per-round kernel groups of a job with their algorithmic bytes, HIP-event times (means over the rounds whose kernels
    carried events) and what that is of the HBM peak.  Bytes: chain / query / consensus / counting step are summed by the kernels
    themselves (per pair, per query, per window, per bucket entry); the write step moves 8 B per seed occurrence (a record in, a
    (gap, seed) pair out: 8 H counted once, as SURVEY's B_scan counts the survivors' pairs); the index build reads those pairs (8 H) and
    writes the two bit matrices, 8 (S ceil(M / 64) + M ceil(S / 64)) - S seeds, M indexed sequences of the round (means over the job)."""
    n = max(1.0, float(rounds))
    nt = max(1.0, float(timed_rounds))
    H = tot.get("idx_hits", 0.0) / n
    S = tot.get("n_seeds", 0.0) / n
    M = tot.get("n_indexed", 0.0) / n
    build_bytes = 8.0 * H + 8.0 * (S * ((M + 63) // 64) + M * ((S + 63) // 64))
    rows = {
        "chain_kernels": (tot.get("k_chain_ms", 0.0) / nt, tot.get("chain_bytes", 0.0) / n, "pair_scan + chain_walk + chain_spec + chain_resolve (A6 + A7 + A8)"),
        "query_kernel": (tot.get("k_query_ms", 0.0) / nt, tot.get("query_bytes", 0.0) / n, "query_kernel (A14 + A5)"),
        "consensus_kernel": (tot.get("k_cons_ms", 0.0) / nt, tot.get("cons_bytes", 0.0) / n, "match_anchor + consensus_full_kernel (A15 + A16 + A17 numbers)"),
        "index_build": (tot.get("k_index_ms", 0.0) / nt, build_bytes, "chunk_kernel + index_fill(_rows) + posting_transpose + posting_meta (A12 + A13)"),
    }
    if index_mode:
        rows["index_counting_step"] = (tot.get("k_count_ms", 0.0) / nt, tot.get("count_bytes", 0.0) / n, "kidx_prepare + kidx_walk_bin + kidx_bin_count + kidx_offsets (A2 / A10 without a scan: count)")
        rows["index_write_step"] = (tot.get("k_write_ms", 0.0) / nt, 8.0 * H, "kidx_bin_fill + kidx_sortwrite, or kidx_bin_sort_dense (A2 / A10: the survivors' segments)")
    else:
        rows["scan_count_pass"] = (tot.get("k_count_ms", 0.0) / nt, tot.get("count_bytes", 0.0) / n, "scan_kernel<0>")
        rows["scan_write_pass"] = (tot.get("k_write_ms", 0.0) / nt, 8.0 * H, "scan_kernel<1>")
    return {kk: {"kernels": v[2], "ms": v[0], "algorithmic_bytes": v[1], "GBs": (v[1] / 1e9) / (v[0] / 1e3) if v[0] > 0 else 0.0,
                 "frac_of_hbm_peak": ((v[1] / 1e9) / (v[0] / 1e3)) / HBM_PEAK_GBS if v[0] > 0 else 0.0} for kk, v in rows.items()}


def k10_traffic():
    """PMC traffic of the k = 10 job's kernels (profiles/r06/pmc_k10.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same job,
    corrected as MI355X_MICROARCH.md prescribes; bytes per launch), or None where the file is missing."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06", "pmc_k10.json")))["kernels"]
        return {kk: v.get("hbm_bytes_per_launch") for kk, v in d.items() if v.get("hbm_bytes_per_launch")}
    except Exception:
        return None


def ground_truth(paf, off, starts, strands, k, sample=200000):
    """tools/truth.py: the first `sample` PAF lines against where the generator took the reads from."""
    from tools.truth import overlap_truth
    return overlap_truth(paf, off, starts, strands, k, sample)


def alt_mode_jobs(mode, reads, args, rank, world, local_rank, torch_device, comm, golden, torch, dist):
    """N > 1: the same whole jobs in the other multi-GPU layout, timed the same way (barrier + synchronize on both sides, MAX over
    ranks), first job held to the same fixture.  Reported next to the headline as `alt_mode`."""
    from downpore_amd.overlap import OverlapPipeline
    pipe = OverlapPipeline(reads, device=local_rank, k=args.k, seed_batch_size=args.seed_batch_size, rank=rank, world=world,
                           torch_device=torch_device, mode=mode, slots=args.slots, defer_init=True, comm=comm)
    if world > 1 and rank != 0:
        pipe.keep_text(False)
    if world > 1 and comm is not None and getattr(pipe, "mode", "") == "round" and os.environ.get("DP_BENCH_TEXT_ROOT") == "1":
        pipe.text_root(0)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def job(verify):
        pipe.init()
        lines = rounds = 0
        while args.max_rounds < 0 or rounds < args.max_rounds:
            c = pipe.step()
            if c == 0:
                break
            rounds += c
            lines += pipe.step_lines()
        ok = None
        if verify and rank == 0 and golden is not None and args.max_rounds < 0:
            paf = pipe.all_paf()
            ok = bool(hashlib.sha256(paf.encode()).hexdigest() == golden["paf_sha256"] and lines == golden["paf_lines"] and rounds == golden["rounds"])
        pipe.reset()
        return lines, rounds, ok
    ok = None
    for i in range(max(1, args.warmup)):
        _, _, o = job(i == 0)
        ok = o if i == 0 else ok
    sync()
    t0 = time.perf_counter()
    lines = rounds = 0
    for _ in range(args.steps):
        jl, jr, _ = job(False)
        lines += jl
        rounds += jr
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch_device if torch_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pipe.close()
    return {"mode": mode, "value": lines / elapsed if elapsed > 0 else 0.0, "unit": "overlaps/s", "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "rounds_per_step": rounds / max(1, args.steps), "paf_sha256_matches_oracle_fixture": ok,
            "parallelism": ("round-parallel over %d GPUs: rank r's executor pipeline runs the rounds r, r+N, ...; finished rounds all-gathered "
                            "(dp_allgather_blobs, RCCL) per superstep and committed in order on every rank" % world) if mode != "scan-shard" else
                           ("reads partitioned over %d GPUs, every round's survivors (its seed index) all-gathered (RCCL, device to device inside "
                            "the library), identical index built on every rank" % world)}


def rounds_leg(pipe, n_rounds, torch, warm=8, fixture=None):
    """`n_rounds` rounds of a fresh job on `pipe` (after `warm` untimed ones): rate and per-round kernel times.  fixture: an
    oracle fixture of exactly warm + n_rounds rounds of this job (tests/golden_full) - the leg's PAF is then held to it."""
    from downpore_amd import hip
    hip.load_library().dp_set_kernel_timing(1)  # (the legs are about kernel durations: every round carries its events)
    try:
        return _rounds_leg(pipe, n_rounds, torch, warm, fixture)
    finally:
        hip.load_library().dp_set_kernel_timing(int(os.environ.get("DP_KERNEL_TIMING", "8")))


def _rounds_leg(pipe, n_rounds, torch, warm, fixture=None):
    pipe.init()
    if fixture is not None:  # exactly the fixture's rounds are committed (a step commits every finished round it finds)
        import ctypes as C
        pipe.H.dph_overlap_set_round_limit.restype = None
        pipe.H.dph_overlap_set_round_limit.argtypes = [C.c_void_p, C.c_int64]
        pipe.H.dph_overlap_set_round_limit(pipe.h, warm + n_rounds)
    # several slots: the untimed rounds' kernels count as well (with five rounds in flight and an issue window of fifteen, most of a
    # sixteen-round leg is EXECUTED during the "warm" steps; the wall time per round is still taken over the timed steps alone)
    base0 = pipe.stats_total() if getattr(pipe, "slots", 1) > 1 else None
    got = w = 0
    while w < warm:
        c = pipe.step()
        if c == 0:
            break
        w += c
    pipe.drain()
    torch.cuda.synchronize()
    base = base0 if base0 is not None else pipe.stats_total()
    t0 = time.perf_counter()
    lines = 0
    while got < n_rounds:
        c = pipe.step()
        if c == 0:
            break
        got += c
        lines += pipe.step_lines()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = pipe.stats_total()
    parity = None
    if fixture is not None:
        paf = pipe.all_paf()
        parity = {"fixture": fixture["case"],
                  "paf_sha256_matches_oracle_fixture": bool(pipe.committed_rounds() == fixture["rounds"] and paf.count("\n") == fixture["paf_lines"] and
                                                            hashlib.sha256(paf.encode()).hexdigest() == fixture["paf_sha256"])}
        del paf
    pipe.reset()
    m = max(1, got)
    mb = max(1, got + (w if base0 is not None else 0))  # rounds the counters below cover
    d = {kk: tot.get(kk, 0.0) - base.get(kk, 0.0) for kk in tot}
    return {"value": lines / dt if dt > 0 else 0.0, "unit": "overlaps/s", "rounds": got, "ms_per_round": 1e3 * dt / m, "parity": parity,
            "kernel_ms_per_round": {kk: d.get(kk, 0.0) / max(1.0, d.get("timed_rounds", 0.0)) for kk in ("k_count_ms", "k_write_ms", "k_query_ms", "k_chain_ms")},
            "rounds_with_kernel_events": d.get("timed_rounds", 0.0),
            "query_bytes_per_round": d.get("query_bytes", 0.0) / mb, "n_indexed_per_round": d.get("n_indexed", 0.0) / mb,
            "_rounds": float(got), "_count_bytes": d.get("count_bytes", 0.0)}


def dense_regime_leg(reads, args, torch, slots=1):
    """The dense-seed regime of SURVEY 8(a) (`-k 10`, the command's default k): every read is indexed (~190 k sequences per
    round, W ~ 3 k words), so the index query streams hundreds of MB of posting words per round - the regime in which the
    north star's "HBM roofline during index-query" is a meaningful number.  Same reads, same code path."""
    from downpore_amd.overlap import OverlapPipeline
    # one executor slot: the leg is here for the kernel's own duration (its roofline), and eight rounds in flight would have
    # eight of these streaming kernels share the HBM bandwidth and each launch take several times longer
    pipe = OverlapPipeline(reads, device=0, k=10, seed_batch_size=args.seed_batch_size, slots=slots, defer_init=True)
    # the leg's rounds (4 untimed + the timed ones) are the first rounds of the k = 10 job on these reads: held to the oracle's
    # fixture for exactly those rounds when one is committed (tests/golden_full/config2_k10_e0_first_16_rounds.json)
    fixture = None
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden_full", "config2_k10_e0_first_%d_rounds.json" % (4 + args.dense_leg_rounds))))
        gen = g["generator"]
        if (gen["seed"] == args.seed and gen["reads"] == args.reads and gen["read_len"] == args.read_len and gen["error"] == args.error and
                not gen["variable"] and g["k"] == 10 and args.seed_batch_size == 10000):
            fixture = g
    except Exception:
        fixture = None
    leg = rounds_leg(pipe, args.dense_leg_rounds, torch, warm=4, fixture=fixture)
    pipe.close()
    m = max(1.0, leg.pop("_rounds"))
    leg.pop("_count_bytes")
    qms, qb = leg["kernel_ms_per_round"]["k_query_ms"], leg["query_bytes_per_round"]
    leg["query_kernel"] = {"launch_ms": qms, "algorithmic_bytes_per_launch": qb,
                           "achieved_GBs": (qb / 1e9) / (qms / 1e3) if qms > 0 else 0.0,
                           "frac_of_hbm_peak": ((qb / 1e9) / (qms / 1e3)) / HBM_PEAK_GBS if qms > 0 else 0.0}
    leg["workload"] = ("same reads, k=10 (dense seeds): %d rounds, %s" %
                       (int(m), "one executor slot (kernel durations without other rounds in flight)" if slots == 1 else
                        "%d executor slots (what the index query keeps of its rate beside other rounds' kernels)" % slots))
    leg["slots"] = slots
    return leg


def dense_job_leg(reads, args, torch, off, truth_starts, truth_strands):
    """The command's DEFAULT regime as a whole job: `downpore overlap` with its default k = 10 (commands/overlap.go:25) on the config-2
    reads - value table, k-mer position index, every round - timed once; its PAF is held to the synthetic genome (tools/truth.py) and
    its first 16 rounds to the oracle's fixture (the prefix of the job's PAF)."""
    from downpore_amd.overlap import OverlapPipeline
    pipe = OverlapPipeline(reads, device=0, k=10, seed_batch_size=args.seed_batch_size, slots=args.slots, defer_init=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.init()
    t_init = time.perf_counter() - t0
    sha_head, fixture, head_left = None, None, 0
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden_full", "config2_k10_e0_first_16_rounds.json")))
        gen = g["generator"]
        if (gen["seed"] == args.seed and gen["reads"] == args.reads and gen["read_len"] == args.read_len and gen["error"] == args.error and
                not gen["variable"] and args.seed_batch_size == 10000):
            fixture, head_left, sha_head = g, g["paf_lines"], hashlib.sha256()
    except Exception:
        fixture = None
    import ctypes as C
    lines = rounds = 0
    sample = []
    sample_lines = 0
    while True:
        c = pipe.step()
        if c == 0:
            break
        rounds += c
        n = C.c_int64(0)
        ptr = pipe.H.dph_overlap_round_paf(pipe.h, C.byref(n))
        text = C.string_at(ptr, n.value)
        nl = text.count(b"\n")
        if head_left > 0:  # the fixture's rounds are the first lines of the job
            cut, pos = 0, -1
            while cut < head_left:
                pos = text.find(b"\n", pos + 1)
                if pos < 0:
                    break
                cut += 1
            sha_head.update(text[:pos + 1] if pos >= 0 else text)
            head_left -= cut
        if sample_lines < 200000 and rounds % 7 == 0:  # ground truth on rounds spread over the job
            sample.append(text)
            sample_lines += nl
        lines += nl
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = pipe.stats_total()
    pipe.reset()
    pipe.close()
    nt = max(1.0, tot.get("timed_rounds", 0.0))
    return {"workload": "downpore overlap with the command's default k = 10 on the same reads (dense seeds: every read indexed in every round), whole job, %d executor slots" % args.slots,
            "value": lines / dt if dt > 0 else 0.0, "unit": "overlaps/s", "wall_s": dt, "setup_s": t_init, "rounds": rounds, "paf_lines": lines,
            "ms_per_round": 1e3 * (dt - t_init) / max(1, rounds),
            "kernel_ms_per_round": {kk: tot.get(kk, 0.0) / nt for kk in ("k_count_ms", "k_write_ms", "k_index_ms", "k_query_ms", "k_chain_ms", "k_cons_ms")},
            "kernels_per_round": kernels_table(tot, rounds, tot.get("timed_rounds", 0.0)),
            "hbm_traffic_per_launch": k10_traffic(),
            "query_kernel_GBs": (tot.get("query_bytes", 0.0) / max(1, rounds) / 1e9) / (tot.get("k_query_ms", 0.0) / nt / 1e3) if tot.get("k_query_ms", 0.0) > 0 else 0.0,
            "parity": {"first_16_rounds_match_oracle_fixture": bool(fixture is not None and head_left == 0 and sha_head.hexdigest() == fixture["paf_sha256"]) if fixture else None},
            "ground_truth": ground_truth(b"".join(sample), off, truth_starts, truth_strands, 10) if sample else None}


def map_config3_leg(repeats, cpu=True):
    """BASELINE config 3 - the whole `downpore map` command (commands/map.go:33-116) on the GPU: 50 000 reads x 8 kb (10 % error)
    against a 4.6 Mb circular reference, k = 11, from the reads in host memory to the last PAF line (reference k-mer table,
    AddSingleSeeds, reference index, upload + packing of both strands, window scans, index query + chaining, mapper control
    flow, text).  PAF against the oracle's fixture (tests/golden_full/config3_map.json)."""
    from tools.synth import gen_genome, gen_reads_truth
    from tools.truth import map_truth
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden_full", "config3_map.json")))
    except Exception:
        return None
    gen = g["generator"]
    genome = np.frombuffer(gen_genome(gen["seed"], gen["genome"]), dtype=np.uint8)
    goff = np.array([0, gen["genome"]], dtype=np.int64)
    bases, off, t_starts, t_strands = gen_reads_truth(gen["seed"], gen["genome"], gen["reads"], gen["read_len"], gen["error"], False)
    ref = Reads(genome, goff, min_len=0, himem=False)
    reads = Reads(bases, off, min_len=500, himem=False)
    best, runs, ok, st, truth, all_st = None, [], True, None, None, []
    for _ in range(repeats):
        t0 = time.perf_counter()
        paf, err, st_ = map_reads(ref, reads, circular=True, k=g["k"])
        dt = time.perf_counter() - t0
        runs.append(dt)
        ok = ok and paf.count("\n") == g["paf_lines"] and hashlib.sha256(paf.encode()).hexdigest() == g["paf_sha256"] and err == g["stderr"]
        if best is None or dt < best:
            best = dt
        all_st.append((dt, st_))
        if truth is None:  # the mappings held to where the generator took the reads from (the reference's own quality figures: README.md:220-237)
            truth = map_truth(paf, off, t_starts, t_strands, gen["genome"])
        del paf
    n = gen["reads"]
    total_bases = float(off[-1])
    # the figure is the MEDIAN of the runs after the process's first (which pays for the library's first pinned and device blocks);
    # best and every run are listed beside it
    timed = sorted(all_st[1:] if len(all_st) > 1 else all_st, key=lambda x: x[0])
    median, st = timed[len(timed) // 2]
    cpu_wanted_flag, cpu = cpu_wanted(cpu), None
    if cpu_wanted_flag:
        from tests import oracle_lib as O
        t0 = time.perf_counter()
        want, werr = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=500, himem=False), circular=True, k=g["k"])
        dtc = time.perf_counter() - t0
        cpu = {"value": n / dtc, "unit": "reads/s", "cores": 1, "kind": "port", "wall_s": dtc,
               "sample": "the whole config-3 command (all %d reads) through the oracle's mapper on one host core" % n,
               "paf_identical": bool(hashlib.sha256(want.encode()).hexdigest() == g["paf_sha256"])}
        del want
    # roofline of the device part ("Map"): per window its packed bases scanned on both strands + the index query's posting
    # words, the prefilter's set words and the chained pairs' segments against the reference index, over the time of the kernels that
    # move them (window scans + query_kernel + map_kernel, HIP events summed over the run's launches)
    k_ms = st["k_map_ms"] + st["k_scan_ms"]
    alg = st.get("map_bytes", 0.0) + st.get("scan_bytes", 0.0)
    return {"workload": "downpore map, BASELINE config 3: %d reads x %d bp (error %.2f) against a %d bp circular reference, k=%d; whole "
                        "command from reads in host memory to the last PAF line" % (n, gen["read_len"], gen["error"], gen["genome"], g["k"]),
            "value": n / median, "unit": "reads/s", "value_is": "median of the %d runs after the first" % len(timed), "wall_s_median": median,
            "wall_s_best": best, "best_reads_per_s": n / best, "wall_s_runs": runs, "read_bases_per_s": total_bases / median,
            "paf_sha256_matches_oracle_fixture": bool(ok), "fixture": g["case"],
            "ground_truth": truth, "cpu_baseline": cpu,
            "roofline": {"bound": "hbm", "kernels": "scan_kernel (window scans) + query_kernel + map_kernel", "achieved": (alg / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ((alg / 1e9) / (k_ms / 1e3)) / HBM_PEAK_GBS if k_ms > 0 else 0.0, "traffic": None,
                         "algorithmic_bytes_per_run": alg, "algorithmic_bytes_map_kernels": st.get("map_bytes", 0.0),
                         "algorithmic_bytes_window_scans": st.get("scan_bytes", 0.0), "kernel_ms_per_run": k_ms,
                         "algorithmic_bytes_per_window": alg / max(1.0, st["n_windows"])},
            "map_kernel": {"kernel": "map_kernel (A19 + A20: prefilter + SeedSequence.Match of every window pair) + query_kernel", "ms_total": st["k_map_ms"],
                           "launches": st["n_batches"], "windows": st["n_windows"], "chains": st["n_chains"],
                           "windows_per_s_in_kernel": st["n_windows"] / (st["k_map_ms"] / 1e3) if st["k_map_ms"] > 0 else 0.0},
            "scan_kernels_ms_total": st["k_scan_ms"],
            "breakdown_s": {"setup_reference_index_upload": st["t_setup_s"], "window_scans": st["t_scan_s"], "index_query_chaining": st["t_chain_s"],
                            "mapper_control_flow_and_text": st["t_host_s"]}}


def cpu_wanted(flag):
    return bool(flag)


def golden_fixture(args):
    """tests/golden_full/config2.json (tools/make_golden_full.py: the oracle alone, every round) when the workload is config 2."""
    path = os.path.join(ROOT, "tests", "golden_full", "config2.json")
    try:
        g = json.load(open(path))
    except Exception:
        return None
    gen = g["generator"]
    same = (gen["seed"] == args.seed and gen["reads"] == args.reads and gen["read_len"] == args.read_len and
            gen["error"] == args.error and g["k"] == args.k and args.seed_batch_size == 10000)
    return g if same else None


def host_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "cpu_quota_cores": cpu_budget()}


def cpu_budget():
    """CPUs this container may use: the cgroup quota when there is one, else the CPU count."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(bases, off, args, values, threads=1, check_against=None):
    """The oracle (a quirk-exact C++ port of the reference's CPU algorithm incl. its two-pass scan over a 4^k-byte
    table) timed on the host for the first rounds of the same workload; threads > 1 spreads the per-read scans (the
    part the reference parallelises with num_workers) over that many threads."""
    from tests import oracle_lib as O
    O.build_oracle()
    os.environ["DPO_SCAN_THREADS"] = str(threads)
    rs = O.ReadSet(bases, off, min_len=1000)
    t0 = time.perf_counter()
    run = O.OverlapRun(rs, k=args.k, seed_batch_size=args.seed_batch_size, values=np.ascontiguousarray(values),
                       max_rounds=args.cpu_rounds, traces=False)
    dt = time.perf_counter() - t0
    lines = run.paf.count("\n")
    res = {}
    if check_against is not None:
        # the oracle doubles as the checker at the full BASELINE size: its first rounds must be the exact prefix of what
        # the GPU pipeline printed for the same rounds
        res["paf_identical_to_gpu_rounds"] = bool(check_against.startswith(run.paf)) and lines > 0
    return {**res, "value": lines / dt if dt > 0 else 0.0, "unit": "overlaps/s", "cores": threads, "kind": "port",
            "sample": "first %d rounds of the same workload (value table supplied), %.1f s, %d PAF lines" % (run.rounds, dt, lines),
            "ms_per_step": 1e3 * dt / max(1, run.rounds)}


if __name__ == "__main__":
    main()
