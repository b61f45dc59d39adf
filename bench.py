#!/usr/bin/env python3
"""bench.py — overlaps/sec of the all-vs-all overlap hot path on synthetic long reads (BASELINE.json metric).

A "step" is one round of the overlap command (commands/overlap.go:115-194): seed selection for the next query batch,
the seed occurrences of every non-ignored read (from the resident k-mer position index at this size, by the scan kernels
below 1 Gbase or with DP_SCAN_INDEX=0), survivor exchange (N>1), index build, index query + chaining, consensus, PAF.
`roofline` describes the dominant kernel of the mode that ran; a second, shorter leg times the other mode.
Workload (N=1 default): BASELINE config 2 — 100 000 synthetic reads x 10 kb, genome 50 Mb (20x), k=13, error-free
(the error-free set is the throughput default at k=13).  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--error", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--seed-batch-size", type=int, default=10000)
    ap.add_argument("--cpu-rounds", type=int, default=2, help="oracle rounds timed for cpu_baseline (0 = skip)")
    ap.add_argument("--slots", type=int, default=6, help="rounds executed concurrently per GPU (executor slots)")
    ap.add_argument("--index-steps", type=int, default=100,
                    help="N=1: after the timed region, time this many further rounds in the other scan mode (index <-> scan "
                         "kernels; reported as scan_kernels_leg / index_mode; 0 = skip)")
    ap.add_argument("--mode", default="round", choices=["round", "round-batch", "scan-shard"], help="multi-GPU decomposition (N > 1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and "DP_HOST_THREADS" not in os.environ:
        # one process per GPU on ONE host: the ranks share the container's CPU quota, so each gets its share of worker threads
        os.environ["DP_HOST_THREADS"] = str(max(2, cpu_budget() // world - (1 if cpu_budget() // world > 3 else 0)))
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
    G = N * L // 20
    t0 = time.time()
    bases, off = gen_reads(args.seed, G, N, L, args.error, False)
    reads = Reads(bases, off, min_len=1000)
    t_gen = time.time() - t0
    t0 = time.time()
    pipe = OverlapPipeline(reads, device=local_rank, k=args.k, seed_batch_size=args.seed_batch_size, rank=rank, world=world,
                           torch_device=torch_device, mode=args.mode, slots=args.slots)
    t_setup = time.time() - t0

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    warm = 0
    while warm < args.warmup:
        c = pipe.step()
        if c == 0:
            break
        warm += c
    pipe.drain()  # rounds the executor pipeline has in flight are thrown away: the timed region starts from an empty pipeline
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

    cs0 = cpu_stat()
    acc = {}
    lines = 0
    steps_done = 0   # rounds committed in the timed region (a step = one round; with N ranks a call commits up to N)
    samples = 0
    t_start = time.perf_counter()
    while steps_done < args.steps:
        c = pipe.step()
        if c == 0:
            break
        st = pipe.stats()  # stats of the last committed round
        for key, v in st.items():
            acc[key] = acc.get(key, 0.0) + v
        samples += 1
        lines += pipe.step_lines()
        steps_done += c
    sync()
    elapsed = time.perf_counter() - t_start
    cs1 = cpu_stat()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch_device if torch_device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Same job continued in the OTHER scan mode (dp_scan_reads answers either from the resident k-mer position index,
    # dp_kindex.hip - the default from 1 Gbase up - or by streaming the packed reads through scan_kernel).  Reported next
    # to the headline number, never instead of it.
    main_index = bool(acc.get("idx_rounds"))
    alt = None
    if world == 1 and args.index_steps > 0 and "DP_SCAN_INDEX" not in os.environ:
        pipe.drain()
        os.environ["DP_SCAN_INDEX"] = "0" if main_index else "1"
        tb0 = time.perf_counter()
        got = pipe.step()  # (index mode: builds the index, one-off, and runs a round)
        tb1 = time.perf_counter()
        w2 = got
        while got and w2 < args.warmup:
            got = pipe.step()
            w2 += got
        pipe.drain()
        sync()
        iacc, ilines, isteps, isamples = {}, 0, 0, 0
        ti0 = time.perf_counter()
        while got and isteps < args.index_steps:
            got = pipe.step()
            if got == 0:
                break
            for key, v in pipe.stats().items():
                iacc[key] = iacc.get(key, 0.0) + v
            isamples += 1
            ilines += pipe.step_lines()
            isteps += got
        sync()
        iel = time.perf_counter() - ti0
        del os.environ["DP_SCAN_INDEX"]
        pipe.drain()
        if isteps:
            m = max(1, isamples)
            alt = {"value": ilines / iel, "unit": "overlaps/s", "steps": isteps, "ms_per_step": 1e3 * iel / isteps,
                   "kernel_ms_per_step": {kk: iacc.get(kk, 0.0) / m for kk in ("k_count_ms", "k_write_ms", "k_query_ms", "k_chain_ms")},
                   "count_step_algorithmic_bytes": iacc.get("count_bytes", 0.0) / m}
            if main_index:  # the alternative leg streamed the reads: its count pass is the HBM-streaming kernel of the path
                cms = iacc.get("k_count_ms", 0.0) / m
                cb = iacc.get("count_bytes", 0.0) / m
                alt["scan_count_pass"] = {"kernel": "scan_kernel<0>", "launch_ms": cms, "algorithmic_bytes_per_launch": cb,
                                          "achieved_GBs": (cb / 1e9) / (cms / 1e3) if cms > 0 else 0.0,
                                          "frac_of_hbm_peak": ((cb / 1e9) / (cms / 1e3)) / HBM_PEAK_GBS if cms > 0 else 0.0}
            else:
                alt.update({"first_round_incl_index_build_s": tb1 - tb0,
                            "rounds_served_by_index": iacc.get("idx_rounds", 0.0),
                            "seed_occurrences_per_step": iacc.get("idx_hits", 0.0) / m,
                            "resident_bytes": 8 * int(reads.total_bases()) + 8 * (4 ** args.k + 1),
                            "note": "no kernel of this mode streams the reads"})

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
        n = max(1, samples)
        count_ms = acc.get("k_count_ms", 0.0) / n
        count_bytes = acc.get("count_bytes", 0.0) / n
        chain_ms = acc.get("k_chain_ms", 0.0) / n
        chain_bytes = acc.get("chain_bytes", 0.0) / n

        def pmc_traffic(name):
            tpath = os.path.join(ROOT, "profiles", name)
            try:
                return json.load(open(tpath)).get("hbm_bytes_per_launch") if os.path.exists(tpath) else None
            except Exception:
                return None

        if main_index:
            # no kernel of the run streams the reads; the dominant kernel (by time, in the run and in the rocprofv3 trace) is
            # chain_kernel: prefilter + seed chaining of every (query, candidate) pair, one wave per query, sequential over its
            # candidates because of the minMatches ratchet (overlap.go:380-382) - latency-bound, priced against HBM all the same
            rl_kernel = ("chain_kernel (A5 prefilter + A6/A7/A8 chaining; latency-bound: one wave walks a query's candidates in "
                         "order) - the run used the resident k-mer position index, no kernel streams the reads")
            rl_bytes, rl_ms, traffic = chain_bytes, chain_ms, pmc_traffic("chain_traffic.json")
        else:
            rl_kernel = "scan_kernel<0> (count pass of the packed k-mer scan, A2/A10)"
            rl_bytes, rl_ms, traffic = count_bytes, count_ms, pmc_traffic("scan_traffic.json")
        achieved = (rl_bytes / 1e9) / (rl_ms / 1e3) if rl_ms > 0 else 0.0
        out = {
            "metric": "overlaps/sec (all-vs-all PAF)", "value": lines / elapsed if elapsed > 0 else 0.0, "unit": "overlaps/s",
            "n_gpus": world, "steps": steps_done, "warmup": warm, "ms_per_step": 1e3 * elapsed / max(1, steps_done),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%d synthetic reads x %d bp, genome %d bp (20x), error %.3g, k=%d, overlap rounds "
                                   "(BASELINE config 2)" % (N, L, G, args.error, args.k),
                       "reads": N, "read_len": L, "k": args.k, "seed_batch_size": args.seed_batch_size, "executor_slots_per_gpu": args.slots,
                       "parallelism": ("single GPU" if world == 1 else
                                       "round-parallel over %d GPUs: rank r's executor pipeline runs the rounds r, r+N, ...; one round "
                                       "per rank all-gathered (RCCL) per superstep and committed in order" % world if args.mode.startswith("round") else
                                       "scan sharded by read over %d GPUs, survivors all-gathered (RCCL)" % world)},
            "roofline": {"bound": "hbm", "kernel": rl_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": rl_bytes, "launch_ms": rl_ms,
                         "measured_stream_GBs": stream_gbs,
                         "frac_of_measured_stream": (achieved / stream_gbs) if stream_gbs else None},
            # the other kernels of a round, same convention (algorithmic bytes / HIP-event time); chain_kernel is a
            # latency-bound per-query state machine, its byte rate is reported for completeness only
            "other_kernels": {
                ("index_counting_step" if main_index else "scan_count_pass"): {
                    "ms": count_ms, "algorithmic_bytes": count_bytes,
                    "GBs": (count_bytes / 1e9) / (count_ms / 1e3) if count_ms > 0 else 0.0},
                ("index_write" if main_index else "scan_write_pass"): {
                    "ms": acc.get("k_write_ms", 0.0) / n,
                    "algorithmic_bytes": (acc.get("scan_bytes", 0.0) - acc.get("count_bytes", 0.0)) / n},
                "index_query": {"ms": acc.get("k_query_ms", 0.0) / n, "algorithmic_bytes": acc.get("query_bytes", 0.0) / n,
                                "GBs": (acc.get("query_bytes", 0.0) / 1e9) / (acc.get("k_query_ms", 0.0) / 1e3) if acc.get("k_query_ms", 0) > 0 else 0.0},
                "chain_kernel": {"ms": chain_ms, "bound": "latency (sequential ratchet per query)", "algorithmic_bytes": chain_bytes,
                                 "GBs": (chain_bytes / 1e9) / (chain_ms / 1e3) if chain_ms > 0 else 0.0,
                                 "matches_per_step": acc.get("n_matches", 0.0) / n},
            },
            "scan_mode": "resident k-mer position index" if main_index else "scan kernels",
            ("scan_kernels_leg" if main_index else "index_mode"): alt,
            "paf_lines": lines, "rounds_per_s": steps_done / elapsed if elapsed > 0 else 0.0,
            "reads_scanned_per_s": (acc.get("scan_items", 0.0) / n) * steps_done / elapsed if elapsed > 0 else 0.0,
            "phase_ms_per_step": {kk: 1e3 * acc.get(kk, 0.0) / n for kk in ("t_prepare", "t_scan", "t_index", "t_query", "t_consensus")},
            "kernel_ms_per_step": {kk: acc.get(kk, 0.0) / n for kk in ("k_count_ms", "k_write_ms", "k_scan_ms", "k_query_ms", "k_chain_ms", "k_cons_ms")},
            "index_query": {"bytes_per_step": acc.get("query_bytes", 0.0) / n,
                            "achieved_GBs": (acc.get("query_bytes", 0.0) / 1e9) / (acc.get("k_query_ms", 1e-9) / 1e3) if acc.get("k_query_ms", 0) > 0 else 0.0},
            "setup_s": {"generate": t_gen, "upload_pack_histogram_values": t_setup},
            # host side of the timed region: CPU seconds used by this container and time it spent throttled by its CPU quota
            "host": host_info(),
            "host_cpu": {"cpu_s": (cs1.get("usage_usec", 0) - cs0.get("usage_usec", 0)) / 1e6,
                         "throttled_s": (cs1.get("throttled_usec", 0) - cs0.get("throttled_usec", 0)) / 1e6,
                         "wall_s": elapsed},
        }
        if world == 1 and args.cpu_rounds > 0:
            vals = pipe.values()
            gpu_paf = pipe.all_paf()  # every committed round so far, in round order
            out["cpu_baseline"] = cpu_baseline(bases, off, args, vals, threads=1, check_against=gpu_paf)
            # the same port with its per-read scans spread over the host cores this container may use
            out["cpu_baseline_all_cores"] = cpu_baseline(bases, off, args, vals, threads=cpu_budget())
        print(json.dumps(out))
    pipe.close()
    if dist is not None:
        dist.destroy_process_group()


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
