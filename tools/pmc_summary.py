#!/usr/bin/env python3
"""Per-kernel means of the round-2 PMC passes (tools/gpu_profile_%s.sh" % ROUND + ") -> profiles/r02/pmc_<workload>.json and the
per-round traffic files bench.py quotes (profiles/chain_traffic.json, query_traffic.json, kindex_traffic.json,
scan_traffic.json, dense_query_traffic.json).

gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced streaming read -> x2 (MI355X_MICROARCH.md, HBM
section); WRITE_SIZE is exact for streaming stores.  Both come out of rocprofv3 in KiB.  The x2 is calibrated on kernels with
a known byte count (pack_kernel reads the ASCII reads once; kb_part2 reads and writes the 8-byte entry array once).
Kernels with scattered narrow reads are uncalibrated: for them the corrected figure is an upper bound."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r06")
R = os.path.join(ROOT, "gpurun_out", ROUND)
OUT = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(OUT, exist_ok=True)
SQ = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY",
      "SQ_WAIT_INST_LDS"]


def short(name):
    n = name.split("(")[0]
    for pre in ("void ",):
        if n.startswith(pre):
            n = n[len(pre):]
    if "rocprim" in n:
        n = "rocprim::" + n.split("::")[-1][:60]
    n = n.strip()
    if n.startswith("dp_kernel<"):  # round 6: per-round kernels are dp_kernel<kernel> (dp_launch.h): the kernel's own name
        n = n[len("dp_kernel<"):].rstrip()
        if n.endswith(">"):
            n = n[:-1].rstrip()
    if n.startswith("dp_multi<"):  # per-round kernels are dp_multi<kernel, N> (N = rounds the launch can carry): the kernel's own name
        n = n[len("dp_multi<"):]
        n = n[:n.rfind(",")] if "," in n else n.rstrip(">")
    return n


def load(run):
    files = sorted(glob.glob(os.path.join(R, "pmc_" + run, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not files:
        return None
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def summarize(workload):
    out = {}
    for suffix in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
        acc = load("%s_%s" % (workload, suffix))
        if acc is None:
            print("missing pass:", workload, suffix)
            continue
        for kern, ctrs in acc.items():
            d = out.setdefault(kern, {})
            for c, v in ctrs.items():
                d[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "sum": sum(v)}
    for kern, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024
        if "SQ_INSTS_LDS" in d and d["SQ_INSTS_LDS"]["sum"] > 0:
            d["lds_bank_conflict_cycles_per_lds_inst"] = d["SQ_LDS_BANK_CONFLICT"]["sum"] / d["SQ_INSTS_LDS"]["sum"]
        if "SQ_ACTIVE_INST_LDS" in d and d["SQ_ACTIVE_INST_LDS"]["sum"] > 0:
            d["lds_bank_conflict_frac_of_lds_active"] = d["SQ_LDS_BANK_CONFLICT"]["sum"] / d["SQ_ACTIVE_INST_LDS"]["sum"]
    return out


NOTE = ("rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ set> --kernel-trace, one pass each (tools/gpu_profile_%s.sh" % ROUND + "); means per dispatch; "
        "FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction, MI355X_MICROARCH.md)")
WORK = {"main": "bench.py --steps 1 --warmup 0 --max-rounds 40 (config 2, resident k-mer position index; set-up kernels at full size)",
        "scan": "DP_SCAN_INDEX=0 bench.py --steps 1 --warmup 0 --max-rounds 40 (config 2, scan kernels)",
        "dense": "bench.py --k 10 --steps 1 --warmup 0 --max-rounds 6 (same reads, dense seeds)"}
allw = {}
for w in WORK:
    s = summarize(w)
    if not s:
        continue
    allw[w] = s
    json.dump({"workload": WORK[w], "note": NOTE, "kernels": s}, open(os.path.join(OUT, "pmc_%s.json" % w), "w"), indent=1, sort_keys=True)
    print("profiles/%s/pmc_%s.json: %d kernels" % (ROUND, w, len(s)))
    if w == "dense":  # (bench.py's k = 10 job quotes this file per kernel: overlap_default_k10_job.hbm_traffic_per_launch)
        json.dump({"workload": WORK[w], "note": NOTE, "kernels": s}, open(os.path.join(OUT, "pmc_k10.json"), "w"), indent=1, sort_keys=True)


def per_round(s, kernels, rounds_of):
    """HBM bytes per round of a group of kernels = sum over the group of (bytes per launch x launches) / rounds"""
    nr = s[rounds_of]["FETCH_SIZE"]["dispatches"]
    tot = 0.0
    parts = {}
    for k in kernels:
        for name, d in s.items():
            if name == k or name.startswith(k + "<"):
                if "hbm_bytes_per_launch" not in d:
                    continue
                b = d["hbm_bytes_per_launch"] * d["FETCH_SIZE"]["dispatches"]
                parts[name] = {"launches_per_round": d["FETCH_SIZE"]["dispatches"] / nr, "hbm_bytes_per_round": b / nr}
                tot += b
    return tot / nr, parts, nr


def traffic(fn, workload, desc, kernels, rounds_of):
    s = allw.get(workload)
    if not s or rounds_of not in s:
        print("no data for", fn)
        return
    b, parts, nr = per_round(s, kernels, rounds_of)
    json.dump({"kernel": desc, "hbm_bytes_per_launch": b, "per": "round (one launch of each kernel of the group, passes included)", "rounds": nr,
               "parts": parts, "workload": WORK[workload], "source": NOTE}, open(os.path.join(ROOT, "profiles", fn), "w"), indent=1, sort_keys=True)
    print(fn, "%.3f MB per round over %d rounds" % (b / 1e6, nr))


CHAIN = ["pair_scan_kernel", "chain_walk_kernel", "chain_spec_kernel", "chain_resolve_kernel", "match_anchor_kernel"]
traffic("chain_traffic.json", "main", "chaining stage (pair_scan + chain_walk + chain_spec + chain_resolve + match_anchor)", CHAIN, "pair_scan_kernel")
traffic("query_traffic.json", "main", "query_kernel<false> (+ <true> where launched)", ["query_kernel"], "pair_scan_kernel")
traffic("kindex_traffic.json", "main", "index-mode counting step (kidx_prepare + kidx_walk_bin + kidx_bin_count [round 4: kidx_walk<false>] + kidx_offsets)",
        ["kidx_prepare", "kidx_walk_bin", "kidx_bin_count", "kidx_walk<false>", "kidx_offsets"], "kidx_offsets")
traffic("kindex_write_traffic.json", "main", "index-mode write step (kidx_bin_fill [round 4: kidx_fill_rec] / kidx_walk<true> + kidx_sortwrite)",
        ["kidx_bin_fill", "kidx_fill_rec", "kidx_walk<true>", "kidx_sortwrite"], "kidx_offsets")
traffic("dense_kindex_traffic.json", "dense", "index-mode counting + write steps at k = 10 (all kidx_* kernels)",
        ["kidx_prepare", "kidx_walk_bin", "kidx_bin_count", "kidx_walk<false>", "kidx_offsets", "kidx_bin_fill", "kidx_fill_rec", "kidx_walk<true>", "kidx_sortwrite",
         "kidx_bin_sort_dense"], "kidx_offsets")
INDEX = ["chunk_kernel", "index_fill_kernel", "index_fill_rows_kernel", "posting_transpose_kernel", "posting_meta_kernel", "zero_regions_kernel"]
traffic("index_build_traffic.json", "main", "index build of a round (chunk + index_fill(_rows) + posting_transpose + posting_meta + zero_regions)", INDEX, "pair_scan_kernel")
traffic("dense_index_build_traffic.json", "dense", "index build of a round at k = 10 (chunk + index_fill_rows + posting_transpose + posting_meta + zero_regions)", INDEX, "pair_scan_kernel")
traffic("dense_chain_traffic.json", "dense", "chaining stage at k = 10", CHAIN, "pair_scan_kernel")
traffic("consensus_traffic.json", "main", "consensus_full_kernel (all layouts)", ["consensus_full_kernel"], "pair_scan_kernel")
traffic("scan_traffic.json", "scan", "scan_kernel<0, 2> (count pass)", ["scan_kernel<0, 2>"], "scan_kernel<0, 2>")
traffic("dense_query_traffic.json", "dense", "query_kernel<false>, k=10", ["query_kernel"], "pair_scan_kernel")
if "main" in allw:
    for k in ("pack_kernel", "kb_part1", "kb_part2", "kb_final", "kb_count1", "kb_count2"):
        d = allw["main"].get(k)
        if d and "hbm_bytes_per_launch" in d:
            print("calibration %-10s FETCH %.1f MiB  WRITE %.1f MiB  corrected total %.1f MB" %
                  (k, d["FETCH_SIZE"]["mean"] / 1024, d["WRITE_SIZE"]["mean"] / 1024, d["hbm_bytes_per_launch"] / 1e6))
