"""Python driver of the product's overlap pipeline (libdownpore_host.so) + the multi-GPU survivor exchange.

One process per GPU.  Each rank keeps the whole read set resident and scans only its contiguous shard of reads per
round (the scan is >95 % of the bytes); the survivors (reads with >= num_seeds seed hits, a few MB at
k=13) are all-gathered in rank order — ranks own ascending contiguous read ranges, so the concatenation is file order —
and every rank then builds the identical (small) index and produces the identical PAF.
"""
import ctypes as C
import os

import numpy as np

from .hip import DpError, lib_path, load_library

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None

STAT_FIELDS = ["t_prepare", "t_scan", "t_index", "t_query", "t_consensus", "k_scan_ms", "k_query_ms", "k_chain_ms",
               "scan_bases", "scan_items", "scan_bytes", "query_bytes", "n_queries", "n_indexed", "n_hits", "n_matches",
               "n_paf", "n_seeds", "round", "bad_back", "empty_match", "k_count_ms", "k_write_ms", "count_bytes", "k_cons_ms", "idx_rounds", "idx_hits", "chain_bytes", "timed_rounds", "gang_members", "cons_bytes", "k_index_ms"]


def host_lib_path():
    # DPH_HOST_LIB: another build of the same library (the ThreadSanitizer build of `make host-tsan`, used by the CPU planner tests)
    from .hip import lib_dir
    return os.environ.get("DPH_HOST_LIB") or os.path.join(lib_dir(), "libdownpore_host.so")


def load_host():
    global _host
    if _host is not None:
        return _host
    load_library()  # the HIP library must exist; no fallback
    p = host_lib_path()
    if not os.path.exists(p):
        raise DpError("libdownpore_host.so is not built (%s): run __graft_entry__.build()" % p)
    H = C.CDLL(p)
    vp = C.c_void_p
    H.dph_last_error.restype = C.c_char_p
    H.dph_last_error.argtypes = [vp]
    H.dph_reads_from_arrays.restype = vp
    H.dph_reads_from_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    H.dph_reads_from_arrays_q.restype = vp
    H.dph_reads_from_arrays_q.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]
    H.dph_reads_from_fasta.restype = vp
    H.dph_reads_from_fasta.argtypes = [C.c_char_p, C.c_int64, C.c_int]
    H.dph_reads_free.argtypes = [vp]
    H.dph_reads_count.restype = C.c_int64
    H.dph_reads_count.argtypes = [vp]
    H.dph_reads_total_bases.restype = C.c_int64
    H.dph_reads_total_bases.argtypes = [vp]
    H.dph_reads_get_ignore.argtypes = [vp, C.c_void_p]
    H.dph_reads_reset_ignore.argtypes = [vp]
    H.dph_overlap_create.restype = vp
    H.dph_overlap_create.argtypes = [vp, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
    H.dph_overlap_destroy.argtypes = [vp]
    H.dph_overlap_open.restype = vp
    H.dph_overlap_open.argtypes = [vp, C.c_int]
    H.dph_overlap_init.argtypes = [vp, C.c_void_p, C.c_double, C.c_void_p]
    H.dph_overlap_reset.argtypes = [vp]
    H.dph_overlap_setup_times.restype = None
    H.dph_overlap_setup_times.argtypes = [vp, C.c_void_p]
    H.dph_overlap_stats_total.argtypes = [vp, C.c_void_p]
    H.dph_overlap_set_shard.argtypes = [vp, C.c_int64, C.c_int64]
    H.dph_overlap_values.restype = C.POINTER(C.c_double)
    H.dph_overlap_values.argtypes = [vp, C.POINTER(C.c_int64)]
    H.dph_overlap_round_scan.argtypes = [vp]
    H.dph_overlap_local.argtypes = [vp] + [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(C.c_uint64)] * 2
    H.dph_overlap_round_finish.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    for f in (H.dph_overlap_round_paf, H.dph_overlap_all_paf, H.dph_overlap_errtext):
        f.restype = C.POINTER(C.c_char)
        f.argtypes = [vp, C.POINTER(C.c_int64)]
    H.dph_overlap_stats.argtypes = [vp, C.c_void_p]
    H.dph_overlap_ctx.restype = vp
    H.dph_overlap_ctx.argtypes = [vp]
    H.dph_overlap_step.argtypes = [vp]
    H.dph_overlap_drain.restype = None
    H.dph_overlap_drain.argtypes = [vp]
    H.dph_overlap_round.restype = C.c_int64
    H.dph_overlap_round.argtypes = [vp]
    H.dph_overlap_exec_round.restype = C.POINTER(C.c_uint8)
    H.dph_overlap_exec_round.argtypes = [vp, C.c_int64, C.POINTER(C.c_uint64)]
    H.dph_overlap_commit_blobs.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_int]
    H.dph_overlap_done.argtypes = [vp]
    H.dph_overlap_keep_text.restype = None
    H.dph_overlap_keep_text.argtypes = [vp, C.c_int]
    H.dph_comm_unique_id.argtypes = [C.c_void_p]
    H.dph_overlap_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    H.dph_overlap_comm_init_local.argtypes = [C.c_void_p, C.c_int]
    H.dph_overlap_round_sharded.argtypes = [vp]
    H.dph_overlap_rounds_sharded.argtypes = [vp]
    H.dph_overlap_comm_init_slots.argtypes = [vp, C.c_int, C.c_int, C.c_void_p, C.c_int]
    H.dph_overlap_comm_init_local_slots.argtypes = [C.c_void_p, C.c_int, C.c_int]
    H.dph_overlap_superstep.argtypes = [vp, C.c_int]
    _host = H
    return H


def shard_bounds(n_reads, rank, world):
    """Ascending contiguous read ranges, one per rank (SURVEY §8(e) ordering caveat)."""
    per = (n_reads + world - 1) // world
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


def allgather_survivors(local, world, device=None):
    """All-gather of the variable-size survivor lists with torch.distributed (backend nccl == RCCL over xGMI on the
    GPU box; gloo in the CPU tests).  `local` = dict(read u32[n], n_seeds u32[n], segs i32[m]).  Returns the rank-ordered
    concatenation."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    sizes = torch.tensor([len(local["read"]), len(local["segs"])], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    max_n, max_m = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    # one padded int32 payload per rank: [read ids | n_seeds | segs]
    pay = np.zeros(2 * max_n + max_m, dtype=np.int32)
    n, m = len(local["read"]), len(local["segs"])
    pay[:n] = local["read"].astype(np.int32)
    pay[max_n:max_n + n] = local["n_seeds"].astype(np.int32)
    pay[2 * max_n:2 * max_n + m] = local["segs"]
    t = torch.from_numpy(pay).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    reads, nseeds, segs = [], [], []
    for r in range(world):
        o = out[r].cpu().numpy()
        rn, rm = int(all_sizes[r, 0]), int(all_sizes[r, 1])
        reads.append(o[:rn].astype(np.uint32))
        nseeds.append(o[max_n:max_n + rn].astype(np.uint32))
        segs.append(o[2 * max_n:2 * max_n + rm])
    return dict(read=np.concatenate(reads), n_seeds=np.concatenate(nseeds), segs=np.concatenate(segs).astype(np.int32))


_FLAT_GATHER = {}


def allgather_blobs(blob, world, device=None):
    """All-gather of one variable-size byte string per rank (torch.distributed; nccl == RCCL on the GPU box).  Returns
    (cat, sizes): the ranks' strings back to back in one uint8 array and their lengths (uint64).  One host synchronisation
    for the sizes, one flat collective and one copy back for the payload."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    mine = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
    if world == 1 and not dist.is_initialized():
        return mine.copy(), np.array([len(mine)], dtype=np.uint64)
    n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    key = str(dev.type)
    flat = _FLAT_GATHER.get(key, True)
    sz = torch.empty(world, dtype=torch.int64, device=dev)
    if flat:
        try:
            dist.all_gather_into_tensor(sz, n)
        except (RuntimeError, NotImplementedError):  # a backend without the flat form
            flat = _FLAT_GATHER[key] = False
    if not flat:
        parts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, n)
        sz = torch.cat(parts)
    sizes = np.array(sz.tolist(), dtype=np.uint64)
    mx = (max(int(sizes.max()), 1) + 4095) // 4096 * 4096
    buf = np.zeros(mx, dtype=np.uint8)
    buf[:len(mine)] = mine
    t = torch.from_numpy(buf).to(dev)
    if flat:
        out = torch.empty(world * mx, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out, t)
        host = out.cpu().numpy().reshape(world, mx)
    else:
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        host = torch.stack(outs).cpu().numpy()
    cat = np.concatenate([host[r, :int(sizes[r])] for r in range(world)]) if world > 1 else np.ascontiguousarray(host[0, :int(sizes[0])])
    return cat, sizes


def allgather_bytes(blob, world, device=None):
    """The same as a list of byte strings, one per rank."""
    cat, sizes = allgather_blobs(blob, world, device)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return [cat[off[r]:off[r + 1]].tobytes() for r in range(world)]


class Reads:
    def __init__(self, bases=None, off=None, min_len=1000, himem=True, fasta=None, quals=None):
        """fasta: a FASTA or FASTQ file (one line per read).  quals: raw FASTQ quality characters at the offsets of bases."""
        H = load_host()
        if fasta is not None:
            self.h = H.dph_reads_from_fasta(fasta.encode(), min_len, 1 if himem else 0)
            if not self.h:
                raise DpError(H.dph_last_error(None).decode())
        else:
            b = np.ascontiguousarray(bases, dtype=np.uint8)
            o = np.ascontiguousarray(off, dtype=np.int64)
            if quals is not None:
                q = np.ascontiguousarray(quals, dtype=np.uint8)
                self.h = H.dph_reads_from_arrays_q(b.ctypes.data, q.ctypes.data, o.ctypes.data, len(o) - 1, min_len, 1 if himem else 0)
            else:
                self.h = H.dph_reads_from_arrays(b.ctypes.data, o.ctypes.data, len(o) - 1, min_len, 1 if himem else 0)
        self.H = H

    def __len__(self):
        return self.H.dph_reads_count(self.h)

    def total_bases(self):
        return self.H.dph_reads_total_bases(self.h)

    def ignore(self):
        out = np.zeros(len(self), dtype=np.uint8)
        self.H.dph_reads_get_ignore(self.h, out.ctypes.data)
        return out

    def reset_ignore(self):
        self.H.dph_reads_reset_ignore(self.h)


class OverlapPipeline:
    """`downpore overlap` on one GPU (or one rank of a multi-GPU job)."""

    def __init__(self, reads, device=0, k=10, overlap_size=1000, num_seeds=15, seed_batch_size=10000, chunk_size=10000,
                 query_batch_size=20000, min_hits=0.25, himem=True, values=None, rank=0, world=1, torch_device=None,
                 mode="round", slots=1, query_type=1, defer_init=False, comm=None):
        """mode (world > 1): "round" = pipelined round-parallel (rank r's executor pipeline works on the rounds
        r, r+world, ...; per superstep every rank contributes its next round, results are all-gathered and committed in
        order with the speculation check); "round-batch" = the same exchange with batch-synchronous supersteps (every rank
        executes `slots` consecutive rounds, then all wait); "scan-shard" = every rank runs every round, the scan is
        sharded by read and the survivors are all-gathered; with comm set and slots > 1 a step runs `slots` consecutive rounds
        concurrently, slot i exchanging on its own communicator, and commits them in order.
        slots: executor slots of this process = rounds it runs concurrently on its GPU (each slot has its own stream and
        per-round buffers; the resident reads are shared).
        defer_init: only create the device context and upload + pack the reads; init() then does what `downpore overlap`
        does before its first round (value table, k-mer position index, executor slots, planner) and reset() returns to
        this state, so whole jobs can be run - and timed - repeatedly on resident reads.
        comm (scan-shard): "rccl" = the survivor exchange runs inside the library over an RCCL communicator (dp_comm_init; the
        128-byte id travels through torch.distributed once); "local" = in-process peers wired with link_local(); None = the
        exchange is done here with torch.distributed on host copies (gloo tests)."""
        self.H = load_host()
        if mode == "scan-shard" and world > 1 and comm is None:
            slots = 1  # (the host-side exchange of the gloo tests is one round at a time)
        # query_type: overlap.QueryEdges=1 (the overlap command), QueryCentre=2, QueryAll=4 (the correct command), +8 WeightEdges
        p = np.array([overlap_size, k, num_seeds, seed_batch_size, chunk_size, query_batch_size,
                      (1 if himem else 0) | (query_type << 8), slots], dtype=np.int64)
        self.slots = slots
        self._values_keepalive = values
        self._params, self._min_hits = p, float(min_hits)
        self.h = self.H.dph_overlap_open(reads.h, device)
        if not self.h:
            raise DpError("dph_overlap_open: " + self.H.dph_last_error(None).decode())
        self.reads = reads
        self.rank, self.world = rank, world
        self.torch_device = torch_device
        self.mode = mode if (world > 1 or (mode in ("scan-shard", "round") and comm is not None)) else "single"
        self._lo_hi = shard_bounds(len(reads), rank, world) if self.mode == "scan-shard" else None
        self.comm = comm if self.mode in ("scan-shard", "round") else None
        if self.comm == "rccl":
            self._init_rccl()
        if not defer_init:
            self.init()

    def init(self):
        values = self._values_keepalive
        vptr = values.ctypes.data if values is not None else None
        if self.H.dph_overlap_init(self.h, self._params.ctypes.data, self._min_hits, vptr) != 0:
            raise DpError("dph_overlap_init: " + self.H.dph_last_error(self.h).decode())
        rank, world = self.rank, self.world
        if self.mode == "round":
            self.H.dph_overlap_set_ranks.restype = None
            self.H.dph_overlap_set_ranks.argtypes = [C.c_void_p, C.c_int, C.c_int]
            self.H.dph_overlap_wait_owned_many.restype = C.c_void_p
            self.H.dph_overlap_wait_owned_many.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
            self.H.dph_overlap_commit_gathered.restype = C.c_int
            self.H.dph_overlap_commit_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            self.H.dph_overlap_set_ranks(self.h, rank, world)
        if self.mode == "scan-shard":
            self.H.dph_overlap_set_shard(self.h, *self._lo_hi)

    def _init_rccl(self):
        """dp_comm_init on this rank's context: rank 0 makes the id, everybody gets it through torch.distributed."""
        ns = max(1, self.slots) if self.mode == "scan-shard" else 1  # (round-parallel: one communicator for the result exchange)
        idb = np.zeros(128 * ns, dtype=np.uint8)  # one communicator per executor slot
        if self.rank == 0:
            for i in range(ns):
                if self.H.dph_comm_unique_id(idb[128 * i:].ctypes.data) != 0:
                    raise DpError("dp_comm_unique_id failed (librccl not loadable?)")
        if self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(idb)
            if self.torch_device is not None:
                t = t.to(self.torch_device)
            dist.broadcast(t, 0)
            idb = t.cpu().numpy().copy()
        idb = np.ascontiguousarray(idb)
        if ns > 1:
            if self.H.dph_overlap_comm_init_slots(self.h, self.world, self.rank, idb.ctypes.data, ns) != 0:
                raise self._err()
        elif self.H.dph_overlap_comm_init(self.h, self.world, self.rank, idb.ctypes.data) != 0:
            raise self._err()

    @staticmethod
    def link_local(pipes):
        """Wires pipelines of ONE process (one per rank, created with mode="scan-shard", comm="local") into an in-process
        communicator: their survivor exchange then copies device to device between the contexts."""
        H = pipes[0].H
        arr = (C.c_void_p * len(pipes))(*[p.h for p in pipes])
        ns = max(1, pipes[0].slots) if pipes[0].mode == "scan-shard" else 1
        if ns > 1:  # one in-process communicator per executor slot
            if H.dph_overlap_comm_init_local_slots(arr, len(pipes), ns) != 0:
                raise DpError("dp_comm_init_local failed")
        elif H.dph_overlap_comm_init_local(arr, len(pipes)) != 0:
            raise DpError("dp_comm_init_local failed")

    def reset(self):
        """Ends the job (executor slots, planner, value table, k-mer index released; ignore flags cleared); the reads stay
        resident and init() starts the next job."""
        if self.H.dph_overlap_reset(self.h) != 0:
            raise self._err()

    def setup_times(self):
        out = np.zeros(3, dtype=np.float64)
        self.H.dph_overlap_setup_times(self.h, out.ctypes.data)
        return dict(context_s=out[0], upload_pack_s=out[1], init_s=out[2])

    def stats_total(self):
        out = np.zeros(len(STAT_FIELDS), dtype=np.float64)
        self.H.dph_overlap_stats_total(self.h, out.ctypes.data)
        return dict(zip(STAT_FIELDS, out.tolist()))

    def close(self):
        if getattr(self, "h", None):
            self.H.dph_overlap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return DpError(self.H.dph_last_error(self.h).decode())

    def values(self):
        n = C.c_int64(0)
        p = self.H.dph_overlap_values(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def local_survivors(self):
        ptrs = [C.c_void_p() for _ in range(4)]
        n, m = C.c_uint64(0), C.c_uint64(0)
        self.H.dph_overlap_local(self.h, *[C.byref(x) for x in ptrs], C.byref(n), C.byref(m))

        def arr(p, cnt, ct, dt):
            if cnt == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(cnt,)).astype(dt, copy=True)
        return dict(read=arr(ptrs[0], n.value, C.c_uint32, np.uint32), n_seeds=arr(ptrs[1], n.value, C.c_uint32, np.uint32),
                    segs=arr(ptrs[3], m.value, C.c_int32, np.int32))

    def step(self):
        """Advances the command.  Returns the number of rounds committed by this call (0 = finished)."""
        if self.mode == "single":
            rc = self.H.dph_overlap_step(self.h)
            if rc < 0:
                raise self._err()
            return rc
        if self.mode == "round" and self.comm is not None:
            # the same protocol with the exchange inside the library (dp_allgather_blobs: RCCL, or host copies between the
            # handles of one process)
            while True:
                if self.H.dph_overlap_done(self.h):
                    return 0
                c = self.H.dph_overlap_superstep(self.h, max(1, self.slots))
                if c < 0:
                    raise self._err()
                if c > 0:
                    return c
        if self.mode == "round":
            # pipelined: this rank's executor slots keep working on the rounds r % world == rank; one superstep = every
            # rank contributes its next owned round, all-gather, commit in round order on every rank
            while True:
                if self.H.dph_overlap_done(self.h):
                    return 0
                n = C.c_uint64(0)
                p = self.H.dph_overlap_wait_owned_many(self.h, max(1, self.slots), C.byref(n))
                if not p:
                    raise self._err()
                mine = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))
                cat, sizes = allgather_blobs(mine, self.world, self.torch_device)
                c = self.H.dph_overlap_commit_gathered(self.h, cat.ctypes.data, sizes.ctypes.data, len(sizes))
                if c < 0:
                    raise self._err()
                if c > 0:
                    return c  # 0 = the superstep's first round was rejected and is being executed again
        if self.mode == "round-batch":
            if self.H.dph_overlap_done(self.h):
                return 0
            base = self.H.dph_overlap_round(self.h)
            n = C.c_uint64(0)
            p = self.H.dph_overlap_exec_round(self.h, base + self.rank * self.slots, C.byref(n))
            if not p:
                raise self._err()
            blobs = allgather_bytes(C.string_at(p, n.value), self.world, self.torch_device)
            sizes = np.array([len(b) for b in blobs], dtype=np.uint64)
            cat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
            c = self.H.dph_overlap_commit_blobs(self.h, cat.ctypes.data, sizes.ctypes.data, len(blobs))
            if c < 0:
                raise self._err()
            return c
        # scan-shard
        if self.comm is not None and self.slots > 1:  # `slots` rounds at once, each on its own communicator
            rc = self.H.dph_overlap_rounds_sharded(self.h)
            if rc < 0:
                raise self._err()
            return rc
        if self.comm is not None:  # exchange inside the library (RCCL / in-process peers): one collective call per round
            rc = self.H.dph_overlap_round_sharded(self.h)
            if rc < 0:
                raise self._err()
            return rc
        rc = self.H.dph_overlap_round_scan(self.h)
        if rc < 0:
            raise self._err()
        if rc == 0:
            return 0
        allv = allgather_survivors(self.local_survivors(), self.world, self.torch_device)
        r = np.ascontiguousarray(allv["read"], dtype=np.uint32)
        ns = np.ascontiguousarray(allv["n_seeds"], dtype=np.uint32)
        sg = np.ascontiguousarray(allv["segs"], dtype=np.int32)
        rc = self.H.dph_overlap_round_finish(self.h, r.ctypes.data, ns.ctypes.data, sg.ctypes.data, len(r))
        if rc < 0:
            raise self._err()
        return 1

    # ---- round-parallel building blocks (also used by the single-GPU simulation test)
    def text_root(self, root):
        """Round-parallel runs with the exchange inside the library: supersteps gather the PAF text to rank `root` alone (every rank
        must say the same); -1 = text and control records to every rank."""
        self.H.dph_overlap_text_root.restype = None
        self.H.dph_overlap_text_root.argtypes = [C.c_void_p, C.c_int]
        self.H.dph_overlap_text_root(self.h, root)

    def keep_text(self, keep):
        """Multi-rank runs: keep=False on a rank that does not print the PAF drops the other ranks' text as it arrives."""
        self.H.dph_overlap_keep_text(self.h, 1 if keep else 0)

    def committed_rounds(self):
        return self.H.dph_overlap_round(self.h)

    def finished(self):
        return bool(self.H.dph_overlap_done(self.h))

    def exec_round_blob(self, rnd):
        """Executes rounds rnd .. rnd+slots-1 concurrently on this process's slots; returns their serialised results."""
        n = C.c_uint64(0)
        p = self.H.dph_overlap_exec_round(self.h, rnd, C.byref(n))
        if not p:
            raise self._err()
        return C.string_at(p, n.value)

    def step_lines(self):
        """PAF lines printed by the rounds the last step() committed."""
        self.H.dph_overlap_step_lines.restype = C.c_int64
        self.H.dph_overlap_step_lines.argtypes = [C.c_void_p]
        return int(self.H.dph_overlap_step_lines(self.h))

    def wait_owned_blob(self, max_rounds=None):
        """This rank's contribution to a superstep: its next owned round and the finished owned rounds after it (serialised)."""
        n = C.c_uint64(0)
        p = self.H.dph_overlap_wait_owned_many(self.h, max(1, self.slots) if max_rounds is None else max_rounds, C.byref(n))
        if not p:
            raise self._err()
        return C.string_at(p, n.value)

    def commit_gathered(self, blobs):
        sizes = np.array([len(b) for b in blobs], dtype=np.uint64)
        cat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        c = self.H.dph_overlap_commit_gathered(self.h, cat.ctypes.data, sizes.ctypes.data, len(blobs))
        if c < 0:
            raise self._err()
        return c

    def drain(self):
        """Discards the rounds the executor pipeline has in flight (they are executed again later)."""
        self.H.dph_overlap_drain(self.h)

    def commit_blobs(self, blobs):
        sizes = np.array([len(b) for b in blobs], dtype=np.uint64)
        cat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        c = self.H.dph_overlap_commit_blobs(self.h, cat.ctypes.data, sizes.ctypes.data, len(blobs))
        if c < 0:
            raise self._err()
        return c

    def stats(self):
        out = np.zeros(len(STAT_FIELDS), dtype=np.float64)
        self.H.dph_overlap_stats(self.h, out.ctypes.data)
        return dict(zip(STAT_FIELDS, out.tolist()))

    def _text(self, fn):
        n = C.c_int64(0)
        p = fn(self.h, C.byref(n))
        return C.string_at(p, n.value).decode()

    def round_paf(self):
        return self._text(self.H.dph_overlap_round_paf)

    def all_paf(self):
        return self._text(self.H.dph_overlap_all_paf)

    def err_text(self):
        return self._text(self.H.dph_overlap_errtext)

    def run(self, max_rounds=-1):
        """Commits rounds until the job is finished, or exactly max_rounds of them (a step commits every finished round it finds)."""
        n = 0
        if self.world == 1:
            self.H.dph_overlap_set_round_limit.restype = None
            self.H.dph_overlap_set_round_limit.argtypes = [C.c_void_p, C.c_int64]
            self.H.dph_overlap_set_round_limit(self.h, max_rounds if max_rounds >= 0 else -1)
        while max_rounds < 0 or n < max_rounds:
            c = self.step()
            if c == 0:
                break
            n += c
        return n
