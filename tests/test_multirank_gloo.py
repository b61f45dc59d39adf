"""N>1 path on CPU: two processes (gloo) shard the reads, each produces its shard's survivors (the ORACLE stands in
for the GPU scan here — it is the checker, the exchange code is what is under test), and the all-gathered list must
equal the single-rank list: ascending contiguous shards concatenated in rank order == file order."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from tests import oracle_lib as O


def _survivors(bases, off, lo, hi, k, seed_kmers, min_seeds):
    table = np.zeros(4 ** k, dtype=np.uint8)
    table[seed_kmers] = 1
    kmap = {int(km): i for i, km in enumerate(seed_kmers)}
    reads, nseeds, segs = [], [], []
    for r in range(lo, hi):
        s = O.Seq(bases[off[r]:off[r + 1]].tobytes().decode())
        seg = s.sub(0, int(off[r + 1] - off[r])).write_segments(k, table)
        n = len(seg) // 2
        if n >= min_seeds:
            seg[1::2] = [kmap[int(x)] for x in seg[1::2]]
            reads.append(r)
            nseeds.append(n)
            segs.append(seg.astype(np.int32))
    return dict(read=np.array(reads, dtype=np.uint32), n_seeds=np.array(nseeds, dtype=np.uint32),
                segs=np.concatenate(segs) if segs else np.zeros(0, dtype=np.int32))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from downpore_amd.overlap import allgather_survivors, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = 8
    bases, off = O.gen_reads(77, 30000, 90, 1500, 0.0, True)
    rng = np.random.default_rng(5)
    seed_kmers = np.unique(rng.integers(1, 4 ** k, 3000)).astype(np.int64)
    lo, hi = shard_bounds(90, rank, world)
    local = _survivors(bases, off, lo, hi, k, seed_kmers, 15)
    allv = allgather_survivors(local, world)
    if rank == 0:
        full = _survivors(bases, off, 0, 90, k, seed_kmers, 15)
        ok = all(np.array_equal(allv[key], full[key]) for key in ("read", "n_seeds", "segs"))
        q.put((ok, len(full["read"]), len(local["read"])))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_survivors_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, n_full, n_local = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and n_full > n_local > 0


def test_shard_bounds_cover_and_are_ascending():
    from downpore_amd.overlap import shard_bounds
    for n in (0, 1, 7, 100, 100001):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert lo == prev and hi >= lo
                prev = hi
            assert prev == n


def _blob_worker(rank, world, port, q):
    import torch.distributed as dist
    from downpore_amd.overlap import allgather_blobs, allgather_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # the result exchange of the round-parallel mode: one byte string per rank, lengths differ from rank to rank and from
    # superstep to superstep (a rank contributes one or several serialised rounds), an empty one included
    for step, lens in enumerate([(5, 70000), (0, 3), (4096, 4096), (123457, 1)]):
        want = [(np.arange(n, dtype=np.uint64) * (7 + r + step) % 251).astype(np.uint8) for r, n in enumerate(lens)]
        cat, sizes = allgather_blobs(want[rank].tobytes(), world)
        ok &= sizes.tolist() == list(lens) and cat.dtype == np.uint8 and np.array_equal(cat, np.concatenate(want))
        ok &= allgather_bytes(want[rank], world) == [w.tobytes() for w in want]
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_blobs_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_blob_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
