"""PAF held to the synthetic genome itself: the generator (tools/synth.py, gen_reads_truth) knows where every read came from, so
overlap and mapping lines can be judged without any second implementation of the reference.  Shared by bench.py and tests/."""
import numpy as np


def _fields(paf, limit=None):
    if isinstance(paf, bytes):
        paf = paf.decode()
    lines = paf.split("\n") if limit is None else paf.split("\n", limit)[:limit]
    rows = [ln.split("\t") for ln in lines if ln]
    return np.array([r[:9] for r in rows], dtype=object) if rows else None


def overlap_truth(paf, off, starts, strands, k, sample=200000):
    """The first `sample` PAF lines of `downpore overlap` against where the generator took the reads from: strand mismatches, lines
    whose reads do not overlap on the genome, fraction of lines whose two parts begin and end within k (30) bases of each other on
    the genome."""
    f = _fields(paf, sample)
    if f is None:
        return None
    q = np.array([int(x[1:]) for x in f[:, 0]])
    t = np.array([int(x[1:]) for x in f[:, 5]])
    qs, qe, ts, te = (f[:, c].astype(np.int64) for c in (2, 3, 7, 8))
    minus = f[:, 4] == "-"
    Ls = np.diff(off)

    def gpos(r, x):
        return np.where(strands[r] == 0, starts[r] + x, starts[r] + Ls[r] - x)
    a0, a1 = np.minimum(gpos(q, qs), gpos(q, qe)), np.maximum(gpos(q, qs), gpos(q, qe))
    b0, b1 = np.minimum(gpos(t, ts), gpos(t, te)), np.maximum(gpos(t, ts), gpos(t, te))
    n = len(f)
    return {"lines_checked": n, "strand_mismatches": int(((strands[q] != strands[t]) != minus).sum()),
            "reads_that_do_not_overlap_on_the_genome": int((~((starts[q] < starts[t] + Ls[t]) & (starts[t] < starts[q] + Ls[q]))).sum()),
            "parts_without_a_shared_base": int((np.minimum(a1, b1) <= np.maximum(a0, b0)).sum()),
            "frac_both_ends_within_k_bases": float(((np.abs(a0 - b0) <= k) & (np.abs(a1 - b1) <= k)).mean()),
            "frac_both_ends_within_30_bases": float(((np.abs(a0 - b0) <= 30) & (np.abs(a1 - b1) <= 30)).mean())}


def map_truth(paf, off, starts, strands, G, n_reads=None):
    """`downpore map` PAF against where the generator took every read from (the reference publishes recall / precision of its
    mapper against a truth set, README.md:220-237).  A mapping is TRUE when its strand is the read's and both ends of its
    reference interval lie where the read's mapped stretch [qstart, qend) really came from, to within 3 % of the read's length
    + 100 bases (the generator's insertions and deletions shift coordinates; a circular reference is compared modulo G).
    Returns dict(mappings, true, precision, reads_mapped, reads_true = reads with at least one true mapping, recall over the
    input reads, covered = mean fraction of a truly mapped read's bases inside its true mappings)."""
    L = np.diff(off)
    f = _fields(paf)
    if f is None:
        return None
    r = np.array([int(x[1:]) for x in f[:, 0]])
    qs, qe, ts, te = (f[:, c].astype(np.int64) for c in (2, 3, 7, 8))
    minus = f[:, 4] == "-"
    exp_s = np.where(strands[r] == 0, starts[r] + qs, starts[r] + L[r] - qe)
    exp_e = np.where(strands[r] == 0, starts[r] + qe, starts[r] + L[r] - qs)
    tol = (0.03 * L[r] + 100).astype(np.int64)

    def near(a, b):
        d = np.abs(a - b) % G
        return np.minimum(d, G - d) <= tol
    true = (minus == (strands[r] == 1)) & near(ts, exp_s) & near(te, exp_e)
    n = len(L) if n_reads is None else n_reads
    covered = np.zeros(len(L))
    np.add.at(covered, r[true], (qe - qs)[true])
    reads_true = np.unique(r[true])
    return dict(mappings=len(f), true=int(true.sum()), precision=float(true.mean()), reads_mapped=int(len(np.unique(r))),
                reads_true=int(len(reads_true)), recall=float(len(reads_true) / n),
                covered=float((covered[reads_true] / L[reads_true]).mean()) if len(reads_true) else 0.0)
