// Product host side: overlap.Overlapper (overlap/overlap.go) and the overlap command's round loop
// (commands/overlap.go:96-233) above the C ABI.  Canonical single-worker order (DESIGN.md §canonical semantics).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <sys/resource.h>
#include <sys/mman.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <time.h>
#include <cstdio>
#include <cstring>

#include "dph.hpp"

namespace dph {

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double threadCpuNow() {  // CPU time consumed by the calling thread
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

unsigned hostThreads();
// DPH_PROFILE=1: pipeline counters printed to stderr when a run shuts down
struct PipeProfile {
    std::atomic<long long> executed{0}, committed{0}, rejected{0}, discarded{0}, ignores{0}, planComputes{0}, planErased{0},
        planDiscarded{0};
    std::atomic<long long> planUs{0}, getWaitUs{0}, execUs{0}, commitUs{0}, consensusCpuUs{0}, selectCpuUs{0}, slotCpuUs{0}, plannerCpuUs{0};
    std::atomic<long long> sub[18];
    const char* subName[18] = {"prep.indexReset", "prep.newOverlapper", "prep.roundBegin", "scan.call", "scan.copy", "idx.copy",
                               "idx.chunk", "idx.build", "idx.queries", "qry.call", "qry.matches", "fc.collate", "fc.parallel",
                               "fc.merge", "round.total", "round.tail", "plan.speculate", "plan.commitLoop"};
    PipeProfile() {
        for (auto& x : sub) x = 0;
    }
    void add(int i, double sec) { sub[i] += (long long)(sec * 1e6); }
    bool on = getenv("DPH_PROFILE") != nullptr;
    void print() {
        if (!on) return;
        fprintf(stderr,
                "[pipe] rounds executed %lld committed %lld rejected %lld discarded %lld | new ignores %lld | plans computed %lld "
                "(%.2f ms each) erased %lld thrown away %lld | plan wait %.1f ms, execute %.1f ms, commit %.1f ms\n",
                executed.load(), committed.load(), rejected.load(), discarded.load(), ignores.load(), planComputes.load(),
                planComputes.load() ? planUs.load() / 1e3 / planComputes.load() : 0.0, planErased.load(), planDiscarded.load(),
                getWaitUs.load() / 1e3, execUs.load() / 1e3, commitUs.load() / 1e3);
        const double n = (double)std::max<long long>(1, executed.load());
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[pipe] host threads %u, process CPU time user %.2f s sys %.2f s\n", hostThreads(),
                ru.ru_utime.tv_sec + ru.ru_utime.tv_usec / 1e6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1e6);
        {
            const double nn = (double)std::max<long long>(1, executed.load());
            fprintf(stderr, "[pipe] thread CPU per round (ms): consensus items %.2f, seed-selection items %.2f, slot threads %.2f, planner thread %.2f\n",
                    consensusCpuUs.load() / 1e3 / nn, selectCpuUs.load() / 1e3 / nn, slotCpuUs.load() / 1e3 / nn,
                    plannerCpuUs.load() / 1e3 / nn);
        }
        fprintf(stderr, "[pipe] per executed round (ms):");
        for (int i = 0; i < 18; i++) fprintf(stderr, " %s %.3f", subName[i], sub[i].load() / 1e3 / n);
        fprintf(stderr, "\n");
    }
};
static PipeProfile g_prof;

Overlapper::Overlapper(dp_ctx* ctx, ReadSet& reads, SeedIndex& index, i64 chunkSize, int, i64 overlap, int minSeeds,
                       double hitFraction)
    : ctx_(ctx), reads_(reads), index_(index), chunkSize_(chunkSize), overlap_(overlap), minSeeds_(minSeeds),
      hitFraction_(hitFraction) {
    ignore_ = reads.ignore.data();
}

// PrepareQueries :157-214 with getEdges :55-89 (QueryEdges): seed selection is sequential and stays on the host.
// Returns the number of query windows.
int Overlapper::PrepareQueries(int numSeeds, i64 seedLimit, const double* values, i64 firstSequence, i64 maxSeqs, int queryType) {
    windows_.clear();
    queries.clear();
    const bool weightSides = (queryType & 8) != 0;  // WeightEdges: seeds come from the two 200-base sides of a window
    if (weightSides) numSeeds /= 2;                  // overlap.go:161-163
    if (numSeeds < 1) return 0;
    // Query windows in file order (getEdges :55-89 / getCentres :91-117 / getAll :119-155), each with the one or two
    // sub-windows AddSeeds actually sees (addWeighted :45-53).  The seed-budget cut-off (tested once per read) is applied
    // in the sequential commit loop below; a sub-window contributes at most 2*numSeeds seeds, so
    // seedLimit/(2*numSeeds) of them are certainly needed and a modest surplus is selected speculatively.
    // (Selection = the rank-table lookups = the expensive part.)
    struct Sel {
        uint32_t read, start, len;
    };
    struct Cand {
        uint32_t read, start, len;
        uint32_t selFirst, selCount;
        bool firstOfRead;
    };
    std::vector<Cand> cand;
    std::vector<Sel> sel;
    const size_t want = (size_t)(seedLimit / std::max(1, 2 * numSeeds)) + 96;
    i64 sent = 0;
    size_t rNext = (size_t)firstSequence;
    auto addWindow = [&](uint32_t read, i64 start, i64 end, bool firstOfRead) {
        Cand c{read, (uint32_t)start, (uint32_t)(end - start), (uint32_t)sel.size(), 0u, firstOfRead};
        const i64 len = end - start, sideSize = 200;
        if (weightSides && len > 400) {
            sel.push_back({read, (uint32_t)start, (uint32_t)sideSize});
            sel.push_back({read, (uint32_t)(end - sideSize), (uint32_t)sideSize});
            c.selCount = 2;
        } else {
            sel.push_back({read, (uint32_t)start, (uint32_t)len});
            c.selCount = 1;
        }
        cand.push_back(c);
    };
    auto moreCands = [&](size_t upTo) {
        if (firstSequence != 0 && firstSequence >= (i64)reads_.size()) return;  // seqio.go:279
        for (; rNext < reads_.size() && sent < maxSeqs && sel.size() < upTo; rNext++) {
            if (ignore_[rNext]) continue;
            sent++;
            const uint32_t r = (uint32_t)rNext;
            const i64 L = reads_.length(rNext);
            if (queryType & 1) {  // QueryEdges
                if (L < overlap_ * 2) {
                    addWindow(r, 0, L, true);
                } else {
                    addWindow(r, 0, overlap_, true);
                    addWindow(r, L - overlap_, L, false);
                }
            } else if (queryType & 2) {  // QueryCentre
                i64 start = (L - overlap_) / 2;
                if (start < 0) start = 0;
                i64 end = start + overlap_;
                if (end >= L) end = L - 1;
                addWindow(r, start, end, true);
            } else {  // QueryAll
                if (L < overlap_ * 2) {
                    addWindow(r, 0, L, true);
                } else {
                    const i64 slices = L / overlap_;
                    for (i64 i = 0; i < slices; i++) {
                        const i64 start = (i * L) / slices;
                        i64 end = ((i + 1) * L) / slices;
                        if (i == slices - 1) end = L;
                        addWindow(r, start, end, i == 0);
                    }
                }
            }
        }
    };
    std::vector<uint32_t> spec;
    size_t specDone = 0;
    bool specFailed = false;
    auto speculate = [&](size_t upTo) {  // selection of every sub-window assuming no evaluated k-mer is a seed yet
        moreCands(upTo);
        const size_t n = sel.size();
        spec.resize(n * (size_t)numSeeds);
        const size_t first = specDone;
        if (n <= first) return;
        if (ctx_) {  // device-side selection over the resident reads (dp_select_seeds); ctx_ is the planner's context
            std::vector<dp_scan_item> items(n - first);
            for (size_t w = first; w < n; w++) {
                dp_scan_item& it = items[w - first];
                it.read = sel[w].read;
                it.start = sel[w].start;
                it.n_kmers = sel[w].len;  // window length in bases
                it.min_seeds = 0;
            }
            const int rc = dp_select_seeds(ctx_, items.data(), (uint32_t)items.size(), index_.k, numSeeds,
                                           &spec[first * (size_t)numSeeds]);
            if (rc != 0) {
                err = dp_last_error(ctx_);
                specFailed = true;
            }
        } else {
            std::atomic<long long> selUs(0);
            parallelFor(n - first, [&](size_t i) {
                const size_t w = first + i;
                const Sel& c = sel[w];
                const double tw = g_prof.on ? threadCpuNow() : 0;
                index_.selectSeeds(reads_.seq(c.read) + c.start, c.len, numSeeds, values, &spec[w * (size_t)numSeeds], false);
                if (g_prof.on) selUs += (long long)((threadCpuNow() - tw) * 1e6);
            });
            g_prof.selectCpuUs += selUs.load();
        }
        specDone = n;
    };
    const double tsp0 = now();
    speculate(want);
    if (specFailed) return -1;
    const double tsp1 = now();
    g_prof.add(16, tsp1 - tsp0);
    struct CommitTick {
        double t0;
        ~CommitTick() { g_prof.add(17, now() - t0); }
    } commitTick{tsp1};
    std::vector<uint32_t> tmp((size_t)numSeeds);
    size_t w = 0;
    for (;;) {
        if (w >= cand.size()) {
            speculate(sel.size() + 64);
            if (specFailed) return -1;
            if (w >= cand.size()) break;  // input exhausted
        }
        // the budget is tested once per READ, before its first window (overlap.go:57-60, 93-96, 121-127)
        if (cand[w].firstOfRead && index_.size() >= seedLimit) break;
        const Cand& c = cand[w];
        for (uint32_t si = c.selFirst; si < c.selFirst + c.selCount; si++) {
            const Sel& sw = sel[si];
            const char* s = reads_.seq(sw.read) + sw.start;
            if (index_.touchesSeed(s, sw.len)) {  // speculation invalid: redo this window against the current seed set
                index_.selectSeeds(s, sw.len, numSeeds, values, tmp.data(), true);
                index_.commitSeeds(tmp.data(), numSeeds);
            } else {
                index_.commitSeeds(&spec[si * (size_t)numSeeds], numSeeds);
            }
        }
        windows_.push_back({c.read, c.start, c.len});
        w++;
    }
    return (int)windows_.size();
}

// AddSequences :217 (scan part).  One dp_scan call: every non-ignored read of this process's shard (segments only for
// reads with >= minSeeds hits: chunkWorker drops the others, :259-261) followed by all query windows.
int Overlapper::ScanLocal(size_t lo, size_t hi, Survivors& local, RoundStats& st) {
    const int k = index_.k;
    std::vector<dp_scan_item> items;  // the query windows; the reads themselves are enumerated on the device
    items.reserve(windows_.size());
    for (const Window& w : windows_) {
        dp_scan_item it;
        it.read = w.read;
        it.start = w.start;
        // a window is a SubSequence view (finalLen in 1..4): the scan examines len-k+1 k-mers; the whole-read window of a
        // short read served top-level (himem=false) inherits the len%4 quirk
        i64 nk = (i64)w.len - k + 1;
        if (!reads_.himem && w.start == 0 && (i64)w.len == reads_.length(w.read) && (w.len % 4) == 0) nk -= 4;
        it.n_kmers = (uint32_t)std::max<i64>(0, nk);
        it.min_seeds = 0;
        items.push_back(it);
    }
    dp_survivor_batch b;
    const double ts0 = now();
    int rc = dp_scan_reads(ctx_, ignore_, ignoreEpoch_, (uint32_t)lo, (uint32_t)hi, reads_.himem ? 0 : 1, (uint32_t)minSeeds_,
                           items.data(), (uint32_t)items.size(), &b);
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    const double ts1 = now();
    g_prof.add(3, ts1 - ts0);
    st.k_scan_ms += b.kernel_ms;
    st.k_count_ms += b.count_kernel_ms;
    st.k_write_ms += b.write_kernel_ms;
    st.scan_bases += b.bases_scanned;
    st.scan_items += b.reads_scanned + items.size();
    // survivors of the local shard (ascending read id); their segments are the leading part of the scan output
    local.read.assign(b.read, b.read + b.n_survivors);
    local.n_seeds.assign(b.n_seeds, b.n_seeds + b.n_survivors);
    local.seg_off.assign(b.seg_off, b.seg_off + b.n_survivors);
    uint64_t survEnd = 0;
    if (b.n_survivors) survEnd = b.seg_off[b.n_survivors - 1] + 2ull * b.n_seeds[b.n_survivors - 1] + 1;
    local.seg_off.push_back(survEnd);
    local.segs.clear();
    local.segsView = b.segs;  // pinned output of this context, untouched until its next scan (= this slot's next round)
    local.segsViewLen = survEnd;
    local.deviceResident = true;
    // query windows
    winSegs_.clear();
    winOff_.assign(1, 0);
    for (uint32_t i = 0; i < b.n_extra; i++) {
        const uint64_t o = b.extra_seg_off[i];
        winSegs_.insert(winSegs_.end(), b.segs + o, b.segs + o + 2ull * b.extra_n_seeds[i] + 1);
        winOff_.push_back(winSegs_.size());
    }
    st.n_hits += b.n_segs / 2;  // hits written this scan, for the roofline's algorithmic bytes
    g_prof.add(4, now() - ts1);
    return 0;
}

// chunkWorker :253-318 for one seed sequence whose segments start at device offset segBase
void Overlapper::chunkAndAdd(SeedSeq* s, uint64_t segBase) {
    const int k = index_.k;
    Arena& ar = index_.arena;
    auto add = [&](SeedSeq* q) {
        index_.sequences.push_back(q);
        dp_seq_ref r;
        r.seg_off = segBase + (uint64_t)(q->seg - s->seg);
        r.n_seeds = (uint32_t)q->numSeeds();
        r.reserved = 0;
        index_.refs.push_back(r);
    };
    const i64 numChunks = s->length / chunkSize_ + 1;
    if (numChunks == 1 || s->numSeeds() < minSeeds_ * 3) {
        if (s->numSeeds() >= minSeeds_) add(s);
        return;
    }
    int prevSeedIndex = 0;
    i64 totalOffset = s->seedOffset(0, k);
    i64 lengthInBases = 0;
    for (;;) {
        int seedCount = 0;
        if (prevSeedIndex >= s->numSeeds() - 150) {
            if (prevSeedIndex == 0) {
                add(s);
            } else {
                const i64 newFirstGap = s->nextSeedOffset(prevSeedIndex - 1, k) - k;
                lengthInBases += s->seedOffsetFromEnd(prevSeedIndex, k) + k + newFirstGap;
                add(seqSubSequence(ar, s, prevSeedIndex, s->numSeeds() - 1, lengthInBases, totalOffset - newFirstGap, 0));
            }
            break;
        }
        for (; lengthInBases < chunkSize_ && seedCount < 100 && prevSeedIndex + seedCount < s->numSeeds(); seedCount++)
            lengthInBases += s->nextSeedOffset(prevSeedIndex + seedCount, k);
        if (seedCount >= minSeeds_) {
            const i64 newFirstGap = s->nextSeedOffset(prevSeedIndex - 1, k) - k;
            lengthInBases += newFirstGap;
            add(seqSubSequence(ar, s, prevSeedIndex, prevSeedIndex + seedCount - 1, lengthInBases, totalOffset - newFirstGap,
                               s->length - totalOffset - lengthInBases + newFirstGap));
            totalOffset += lengthInBases - newFirstGap;
            lengthInBases = 0;
            prevSeedIndex += seedCount;
            if (prevSeedIndex >= s->numSeeds()) break;
            for (seedCount = 0; seedCount < 5 && lengthInBases < overlap_ / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                const i64 step = s->nextSeedOffset(prevSeedIndex, k);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        } else {
            prevSeedIndex += seedCount;
            for (seedCount = 0; lengthInBases < overlap_ / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                const i64 step = s->nextSeedOffset(prevSeedIndex, k);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        }
    }
}

// AddSequences :217 (chunk + index part) from the complete survivor list (file order).
int Overlapper::IndexSurvivors(const Survivors& all, RoundStats& st) {
    double tp0 = now();
    allSegs_ = all.segData();
    // the device-resident scan output the index refers to must hold exactly this survivor array at the same offsets:
    // true right after a local full scan; after a multi-GPU exchange the gathered array is imported
    int rc = 0;
    if (!all.deviceResident) rc = dp_scan_import_segments(ctx_, allSegs_, all.segCount());
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    index_.sequences.clear();
    index_.refs.clear();
    double tp1 = now();
    g_prof.add(5, tp1 - tp0);
    for (size_t i = 0; i < all.read.size(); i++) {
        const uint32_t r = all.read[i];
        SeedSeq* s = index_.arena.make();
        s->seg = allSegs_ + all.seg_off[i];
        s->n = (int)(all.seg_off[i + 1] - all.seg_off[i]);
        s->id = (int)r;
        s->length = reads_.length(r);
        s->offset = 0;
        s->inset = reads_.servedInset();
        chunkAndAdd(s, all.seg_off[i]);
    }
    double tp2 = now();
    g_prof.add(6, tp2 - tp1);
    rc = dp_index_build(ctx_, index_.refs.data(), (uint32_t)index_.refs.size());
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    double tp3 = now();
    g_prof.add(7, tp3 - tp2);
    st.n_indexed = index_.refs.size();
    // queries: [fwd, rc] per window (PrepareQueries :189-201)
    queries.clear();
    int queryID = 0;
    for (size_t w = 0; w < windows_.size(); w++) {
        const Window& win = windows_[w];
        SeedSeq* s = index_.arena.make();
        s->seg = winSegs_.data() + winOff_[w];
        s->n = (int)(winOff_[w + 1] - winOff_[w]);
        s->id = (int)win.read;
        s->length = win.len;
        const i64 L = reads_.length(win.read);
        if (win.start == 0 && (i64)win.len == L) {  // the served view itself
            s->offset = 0;
            s->inset = reads_.servedInset();
        } else {  // SubSequence(start, start+len) of the served view (sequence.go:353-370)
            s->offset = win.start;
            s->inset = reads_.servedInset() + L - ((i64)win.start + win.len - 1);
        }
        SeedQuery q;
        q.ID = queryID;
        q.SequenceID = s->id;
        q.Query = s;
        q.ReverseComplement = false;
        queries.push_back(q);
        SeedQuery rcq = q;
        rcq.Query = seqReverseComplement(index_.arena, s, index_);
        rcq.ReverseComplement = true;
        queries.push_back(rcq);
        queryID++;
    }
    st.n_queries = queries.size();
    g_prof.add(8, now() - tp3);
    return 0;
}

// FindOverlaps :320 + matchWorker :346
int Overlapper::FindOverlaps(std::vector<SeedMatch>& pool, std::vector<SeedMatch*>& out, RoundStats& st) {
    querySegs_.clear();
    queryOff_.assign(1, 0);
    for (const SeedQuery& q : queries) {
        querySegs_.insert(querySegs_.end(), q.Query->seg, q.Query->seg + q.Query->n);
        queryOff_.push_back(querySegs_.size());
    }
    dp_match_batch mb;
    const double tq0 = now();
    int rc = dp_find_overlaps(ctx_, querySegs_.data(), queryOff_.data(), (uint32_t)queries.size(), hitFraction_, index_.k,
                              (uint32_t)(overlap_ / 2), 0, &mb);
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    const double tq1 = now();
    g_prof.add(9, tq1 - tq0);
    st.k_query_ms += mb.query_kernel_ms;
    st.k_chain_ms += mb.chain_kernel_ms;
    st.query_bytes += mb.query_bytes;
    out.clear();
    out.reserve(mb.n_matches);
    if (pool.size() < mb.n_matches) pool.resize(mb.n_matches);
    for (uint32_t i = 0; i < mb.n_matches; i++) {
        SeedMatch* m = &pool[i];
        const SeedQuery& q = queries[mb.query[i]];
        m->MatchA.assign(mb.match_a + mb.off[i], mb.match_a + mb.off[i + 1]);
        m->MatchB.assign(mb.match_b + mb.off[i], mb.match_b + mb.off[i + 1]);
        m->SeqA = q.Query;
        m->SeqB = index_.sequences[mb.target[i]];
        m->QueryID = q.ID;
        m->ReverseComplementQuery = q.ReverseComplement;
        out.push_back(m);
    }
    st.n_matches = out.size();
    g_prof.add(10, now() - tq1);
    return 0;
}

// finalCheckWorker commands/overlap.go:197-233 (+ collation :158-173).  Queries are independent (the reference runs
// num_workers finalCheckWorkers); they are spread over host threads here, and the PAF text and SetIgnore effects are
// applied in query order afterwards, so the result is the canonical single-worker output.
static void finalCheckOne(Arena& arena, const SeedIndex& index, const ReadSet& reads, std::vector<SeedMatch*>& results,
                          i64 overlapSize, std::string& paf, std::vector<int>& ignoreIds, FinalCheckStats& fs) {
    const int k = index.k;
    SeedContig* contig = buildConsensus(arena, index, results, &fs.badBack);
    if (!contig || contig->Parts.size() <= 1) return;
    if (contig->SeqLengths[0] <= overlapSize * 2) ignoreIds.push_back(contig->Parts[0]);
    const i64 queryStart = contig->Offsets[0], queryEnd = queryStart + contig->Lengths[0];
    char num[24];
    auto app = [&](i64 v) {  // %d
        char* e = num + sizeof num;
        char* p = e;
        uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        do {
            *--p = (char)('0' + u % 10);
            u /= 10;
        } while (u);
        if (v < 0) *--p = '-';
        paf.append(p, (size_t)(e - p));
    };
    for (size_t i = 0; i + 1 < contig->Parts.size(); i++) {
        const size_t id = i + 1;
        const int part = contig->Parts[id];
        const i64 start = contig->Offsets[id], end = start + contig->Lengths[id];
        const char* rcs = contig->ReverseComplement[0] != contig->ReverseComplement[id] ? "-" : "+";
        i64 covered = overlapSize;
        if (end - start > overlapSize) covered = end - start;
        if (contig->SeqLengths[id] * 9 <= covered * 10) ignoreIds.push_back(part);
        i64 ident = 0, identB = 0;
        bool panic = false;
        matchBasesCovered(*contig->Matches[i], k, &ident, &identB, &panic);
        if (panic) fs.emptyMatch++;  // the reference panics here; canonical: ident 0
        paf += reads.names[(size_t)contig->Parts[0]];
        paf += '\t';
        app(contig->SeqLengths[0]);
        paf += '\t';
        app(queryStart);
        paf += '\t';
        app(queryEnd);
        paf += '\t';
        paf += rcs;
        paf += '\t';
        paf += reads.names[(size_t)part];
        paf += '\t';
        app(contig->SeqLengths[id]);
        paf += '\t';
        app(start);
        paf += '\t';
        app(end);
        paf += '\t';
        app(ident);
        paf += "\t0\t255\n";
        fs.lines++;
    }
}

void finalCheck(Arena& arena, const SeedIndex& index, ReadSet& reads, std::vector<std::unique_ptr<SeedMatch>>& matches,
                i64 numQuerySeqs, i64 overlapSize, std::string& paf, FinalCheckStats& fs, std::vector<int>* ignoreOut) {
    std::vector<SeedMatch*> ptrs;
    ptrs.reserve(matches.size());
    for (auto& m : matches) ptrs.push_back(m.get());
    finalCheck(arena, index, reads, ptrs, numQuerySeqs, overlapSize, paf, fs, ignoreOut);
}

void finalCheck(Arena& arena, const SeedIndex& index, ReadSet& reads, const std::vector<SeedMatch*>& matches, i64 numQuerySeqs,
                i64 overlapSize, std::string& paf, FinalCheckStats& fs, std::vector<int>* ignoreOut) {
    // collate by QueryID (:158-173)
    const double tf0 = now();
    std::vector<std::vector<SeedMatch*>> queryResults((size_t)numQuerySeqs);
    i64 hits = 0, qHits = 0;
    for (SeedMatch* m : matches) {
        hits++;
        auto& qr = queryResults[(size_t)m->QueryID];
        if (qr.size() == 1) qHits++;
        qr.push_back(m);
    }
    std::vector<size_t> work;
    for (size_t q = 0; q < queryResults.size(); q++)
        if (queryResults[q].size() > 1) work.push_back(q);
    const size_t nw = work.size();
    std::vector<std::string> outs(nw);
    std::vector<std::vector<int>> ign(nw);
    std::vector<FinalCheckStats> tfs(nw);
    (void)arena;
    const double tf1 = now();
    g_prof.add(11, tf1 - tf0);
    std::atomic<long long> cpuUs(0);
    parallelFor(nw, [&](size_t w) {
        static thread_local Arena local;  // scratch SeedSeqs of this worker; nothing outlives the call
        const double tw = g_prof.on ? threadCpuNow() : 0;
        // results are built in worker-local objects and published once: neighbouring outs[]/ign[]/tfs[] elements share
        // cache lines, and finalCheckOne appends to them per PAF field
        std::string pafLocal;
        std::vector<int> ignLocal;
        FinalCheckStats fsLocal;
        pafLocal.reserve(2048);
        finalCheckOne(local, index, reads, queryResults[work[w]], overlapSize, pafLocal, ignLocal, fsLocal);
        outs[w] = std::move(pafLocal);
        ign[w] = std::move(ignLocal);
        tfs[w] = fsLocal;
        local.clear();
        if (g_prof.on) cpuUs += (long long)((threadCpuNow() - tw) * 1e6);
    });
    g_prof.consensusCpuUs += cpuUs.load();
    const double tf2 = now();
    g_prof.add(12, tf2 - tf1);
    for (size_t w = 0; w < nw; w++) {
        paf += outs[w];
        for (int id : ign[w]) {
            if (ignoreOut) ignoreOut->push_back(id);
            else reads.ignore[(size_t)id] = 1;
        }
    }
    for (auto& t : tfs) {
        fs.badBack += t.badBack;
        fs.emptyMatch += t.emptyMatch;
        fs.lines += t.lines;
    }
    fs.hits = (uint64_t)hits;
    fs.qHits = (uint64_t)qHits;
    g_prof.add(13, now() - tf2);
}

void profilePrint() { g_prof.print(); }
void setHostThreadShare(unsigned) {}  // kept for callers; the shared pool needs no per-slot split
// CPUs this process may actually use: the cgroup CPU quota (containers often expose every host CPU but cap the CPU
// time; exceeding the cap gets the whole process throttled for the rest of the scheduler period) or the CPU count.
static unsigned cpuBudget() {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char a[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        fclose(f);
    } else {
        long long q = -1, per = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &q) != 1) q = -1;
            fclose(g);
        }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(g, "%lld", &per) != 1) per = 0;
            fclose(g);
        }
        if (q > 0 && per > 0) quota = (double)q / (double)per;
    }
    if (quota >= 1.0 && quota < (double)hw) hw = (unsigned)quota;
    return hw;
}

unsigned hostThreads() {
    static unsigned n = [] {
        const char* e = getenv("DP_HOST_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : cpuBudget();
        if (v == 0) v = 1;
        return std::min(v, 96u);
    }();
    return n;
}

namespace {
struct PoolJob {
    size_t n = 0;
    const std::function<void(size_t)>* fn = nullptr;
    std::atomic<size_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
};
class WorkPool {
   public:
    static WorkPool& get() {
        static WorkPool* p = new WorkPool();  // intentionally leaked: workers may outlive static destruction order
        return *p;
    }
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || threads_.empty()) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<PoolJob>();
        job->n = n;
        job->fn = &fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work(*job);
        if (job->done.load(std::memory_order_acquire) < n) {  // items still running on pool threads
            std::unique_lock<std::mutex> jl(job->mu);
            job->cv.wait(jl, [&] { return job->done.load(std::memory_order_acquire) >= n; });
        }
        std::lock_guard<std::mutex> lk(mu_);
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (it->get() == job.get()) {
                jobs_.erase(it);
                break;
            }
    }

   private:
    WorkPool() {
        const unsigned n = hostThreads();
        // Workers sleep between jobs and are woken together by the submitting thread; the scheduler tends to leave such
        // short bursts stacked on the waker's CPU.  Each worker is therefore pinned to its own CPU of the allowed set
        // (spread evenly) when DP_PIN_WORKERS=1.
        std::vector<int> cpus;
        const char* pin = getenv("DP_PIN_WORKERS");
        if (pin && pin[0] != '0') {  // 1: one worker per physical core, spread over all cores; 2: the same within NUMA node 0
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof set, &set) == 0)
                for (int c = 0; c < CPU_SETSIZE; c++) {
                    if (!CPU_ISSET(c, &set)) continue;
                    char path[128];
                    int first = c, node0 = 1;
                    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
                    if (FILE* f = fopen(path, "r")) {
                        if (fscanf(f, "%d", &first) != 1) first = c;
                        fclose(f);
                    }
                    if (first != c) continue;  // SMT sibling of a lower-numbered CPU
                    if (pin[0] == '2') {
                        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node0", c);
                        node0 = access(path, F_OK) == 0;
                    }
                    if (node0) cpus.push_back(c);
                }
        }
        for (unsigned i = 1; i < n; i++) {
            threads_.emplace_back([this] { loop(); });
            if (cpus.size() >= n) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[(size_t)i * cpus.size() / n], &one);
                pthread_setaffinity_np(threads_.back().native_handle(), sizeof one, &one);
            }
        }
        for (auto& t : threads_) t.detach();
    }
    static void work(PoolJob& j) {
        for (;;) {
            const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) return;
            (*j.fn)(i);
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.n) {
                std::lock_guard<std::mutex> jl(j.mu);
                j.cv.notify_all();
            }
        }
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            std::shared_ptr<PoolJob> job;
            for (auto& j : jobs_)
                if (j->next.load(std::memory_order_relaxed) < j->n) {
                    job = j;
                    break;
                }
            if (!job) {
                cv_.wait(lk);
                continue;
            }
            lk.unlock();
            work(*job);
            lk.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::shared_ptr<PoolJob>> jobs_;
    std::vector<std::thread> threads_;
};
}  // namespace

void parallelFor(size_t n, const std::function<void(size_t)>& fn) { WorkPool::get().run(n, fn); }

// ---------------------------------------------------------------------------------------------------------------
// Planner: the PrepareQueries chain (overlap.go:157-214 seed selection + commands/overlap.go:128-143 bookkeeping)

struct Planner::Impl {
    ReadSet& reads;
    OverlapParams p;
    const double* values;
    bool threaded;
    dp_ctx* selCtx;
    SeedIndex index;  // selection-side seed set of the plan being computed
    std::mutex mu;
    std::condition_variable cv;
    std::map<i64, std::shared_ptr<RoundPlan>> cache;
    i64 wantUpTo = -1;        // prefetch target (highest requested round + depth)
    i64 base = 0;             // rounds below are committed and gone
    i64 startFirstIn = 0;     // firstSequence of round `base` (the committed state): the chain can always restart here
    uint64_t epoch = 0;       // bumped whenever an ignore flag is set
    i64 epochMinId = -1;      // smallest read id flagged in the last bump(s) while a compute was running
    bool stop = false;
    std::thread th;
    Impl(ReadSet& r, const OverlapParams& pp, const double* v, bool t, dp_ctx* sc)
        : reads(r), p(pp), values(v), threaded(t), selCtx(sc), index(pp.k) {}
};

Planner::Planner(ReadSet& reads, const OverlapParams& p, const double* values, bool threaded, dp_ctx* selCtx)
    : d(new Impl(reads, p, values, threaded, selCtx)) {
    if (threaded) d->th = std::thread([this] { threadMain(); });
}

Planner::~Planner() {
    {
        std::lock_guard<std::mutex> lk(d->mu);
        d->stop = true;
    }
    d->cv.notify_all();
    if (d->th.joinable()) d->th.join();
}

std::shared_ptr<RoundPlan> Planner::compute(i64 round, i64 firstIn) {
    const double tc0 = now();
    struct Tick {
        double t0, c0;
        ~Tick() {
            g_prof.planComputes++;
            g_prof.planUs += (long long)((now() - t0) * 1e6);
            g_prof.plannerCpuUs += (long long)((threadCpuNow() - c0) * 1e6);
        }
    } tick{tc0, threadCpuNow()};
    auto plan = std::make_shared<RoundPlan>();
    plan->round = round;
    plan->firstIn = firstIn;
    d->index.reset();
    // the device path needs the window in the resident (cached-view) form and at most 64 list slots
    dp_ctx* sel = (d->selCtx && d->p.numSeeds <= 64) ? d->selCtx : nullptr;
    Overlapper lap(sel, d->reads, d->index, d->p.chunkSize, d->p.numWorkers, d->p.overlapSize, d->p.numSeeds, d->p.minHits);
    const int nw = lap.PrepareQueries(d->p.numSeeds, d->p.seedBatchSize, d->values, firstIn, d->p.queryBatchSize, d->p.queryType);
    if (nw < 0) {
        plan->error = lap.err;
        plan->failed = true;
    }
    plan->empty = nw <= 0;
    plan->windows = lap.windows();
    plan->seedMap = d->index.seedMap;
    // firstSequence = max query SequenceID + 1 (commands/overlap.go:135-142); windows are in ascending read order
    plan->firstOut = nw ? (i64)plan->windows.back().read + 1 : firstIn;
    return plan;
}

void Planner::threadMain() {
    std::unique_lock<std::mutex> lk(d->mu);
    for (;;) {
        if (d->stop) return;
        // next plan of the chain that is missing
        i64 m = d->base;
        i64 firstIn = -1;
        bool can = false;
        while (m <= d->wantUpTo) {
            auto it = d->cache.find(m);
            if (it == d->cache.end()) {
                if (m == d->base) {  // first uncommitted round: its firstSequence is the committed state
                    firstIn = d->startFirstIn;
                    can = true;
                } else {
                    auto pr = d->cache.find(m - 1);
                    if (pr != d->cache.end() && !pr->second->empty) {
                        firstIn = pr->second->firstOut;
                        can = true;
                    }
                }
                break;
            }
            if (it->second->empty) break;  // chain ends here
            m++;
        }
        if (!can) {
            d->cv.wait(lk);
            continue;
        }
        const uint64_t e0 = d->epoch;
        d->epochMinId = -1;
        lk.unlock();
        static const bool dbg = getenv("DPH_DEBUG_PLANNER") != nullptr;
        if (dbg) fprintf(stderr, "[planner] computing plan %lld (firstIn %lld, wantUpTo %lld, base %lld)\n", (long long)m, (long long)firstIn, (long long)d->wantUpTo, (long long)d->base);
        std::shared_ptr<RoundPlan> plan = compute(m, firstIn);
        if (dbg) fprintf(stderr, "[planner] plan %lld done: %zu windows, %zu seeds, empty %d failed %d\n", (long long)m, plan->windows.size(), plan->seedMap.size(), (int)plan->empty, (int)plan->failed);
        lk.lock();
        if (d->stop) return;
        // discard if flags that could matter changed meanwhile, or if the chain below was invalidated
        bool ok = true;
        if (d->epoch != e0 && d->epochMinId >= 0 && d->epochMinId >= firstIn) ok = false;
        if (m > d->base) {
            auto pr = d->cache.find(m - 1);
            if (pr == d->cache.end() || pr->second->firstOut != firstIn) ok = false;
        } else if (m == d->base) {
            if (firstIn != d->startFirstIn) ok = false;
        } else {
            ok = false;  // committed meanwhile
        }
        if (ok && !d->cache.count(m)) d->cache[m] = plan;
        else g_prof.planDiscarded++;
        d->cv.notify_all();
    }
}

std::shared_ptr<const RoundPlan> Planner::get(i64 round) {
    std::unique_lock<std::mutex> lk(d->mu);
    if (!d->threaded) {  // inline chain: compute the first missing plan until `round` is there (or the chain has ended)
        for (;;) {
            auto it = d->cache.find(round);
            if (it != d->cache.end()) return it->second;
            i64 m = d->base, firstIn = d->startFirstIn;
            for (;;) {
                auto e = d->cache.find(m);
                if (e == d->cache.end()) break;
                if (e->second->empty) return e->second;
                firstIn = e->second->firstOut;
                m++;
            }
            if (m > round) return nullptr;
            lk.unlock();
            auto plan = compute(m, firstIn);
            lk.lock();
            d->cache[m] = plan;
        }
    }
    const i64 depth = 6;
    if (round + depth > d->wantUpTo) d->wantUpTo = round + depth;
    d->cv.notify_all();
    for (;;) {
        auto it = d->cache.find(round);
        if (it != d->cache.end()) return it->second;
        // the chain may have ended before `round`
        for (i64 m = d->base; m < round; m++) {
            auto e = d->cache.find(m);
            if (e == d->cache.end()) break;
            if (e->second->empty) return e->second;
        }
        d->cv.wait(lk);
    }
}

i64 Planner::applyIgnores(const std::vector<int>& ids, i64 committedRound) {
    std::lock_guard<std::mutex> lk(d->mu);
    i64 maxNew = -1, minNew = -1;
    for (int id : ids) {
        if (!d->reads.ignore[(size_t)id]) {
            d->reads.ignore[(size_t)id] = 1;
            if (id > maxNew) maxNew = id;
            if (minNew < 0 || id < minNew) minNew = id;
        }
    }
    if (maxNew < 0) return -1;
    d->epoch++;
    if (d->epochMinId < 0 || minNew < d->epochMinId) d->epochMinId = minNew;
    // every cached plan of a later round that starts at or before a newly flagged read may change
    i64 firstBad = -1;
    for (auto it = d->cache.begin(); it != d->cache.end(); ++it) {
        if (it->first > committedRound && it->second->firstIn <= maxNew) {
            firstBad = it->first;
            break;
        }
    }
    if (firstBad >= 0) {
        auto lb = d->cache.lower_bound(firstBad);
        g_prof.planErased += (long long)std::distance(lb, d->cache.end());
        d->cache.erase(lb, d->cache.end());
    }
    d->cv.notify_all();
    return firstBad;
}

uint64_t Planner::ignoreEpoch() {
    std::lock_guard<std::mutex> lk(d->mu);
    return d->epoch;
}

void Planner::dropBefore(i64 round, i64 firstInOfRound) {
    std::lock_guard<std::mutex> lk(d->mu);
    // `round` is the first uncommitted round and starts at firstInOfRound whatever this planner has cached (in a
    // multi-rank run other ranks executed rounds this planner never looked at)
    d->cache.erase(d->cache.begin(), d->cache.lower_bound(round));
    d->base = round;
    d->startFirstIn = firstInOfRound;
    auto it = d->cache.find(round);
    if (it != d->cache.end() && it->second->firstIn != firstInOfRound) d->cache.erase(it, d->cache.end());  // stale chain
    d->cv.notify_all();
}

// ---------------------------------------------------------------------------------------------------------------
// commands/overlap.go Run :96-195

OverlapRun::~OverlapRun() { shutdown(); }

void OverlapRun::HugeTable::assign(const double* src, size_t count) {
    clear();
    const size_t huge = (size_t)2 << 20;
    bytes = (count * sizeof(double) + huge - 1) / huge * huge;
    void* m = mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) throw std::bad_alloc();
    base_ = m;
    mapped_ = bytes + huge;
    p = (double*)(((uintptr_t)m + huge - 1) / huge * huge);
    madvise(p, bytes, MADV_HUGEPAGE);  // advisory: falls back to small pages silently
    memcpy(p, src, count * sizeof(double));
    n = count;
}

void OverlapRun::HugeTable::clear() {
    if (base_) munmap(base_, mapped_);
    base_ = nullptr;
    p = nullptr;
    n = bytes = mapped_ = 0;
}

void OverlapRun::shutdown() {
    {
        std::lock_guard<std::mutex> lk(pmu_);
        stopWorkers_ = true;
    }
    cvWork_.notify_all();
    for (auto& t : workers_)
        if (t.joinable()) t.join();
    workers_.clear();
    ready_.clear();
    redo_.clear();
    if (planner) g_prof.print();
    planner.reset();
    if (plannerCtx) dp_ctx_destroy(plannerCtx);
    plannerCtx = nullptr;
    for (auto& sl : slots) {
        sl->lap.reset();
        sl->index.reset();
        if (sl->ownsCtx && sl->ctx) dp_ctx_destroy(sl->ctx);
        sl->ctx = nullptr;
    }
    slots.clear();
}

int OverlapRun::init(dp_ctx* c, ReadSet* r, const OverlapParams& params, const double* valuesOrNull, int nSlots) {
    ctx = c;
    reads = r;
    p = params;
    reads->himem = p.himem;
    char line[160];
    snprintf(line, sizeof line, "Counting all %d-mers in the input...\n", p.k);
    errText += line;
    if (valuesOrNull) {
        values.assign(valuesOrNull, (size_t)1 << (2 * p.k));
    } else {
        std::vector<uint64_t> counts((size_t)1 << (2 * p.k));
        int rc = dp_kmer_histogram(ctx, p.k, counts.data());  // KmerOccurrences on the GPU
        if (rc != 0) {
            error = dp_last_error(ctx);
            return rc;
        }
        std::vector<double> v = kmerValuesFromCounts(counts, p.k);
        values.assign(v.data(), v.size());
    }
    errText += "Counting complete. Starting indexing and querying...";
    {
        const char* hostsel = getenv("DP_HOST_SELECT");
        if (!(hostsel && hostsel[0] == '1') && p.numSeeds <= 64) {  // value table resident for dp_select_seeds
            int rc = dp_values_upload(ctx, values.data(), values.size());
            if (rc != 0) {
                error = dp_last_error(ctx);
                return rc;
            }
        }
    }
    slots.clear();
    setHostThreadShare((unsigned)std::max(1, nSlots));
    for (int i = 0; i < std::max(1, nSlots); i++) {
        std::unique_ptr<ExecSlot> sl(new ExecSlot());
        if (i == 0) {
            sl->ctx = ctx;
        } else {
            int rc = dp_ctx_create_shared(ctx, &sl->ctx);
            if (rc != 0) {
                error = dp_last_error(nullptr);
                return rc;
            }
            sl->ownsCtx = true;
        }
        sl->index.reset(new SeedIndex(p.k));
        slots.push_back(std::move(sl));
    }
    const char* nothread = getenv("DP_NO_PLANNER_THREAD");
    const char* hostsel = getenv("DP_HOST_SELECT");  // 1: keep the speculative seed selection on the host threads
    if (!(hostsel && hostsel[0] == '1') && p.numSeeds <= 64) {
        int rc = dp_ctx_create_shared(ctx, &plannerCtx);
        if (rc != 0) {
            error = dp_last_error(nullptr);
            return rc;
        }
    }
    planner.reset(new Planner(*reads, p, values.data(), !(nothread && nothread[0] == '1'), plannerCtx));
    firstSequence = 0;
    round = 0;
    done = false;
    stopWorkers_ = issueEnd_ = draining_ = false;
    nextIssue_ = 0;
    inflight_ = workerRc_ = 0;
    flagRound_.assign(reads->size(), -1);
    shardLo = 0;
    shardHi = reads->size();
    return 0;
}

// seeds.NewSeedIndex + the plan's seeds on host and device; queries are built after the scan
int OverlapRun::beginRound(ExecSlot& sl, const RoundPlan& plan) {
    const double tb0 = now();
    sl.index->reset();  // seeds.NewSeedIndex(k) per round (:125) — sparse reset instead of reallocating 4^k tables
    for (uint32_t km : plan.seedMap) sl.index->addSeedKmer(km);
    sl.index->buildRcTable();
    const double tb1 = now();
    g_prof.add(0, tb1 - tb0);
    sl.lap.reset(new Overlapper(sl.ctx, *reads, *sl.index, p.chunkSize, p.numWorkers, p.overlapSize, p.numSeeds, p.minHits));
    sl.lap->setWindows(plan.windows);
    sl.lap->setIgnoreView(reads->ignore.data(), planner->ignoreEpoch());
    const double tb2 = now();
    g_prof.add(1, tb2 - tb1);
    int rc = dp_round_begin(sl.ctx, p.k, sl.index->seedMap.data(), (uint32_t)sl.index->seedMap.size());
    if (rc != 0) {
        sl.error = dp_last_error(sl.ctx);
        return rc;
    }
    g_prof.add(2, now() - tb2);
    return 0;
}

int OverlapRun::finishRound(ExecSlot& sl, const Survivors& all, RoundResult& out) {
    RoundStats& st = out.st;
    double t0 = now();
    int rc = sl.lap->IndexSurvivors(all, st);
    if (rc != 0) {
        sl.error = sl.lap->err;
        return rc;
    }
    out.indexedReads = all.read;
    out.numQuerySeqs = 0;
    for (const SeedQuery& q : sl.lap->queries)
        if (q.ID >= out.numQuerySeqs) out.numQuerySeqs = q.ID + 1;
    double t1 = now();
    st.t_index = t1 - t0;
    std::vector<SeedMatch*>& matches = sl.matches;
    rc = sl.lap->FindOverlaps(sl.matchPool, matches, st);
    if (rc != 0) {
        sl.error = sl.lap->err;
        return rc;
    }
    double t2 = now();
    st.t_query = t2 - t1;
    finalCheck(sl.index->arena, *sl.index, *reads, matches, out.numQuerySeqs, p.overlapSize, out.paf, out.fs, &out.ignores);
    st.n_paf = out.fs.lines;
    st.t_consensus = now() - t2;
    const int k = p.k;
    // algorithmic bytes of the scan: packed bytes of the scanned items + bit table + 8 B per hit
    st.scan_bytes = st.scan_bases / 4 + (((uint64_t)1 << (2 * k)) / 8) + 8 * st.n_hits;
    st.count_bytes = st.scan_bases / 4 + (((uint64_t)1 << (2 * k)) / 8);
    return 0;
}

int OverlapRun::executeRoundOn(ExecSlot& sl, i64 r, RoundResult& out) {
    out = RoundResult();
    out.round = r;
    double t0 = now();
    const double tc0 = g_prof.on ? threadCpuNow() : 0;
    struct SlotCpu {
        double t0;
        ~SlotCpu() {
            if (g_prof.on) g_prof.slotCpuUs += (long long)((threadCpuNow() - t0) * 1e6);
        }
    } slotCpu{tc0};
    static const bool dbgExec = getenv("DPH_DEBUG_PLANNER") != nullptr;
    if (dbgExec) fprintf(stderr, "[exec] round %lld waiting for its plan\n", (long long)r);
    std::shared_ptr<const RoundPlan> plan = planner->get(r);
    if (dbgExec) fprintf(stderr, "[exec] round %lld got plan\n", (long long)r);
    g_prof.getWaitUs += (long long)((now() - t0) * 1e6);
    if (plan && plan->failed) {
        sl.error = "seed selection failed: " + plan->error;
        return -1;
    }
    if (!plan || plan->empty || plan->round != r) {
        out.empty = true;
        if (plan) out.firstIn = out.firstOut = plan->firstOut;
        return 0;
    }
    out.empty = false;
    out.firstIn = plan->firstIn;
    out.firstOut = plan->firstOut;
    out.queryReads.reserve(plan->windows.size());
    for (const auto& w : plan->windows) out.queryReads.push_back(w.read);
    int rc = beginRound(sl, *plan);
    if (rc) return rc;
    out.st.n_seeds = plan->seedMap.size();
    double t1 = now();
    out.st.t_prepare = t1 - t0;
    rc = sl.lap->ScanLocal(0, reads->size(), sl.local, out.st);
    if (rc != 0) {
        sl.error = sl.lap->err;
        return rc;
    }
    out.st.t_scan = now() - t1;
    rc = finishRound(sl, sl.local, out);
    if (dbgExec) fprintf(stderr, "[exec] round %lld finished rc %d\n", (long long)r, rc);
    const double t2 = now();
    g_prof.add(14, t2 - t0);
    g_prof.add(15, (t2 - t0) - (out.st.t_prepare + out.st.t_scan + out.st.t_index + out.st.t_query + out.st.t_consensus));
    return rc;
}

int OverlapRun::executeRound(i64 r, RoundResult& out) {
    int rc = executeRoundOn(*slots[0], r, out);
    if (rc) error = slots[0]->error;
    return rc;
}

int OverlapRun::executeRounds(const std::vector<i64>& rounds, std::vector<RoundResult>& outs) {
    const size_t n = std::min(rounds.size(), slots.size());
    outs.assign(n, RoundResult());
    std::vector<int> rcs(n, 0);
    if (n == 1) {
        rcs[0] = executeRoundOn(*slots[0], rounds[0], outs[0]);
    } else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; i++) th.emplace_back([&, i] { rcs[i] = executeRoundOn(*slots[i], rounds[i], outs[i]); });
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < n; i++)
        if (rcs[i]) {
            error = slots[i]->error;
            return rcs[i];
        }
    return 0;
}

void OverlapRun::commitOne(RoundResult& r) {
    char line[200];
    firstSequence = r.firstOut;
    numQuerySeqs = r.numQuerySeqs;
    if (round == 0)
        snprintf(line, sizeof line, "Using query sets of around %lld sequences against %lld sequences.\n", (long long)firstSequence,
                 (long long)reads->size());
    else
        snprintf(line, sizeof line, "Using query set with %lld  sequences starting from %lld sequences against %lld sequences.\n",
                 (long long)numQuerySeqs, (long long)firstSequence, (long long)reads->size());
    errText += line;
    snprintf(line, sizeof line, "Total %lld hits across %lld overlaps.\n", (long long)r.fs.hits, (long long)r.fs.qHits);
    errText += line;
    badBack += r.fs.badBack;
    emptyMatch += r.fs.emptyMatch;
    paf += r.paf;
    last = r.st;
    g_prof.ignores += (long long)r.ignores.size();
    for (int id : r.ignores)
        if (!reads->ignore[(size_t)id] && flagRound_[(size_t)id] < 0) flagRound_[(size_t)id] = (int32_t)round;
    planner->applyIgnores(r.ignores, round);
    round++;
    planner->dropBefore(round, firstSequence);
}

int OverlapRun::commitResults(std::vector<RoundResult>& results) {
    paf.clear();
    int committed = 0;
    std::vector<uint8_t> newly;  // flags set by rounds committed in THIS call
    std::vector<int> newIds;
    for (RoundResult& r : results) {
        if (r.round != round) break;
        if (r.empty) {
            done = true;
            break;
        }
        // speculation check: the round ran against the flags at the start of this batch.  It is exact iff no read
        // flagged by the earlier rounds of the batch could have been one of its queries (id >= firstIn) or entered
        // its index.
        bool ok = true;
        if (!newIds.empty()) {
            for (int id : newIds)
                if (id >= r.firstIn) {
                    ok = false;
                    break;
                }
            if (ok) {
                if (newly.empty()) {
                    newly.assign(reads->size(), 0);
                    for (int id : newIds) newly[(size_t)id] = 1;
                }
                for (uint32_t rd : r.indexedReads)
                    if (newly[rd]) {
                        ok = false;
                        break;
                    }
            }
        }
        if (!ok) break;
        for (int id : r.ignores)
            if (!reads->ignore[(size_t)id]) {
                newIds.push_back(id);
                if (!newly.empty()) newly[(size_t)id] = 1;
            }
        commitOne(r);
        committed++;
    }
    return committed;
}

void OverlapRun::startWorkers() {
    if (!workers_.empty()) return;
    nextIssue_ = round;
    for (size_t i = 0; i < slots.size(); i++) workers_.emplace_back([this, i] { workerMain(i); });
}

void OverlapRun::workerMain(size_t si) {
    ExecSlot& sl = *slots[si];
    const i64 window = (i64)slots.size() + 2;  // rounds issued ahead of the commit point
    std::unique_lock<std::mutex> lk(pmu_);
    for (;;) {
        cvWork_.wait(lk, [&] {
            return stopWorkers_ || (!draining_ && workerRc_ == 0 && (!redo_.empty() || (!issueEnd_ && nextIssue_ < round + window)));
        });
        if (stopWorkers_) return;
        i64 r;
        if (!redo_.empty()) {
            r = redo_.front();
            redo_.pop_front();
        } else {
            r = nextIssue_++;
        }
        const i64 snap = round;
        inflight_++;
        lk.unlock();
        RoundResult res;
        const int rc = executeRoundOn(sl, r, res);
        res.snapshot = snap;
        res.round = r;
        lk.lock();
        inflight_--;
        if (rc != 0 && workerRc_ == 0) {
            workerRc_ = rc;
            workerErr_ = sl.error;
        }
        if (rc == 0) {
            if (res.empty) issueEnd_ = true;  // flags only accumulate: an exhausted input stays exhausted
            if (!draining_) ready_[r] = std::move(res);
        }
        cvDone_.notify_all();
    }
}

// A round executed against a flag snapshot is exact iff no read flagged since then is one of its queries or entered
// its index, and its plan continues the committed chain.
bool OverlapRun::resultValid(const RoundResult& r) const {
    if (r.firstIn != firstSequence) return false;
    for (uint32_t id : r.queryReads)
        if (flagRound_[id] >= r.snapshot) return false;
    for (uint32_t id : r.indexedReads)
        if (flagRound_[id] >= r.snapshot) return false;
    return true;
}

int OverlapRun::step() {
    if (done) return 0;
    startWorkers();
    const double t0 = now();
    std::unique_lock<std::mutex> lk(pmu_);
    paf.clear();
    int committed = 0;
    for (;;) {
        if (workerRc_ != 0) {
            error = workerErr_;
            return workerRc_;
        }
        auto it = ready_.find(round);
        if (it == ready_.end()) {
            if (committed) break;
            cvDone_.wait(lk);
            continue;
        }
        RoundResult res = std::move(it->second);
        ready_.erase(it);
        if (res.empty) {
            done = true;
            break;
        }
        g_prof.executed++;
        if (!resultValid(res)) {
            g_prof.rejected++;
            redo_.push_back(round);
            cvWork_.notify_all();
            continue;
        }
        commitOne(res);
        committed++;
        g_prof.committed++;
        cvWork_.notify_all();
    }
    g_prof.execUs += (long long)((now() - t0) * 1e6);
    return committed;
}

void OverlapRun::drain() {
    std::unique_lock<std::mutex> lk(pmu_);
    if (workers_.empty()) return;
    draining_ = true;
    cvDone_.wait(lk, [&] { return inflight_ == 0; });
    ready_.clear();
    redo_.clear();
    nextIssue_ = round;
    issueEnd_ = false;
    draining_ = false;
    cvWork_.notify_all();
}

// ---- scan-shard mode: plan + local scan, then (after the survivor exchange) the rest of the round
int OverlapRun::roundPrepareAndScan() {
    if (done) return 0;
    cur = RoundResult();
    cur.round = round;
    double t0 = now();
    curPlan = planner->get(round);
    if (curPlan && curPlan->failed) {
        error = "seed selection failed: " + curPlan->error;
        return -1;
    }
    if (!curPlan || curPlan->empty) {
        done = true;
        return 0;
    }
    cur.empty = false;
    cur.firstIn = curPlan->firstIn;
    cur.firstOut = curPlan->firstOut;
    ExecSlot& sl = *slots[0];
    int rc = beginRound(sl, *curPlan);
    if (rc) {
        error = sl.error;
        return rc;
    }
    cur.st.n_seeds = curPlan->seedMap.size();
    double t1 = now();
    cur.st.t_prepare = t1 - t0;
    rc = sl.lap->ScanLocal(shardLo, shardHi, sl.local, cur.st);
    if (rc != 0) {
        error = sl.lap->err;
        return rc;
    }
    cur.st.t_scan = now() - t1;
    return 1;
}

int OverlapRun::roundFinish(const Survivors& all) {
    int rc = finishRound(*slots[0], all, cur);
    if (rc) {
        error = slots[0]->error;
        return rc;
    }
    paf.clear();
    commitOne(cur);
    return 0;
}

}  // namespace dph
