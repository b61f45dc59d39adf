// Product host side: overlap.Overlapper (overlap/overlap.go) — PrepareQueries, AddSequences (scan + chunking + index),
// FindOverlaps — and finalCheckWorker (commands/overlap.go:197-233) above the C ABI.  Canonical single-worker order
// (DESIGN.md 2).  The round loop lives in host_pipeline.cpp, the worker pool in host_pool.cpp.
#include <cstring>
#include <mutex>
#include <thread>

#include "host_util.hpp"

namespace dph {

Overlapper::Overlapper(dp_ctx* ctx, ReadSet& reads, SeedIndex& index, i64 chunkSize, int, i64 overlap, int minSeeds,
                       double hitFraction)
    : ctx_(ctx), reads_(reads), index_(index), chunkSize_(chunkSize), overlap_(overlap), minSeeds_(minSeeds),
      hitFraction_(hitFraction) {
    ignore_ = reads.ignore.data();
}

// PrepareQueries :157-214 with getEdges :55-89 (QueryEdges): seed selection is sequential and stays on the host.
// Returns the number of query windows.
// PrepareQueries for QueryEdges from the window cache: per window the evaluated k-mers are probed against the seeds
// committed so far; untouched, the cached selection is what AddSeeds would pick and is committed as is; touched, AddSeeds
// runs for real (the block that contains the seed is abandoned, seeds.go:94-97, and the walk goes on from there).
int Overlapper::prepareFromCache(int numSeeds, i64 seedLimit, ValueView values, i64 firstSequence, i64 maxSeqs) {
    if (firstSequence != 0 && firstSequence >= (i64)reads_.size()) return 0;  // seqio.go:279
    const double t0 = now();
    std::vector<uint32_t> tmp((size_t)numSeeds);
    i64 sent = 0;
    const bool prof = g_prof.on;
    unsigned long long cTouch = 0, cResel = 0, cCommit = 0, c0 = prof ? __rdtsc() : 0;
    const unsigned long long cStart = c0;
    const uint32_t *runSpec = nullptr, *runKmers = nullptr;  // the cache's block of windows [runLo, runHi) (one lock per block, not per window)
    uint32_t runLo = 0, runHi = 0;
    for (size_t r = (size_t)firstSequence; r < reads_.size() && sent < maxSeqs; r++) {
        if (flagLoad(ignore_ + r)) continue;
        sent++;
        if (index_.size() >= seedLimit) break;  // the budget is tested once per read, before its first window (overlap.go:57-60)
        for (uint32_t w = cache_->first[r]; w < cache_->first[r + 1]; w++) {
            const WindowCache::Win& win = cache_->wins[w];
            if (w < runLo || w >= runHi) {
                if (!cache_->getRun(w, &runSpec, &runKmers, &runHi, &err)) return -1;
                runLo = w;
            }
            const uint32_t* spec = runSpec + (size_t)(w - runLo) * (size_t)cache_->numSeeds;
            const uint32_t* kmers = runKmers + (size_t)(w - runLo) * (size_t)cache_->stride;
            if (prof) c0 = __rdtsc();
            const bool touched = index_.touchesSeed(kmers, cache_->stride);
            if (prof) {
                const unsigned long long c1 = __rdtsc();
                cTouch += c1 - c0;
                c0 = c1;
            }
            if (touched) {
                const uint8_t* q = reads_.quality(win.read);
                index_.selectSeeds(reads_.seq(win.read) + win.start, win.len, numSeeds, values, tmp.data(), true, q ? q + win.start : nullptr);
                if (prof) {
                    const unsigned long long c1 = __rdtsc();
                    cResel += c1 - c0;
                    c0 = c1;
                }
                index_.commitSeeds(tmp.data(), numSeeds);
                g_prof.reselected++;
            } else {
                index_.commitSeeds(spec, numSeeds);
            }
            if (prof) cCommit += __rdtsc() - c0;
            windows_.push_back({win.read, win.start, win.len});
        }
    }
    if (prof) {
        const unsigned long long all = __rdtsc() - cStart;
        g_prof.planTouchCyc += (long long)cTouch;
        g_prof.planReselCyc += (long long)cResel;
        g_prof.planCommitCyc += (long long)cCommit;
        g_prof.planOtherCyc += (long long)(all - cTouch - cResel - cCommit);
    }
    g_prof.add(17, now() - t0);
    return (int)windows_.size();
}

int Overlapper::PrepareQueries(int numSeeds, i64 seedLimit, ValueView values, i64 firstSequence, i64 maxSeqs, int queryType) {
    windows_.clear();
    queries.clear();
    assembled_ = false;
    const bool weightSides = (queryType & 8) != 0;  // WeightEdges: seeds come from the two 200-base sides of a window
    if (weightSides) numSeeds /= 2;                  // overlap.go:161-163
    if (numSeeds < 1) return 0;
    if (cache_ && (queryType & 1) && !weightSides && numSeeds == cache_->numSeeds) return prepareFromCache(numSeeds, seedLimit, values, firstSequence, maxSeqs);
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
            if (flagLoad(ignore_ + rNext)) continue;
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
                const uint8_t* q = reads_.quality(c.read);
                index_.selectSeeds(reads_.seq(c.read) + c.start, c.len, numSeeds, values, &spec[w * (size_t)numSeeds], false,
                                   q ? q + c.start : nullptr);
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
                const uint8_t* q = reads_.quality(sw.read);
                index_.selectSeeds(s, sw.len, numSeeds, values, tmp.data(), true, q ? q + sw.start : nullptr);
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
    dp_survivor_batch& b = lastScan_;
    const double ts0 = now();
    dp_scan_fetch_mode(ctx_, deviceChunkWanted_ ? 1 : 0);
    // the chunk stage of IndexSurvivors goes behind the scan's own kernels (dp_index_prechain) when this context will chunk its own
    // survivors on the device; survivors gathered from other ranks are chunked after the exchange, as before
    if (deviceChunkWanted_ && minSeeds_ > 5 && prechainOk_) dp_index_prechain(ctx_, chunkSize_, overlap_, (uint32_t)minSeeds_, (int32_t)reads_.servedInset());
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
    st.idx_rounds += b.index_mode;
    st.idx_hits += b.index_hits;
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
    local.segsOnHost = !(deviceChunkWanted_ && b.index_mode);  // (fetch mode 1 is honoured by the index-mode scan only)
    local.chunkCtx = ctx_;
    lastLocal_ = &local;
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

int Overlapper::ExchangeSurvivors(dp_comm* comm, Survivors& all) {
    dp_survivor_batch g;
    const int rc = dp_allgather_survivors(comm, ctx_, &lastScan_, &g);
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    all.read.assign(g.read, g.read + g.n_survivors);
    all.n_seeds.assign(g.n_seeds, g.n_seeds + g.n_survivors);
    all.seg_off.assign(g.seg_off, g.seg_off + g.n_survivors);
    uint64_t survEnd = 0;
    if (g.n_survivors) survEnd = g.seg_off[g.n_survivors - 1] + 2ull * g.n_seeds[g.n_survivors - 1] + 1;
    all.seg_off.push_back(survEnd);
    all.segs.clear();
    all.segsView = g.segs;
    all.segsViewLen = survEnd;
    all.deviceResident = true;  // the library installed the gathered array as the context's scan output
    // ... and the gathered survivor list next to it: chunking and indexing run on the device as for a single GPU; with
    // dp_scan_fetch_mode(1) the survivors' segments never came to the host (only the query windows' are valid in g.segs)
    all.chunkCtx = ctx_;
    all.segsOnHost = !deviceChunkWanted_;
    winSegs_.clear();
    winOff_.assign(1, 0);
    for (uint32_t i = 0; i < g.n_extra; i++) {
        const uint64_t o = g.extra_seg_off[i];
        winSegs_.insert(winSegs_.end(), g.segs + o, g.segs + o + 2ull * g.extra_n_seeds[i] + 1);
        winOff_.push_back(winSegs_.size());
    }
    return 0;
}

// chunkWorker :253-318 for one seed sequence whose segments start at device offset segBase
void Overlapper::chunkAndAdd(SeedSeq* s, uint64_t segBase, Arena& ar, std::vector<SeedSeq*>& seqOut,
                            std::vector<dp_seq_ref>& refOut) {
    const int k = index_.k;
    auto add = [&](SeedSeq* q) {
        seqOut.push_back(q);
        dp_seq_ref r;
        r.seg_off = segBase + (uint64_t)(q->seg - s->seg);
        r.n_seeds = (uint32_t)q->numSeeds();
        r.reserved = 0;
        refOut.push_back(r);
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
    chunksOnDevice_ = false;
    // (minSeeds <= 5: the reference's walk backs up as many seeds as a chunk holds - no bound on the number of chunks to size the
    // device buffers from; the host loop below does what the reference does)
    if (deviceChunkWanted_ && minSeeds_ > 5 && all.deviceResident && all.chunkCtx == ctx_) {
        // A12 + A13 on the device: the survivors of this context's own scan - or the set gathered from every rank - still in
        // its scan buffer
        index_.sequences.clear();
        index_.refs.clear();
        // the queries first (they come from the scan's output, not from the index): announced to the library, they travel to the
        // device with the index build's first launch, and the query stage starts with its kernel (dp_query_prestage)
        const double tq0 = now();
        buildQueries(st);
        assembleQueries();
        const bool prestage = !dph_tune("no_query_prestage", 0) && !dp_index_prechained(ctx_);  // (tests: dp_find_overlaps uploads them itself)  // (chunk stage launched already: nothing to ride on)
        if (prestage && !queries.empty()) {
            int prc = dp_query_prestage(ctx_, querySegs_.data(), queryOff_.data(), (uint32_t)queries.size(), hitFraction_);
            if (prc != 0) {
                err = dp_last_error(ctx_);
                return prc;
            }
        }
        const double tq1 = now();
        uint32_t cap = 0;
        int rc = dp_index_build_chunked(ctx_, chunkSize_, overlap_, (uint32_t)minSeeds_, (int32_t)reads_.servedInset(), (uint32_t)all.read.size(), &cap);
        if (rc != 0) {
            err = dp_last_error(ctx_);
            return rc;
        }
        chunksOnDevice_ = true;
        nIndexedCap_ = cap;
        nIndexedExact_ = 0;
        st.n_indexed = cap;  // (an upper bound until the consensus call reports the exact number)
        g_prof.add(5, 0);
        g_prof.add(6, 0);
        g_prof.add(7, now() - tq1 + (tq0 - tp0));
        g_prof.add(8, tq1 - tq0);
        return 0;
    }
    const int32_t* fetched = nullptr;
    if (!all.segsOnHost) {  // (left on the device by the scan / the exchange, but this round is chunked on the host after all)
        uint64_t n = 0;
        int rc = dp_scan_fetch_segments(ctx_, &fetched, &n);
        if (rc != 0) {
            err = dp_last_error(ctx_);
            return rc;
        }
    }
    allSegs_ = fetched ? fetched : all.segData();  // (same layout: the context's scan output is exactly this survivor array)
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
    auto chunkRange = [&](size_t lo, size_t hi, Arena& ar, std::vector<SeedSeq*>& seqOut, std::vector<dp_seq_ref>& refOut) {
        for (size_t i = lo; i < hi; i++) {
            const uint32_t r = all.read[i];
            SeedSeq* s = ar.make();
            s->seg = allSegs_ + all.seg_off[i];
            s->n = (int)(all.seg_off[i + 1] - all.seg_off[i]);
            s->id = (int)r;
            s->length = reads_.length(r);
            s->offset = 0;
            s->inset = reads_.servedInset();
            chunkAndAdd(s, all.seg_off[i], ar, seqOut, refOut);
        }
    };
    const size_t nSurv = all.read.size(), block = 1024;
    if (nSurv <= 2 * block) {
        chunkRange(0, nSurv, index_.arena, index_.sequences, index_.refs);
    } else {  // chunkWorker is per sequence: blocks of survivors on the worker pool, results concatenated in file order
        const size_t nb = (nSurv + block - 1) / block;
        if (index_.chunkArenas.size() < nb) index_.chunkArenas.resize(nb);
        std::vector<std::vector<SeedSeq*>> seqB(nb);
        std::vector<std::vector<dp_seq_ref>> refB(nb);
        parallelFor(nb, [&](size_t b) {
            index_.chunkArenas[b].clear();
            chunkRange(b * block, std::min(nSurv, (b + 1) * block), index_.chunkArenas[b], seqB[b], refB[b]);
        });
        size_t total = 0;
        for (auto& v : seqB) total += v.size();
        index_.sequences.reserve(total);
        index_.refs.reserve(total);
        for (size_t b = 0; b < nb; b++) {
            index_.sequences.insert(index_.sequences.end(), seqB[b].begin(), seqB[b].end());
            index_.refs.insert(index_.refs.end(), refB[b].begin(), refB[b].end());
        }
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
    buildQueries(st);
    g_prof.add(8, now() - tp3);
    return 0;
}

// queries: [fwd, rc] per window (PrepareQueries :189-201), from the windows' scan output
// the queries' segment arrays in one block, as dp_find_overlaps / dp_query_prestage take them
void Overlapper::assembleQueries() {
    if (assembled_) return;
    assembled_ = true;
    querySegs_.clear();
    queryOff_.assign(1, 0);
    for (const SeedQuery& q : queries) {
        querySegs_.insert(querySegs_.end(), q.Query->seg, q.Query->seg + q.Query->n);
        queryOff_.push_back(querySegs_.size());
    }
}

void Overlapper::buildQueries(RoundStats& st) {
    assembled_ = false;
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
}

// the chunks dp_index_build_chunked made, as the host objects the host consensus path works on (windows the device flags)
int Overlapper::materializeChunks() {
    if (!chunksOnDevice_ || !index_.sequences.empty()) return 0;
    uint32_t n = 0;
    std::vector<dp_seq_ref> refs(nIndexedCap_);
    std::vector<dp_seq_meta> metas(nIndexedCap_);
    int rc = dp_index_chunks(ctx_, refs.data(), metas.data(), nIndexedCap_, &n);
    const int32_t* segs = nullptr;
    uint64_t nsegs = 0;
    if (rc == 0) rc = dp_scan_fetch_segments(ctx_, &segs, &nsegs);
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    allSegs_ = segs;
    index_.sequences.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        SeedSeq* s = index_.arena.make();
        s->seg = segs + refs[i].seg_off;
        s->n = (int)(2 * refs[i].n_seeds + 1);
        s->id = (int)metas[i].read;
        s->length = metas[i].length;
        s->offset = metas[i].offset;
        s->inset = metas[i].inset;
        const i64 L = reads_.length(metas[i].read);
        if (!(s->offset == 0 && s->inset == reads_.servedInset() && s->length == L)) {  // a chunk: its parent is the read's view
            SeedSeq* root = index_.arena.make();
            root->id = s->id;
            root->length = L;
            root->offset = 0;
            root->inset = reads_.servedInset();
            s->parent = root;
        }
        index_.sequences.push_back(s);
    }
    return 0;
}

// FindOverlaps :320 + matchWorker :346
int Overlapper::FindOverlaps(std::vector<SeedMatch>& pool, std::vector<SeedMatch*>& out, RoundStats& st) {
    if (int rcm = materializeChunks()) return rcm;  // (this path works on host objects)
    if (chunksOnDevice_) st.n_indexed = index_.sequences.size();
    assembleQueries();
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
    st.chain_bytes += mb.chain_bytes;
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
        m->anchorFirstB = mb.target_anchor ? mb.target_anchor[2 * (size_t)i] : -1;
        m->anchorLastFromEndB = mb.target_anchor ? mb.target_anchor[2 * (size_t)i + 1] : -1;
        out.push_back(m);
    }
    st.n_matches = out.size();
    g_prof.add(10, now() - tq1);
    return 0;
}

static void finalCheckOne(Arena& arena, const SeedIndex& index, const ReadSet& reads, std::vector<SeedMatch*>& results,
                          i64 overlapSize, std::string& paf, std::vector<int>& ignoreIds, FinalCheckStats& fs);

static inline char* putInt(char* w, i64 v) {  // %d
    char num[24];
    char* e = num + sizeof num;
    char* p = e;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do {
        *--p = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (v < 0) *--p = '-';
    memcpy(w, p, (size_t)(e - p));
    return w + (e - p);
}

int Overlapper::FindOverlapsAndFinalCheck(std::vector<SeedMatch>& pool, i64 overlapSize, std::string& paf, FinalCheckStats& fs,
                                          std::vector<int>* ignoreOut, RoundStats& st, std::shared_ptr<TextJob>* textOut) {
    assembleQueries();
    dp_match_batch mb;
    const double tq0 = now();
    // want_candidates 6: the matches stay on the device and the stage is left pending - dp_consensus_paf below evaluates it in the
    // wait it needs anyway (DP_FIND_PENDING=0: wait here, as every other caller of dp_find_overlaps does)
    static const bool pendingFind = true;
    int rc = dp_find_overlaps(ctx_, querySegs_.data(), queryOff_.data(), (uint32_t)queries.size(), hitFraction_, index_.k,
                              (uint32_t)(overlap_ / 2), pendingFind ? 6 : 2, &mb);
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    const double tq1 = now();
    g_prof.add(9, tq1 - tq0);
    if (!pendingFind) {
        st.k_query_ms += mb.query_kernel_ms;
        st.k_chain_ms += mb.chain_kernel_ms;
        st.query_bytes += mb.query_bytes;
        st.chain_bytes += mb.chain_bytes;
    }
    if ((!chunksOnDevice_ && index_.sequences.empty()) || (chunksOnDevice_ && nIndexedCap_ == 0) || queries.empty())
        return 0;  // nothing indexed: no candidate, no line
    if (index_.rcOf.size() != index_.seedMap.size()) index_.buildRcTable();
    dp_paf_batch pb;
    if (chunksOnDevice_) {  // the chunks' fields are on the device already (dp_index_build_chunked)
        rc = dp_consensus_paf(ctx_, nullptr, 0, index_.rcOf.data(), (uint32_t)index_.rcOf.size(), index_.k, (int)overlapSize, &pb);
    } else {
        // SeedSequence fields of the indexed sequences and the seed -> reverse-complement-seed table of the round
        static thread_local std::vector<dp_seq_meta> metas;
        metas.resize(index_.sequences.size());
        for (size_t i = 0; i < metas.size(); i++) {
            const SeedSeq* s = index_.sequences[i];
            metas[i].read = (uint32_t)s->id;
            metas[i].length = (int32_t)s->length;
            metas[i].offset = (int32_t)s->offset;
            metas[i].inset = (int32_t)s->inset;
        }
        rc = dp_consensus_paf(ctx_, metas.data(), (uint32_t)metas.size(), index_.rcOf.data(), (uint32_t)index_.rcOf.size(), index_.k,
                              (int)overlapSize, &pb);
    }
    if (rc != 0) {
        err = dp_last_error(ctx_);
        return rc;
    }
    if (chunksOnDevice_) {
        nIndexedExact_ = pb.n_indexed;
        st.n_indexed = pb.n_indexed;
    }
    if (pendingFind) {
        st.k_query_ms += pb.query_kernel_ms;
        st.k_chain_ms += pb.chain_kernel_ms;
        st.query_bytes += pb.query_bytes;
        st.chain_bytes += pb.chain_bytes;
    }
    st.k_cons_ms += pb.kernel_ms;
    st.k_index_ms += pb.index_kernel_ms;
    const double tq2 = now();
    g_prof.add(10, tq2 - tq1);
    // windows the device left to the host: BuildConsensus + finalCheckWorker on the fetched matches, one window at a time
    std::vector<std::string> hostPaf;
    std::vector<std::vector<int>> hostIgn;
    std::vector<uint32_t> hostOf(pb.n_groups, 0xffffffffu);
    uint32_t nFlag = 0;
    for (uint32_t g = 0; g < pb.n_groups; g++)
        if (pb.groups[g].flag) hostOf[g] = nFlag++;
    if (nFlag) {
        // (pb points into the context's pinned output, which the calls below leave alone)
        if (int rcm = materializeChunks()) return rcm;
        dp_match_batch fb;
        rc = dp_fetch_overlaps(ctx_, &fb);
        if (rc != 0) {
            err = dp_last_error(ctx_);
            return rc;
        }
        hostPaf.resize(nFlag);
        hostIgn.resize(nFlag);
        std::vector<std::vector<SeedMatch*>> res(nFlag);
        size_t used = 0;
        for (uint32_t i = 0; i < fb.n_matches; i++) {
            const uint32_t g = fb.query[i] / 2;
            if (hostOf[g] == 0xffffffffu) continue;
            if (pool.size() <= used) pool.resize(used + 64);
            used++;
        }
        used = 0;
        for (uint32_t i = 0; i < fb.n_matches; i++) {  // (second pass: pool no longer moves)
            const uint32_t g = fb.query[i] / 2;
            if (hostOf[g] == 0xffffffffu) continue;
            SeedMatch* m = &pool[used++];
            const SeedQuery& q = queries[fb.query[i]];
            m->MatchA.assign(fb.match_a + fb.off[i], fb.match_a + fb.off[i + 1]);
            m->MatchB.assign(fb.match_b + fb.off[i], fb.match_b + fb.off[i + 1]);
            m->SeqA = q.Query;
            m->SeqB = index_.sequences[fb.target[i]];
            m->QueryID = q.ID;
            m->ReverseComplementQuery = q.ReverseComplement;
            m->anchorFirstB = fb.target_anchor ? fb.target_anchor[2 * (size_t)i] : -1;
            m->anchorLastFromEndB = fb.target_anchor ? fb.target_anchor[2 * (size_t)i + 1] : -1;
            res[hostOf[g]].push_back(m);
        }
        Arena local;
        for (uint32_t h = 0; h < nFlag; h++) {
            if (res[h].size() > 1) finalCheckOne(local, index_, reads_, res[h], overlapSize, hostPaf[h], hostIgn[h], fs);
            local.clear();
        }
        g_prof.hostGroups += nFlag;
    }
    // everything but the text: hit counts, diagnostics, SetIgnore ids (query order)
    uint64_t hits = 0, qHits = 0;
    for (uint32_t g = 0; g < pb.n_groups; g++) {
        const dp_group_meta& gm = pb.groups[g];
        hits += gm.n_matches;
        if (gm.n_matches >= 2) qHits++;
        if (gm.flag) {
            for (int id : hostIgn[hostOf[g]]) {
                if (ignoreOut) ignoreOut->push_back(id);
                else flagStore(&reads_.ignore[(size_t)id], 1);
            }
            continue;
        }
        fs.badBack += gm.bad_back;
        fs.emptyMatch += gm.empty_match;
        fs.lines += gm.n_lines;
        st.cons_bytes += gm.reserved;
        for (uint32_t j = 0; j < gm.n_ignore; j++) {
            const int id = (int)pb.ignore_ids[gm.slot + j];
            if (ignoreOut) ignoreOut->push_back(id);
            else flagStore(&reads_.ignore[(size_t)id], 1);
        }
    }
    // the text: one line per part after the first (commands/overlap.go:223-228), windows in query order - here, or on a
    // formatter thread while this slot starts its next round (the records are copied out of the context's pinned buffer)
    {
        size_t nRecs = 0;
        for (uint32_t g = 0; g < pb.n_groups; g++)
            if (!pb.groups[g].flag) nRecs = std::max<size_t>(nRecs, (size_t)pb.groups[g].slot + pb.groups[g].n_lines);
        std::shared_ptr<TextJob> job = std::make_shared<TextJob>();
        job->reads = &reads_;
        job->recs.assign(pb.paf, pb.paf + nRecs);
        job->groups.assign(pb.groups, pb.groups + pb.n_groups);
        job->hostPaf = std::move(hostPaf);
        job->hostOf = std::move(hostOf);
        if (textOut && textPool_) {
            *textOut = job;
            textPool_->submit(job);
        } else {
            job->format();
            paf += job->text;
        }
    }
    fs.hits = hits;
    fs.qHits = qHits;
    st.n_matches = hits;
    g_prof.add(12, now() - tq2);
    return 0;
}

namespace {
std::mutex g_tjb_mu;
std::vector<std::pair<std::vector<dp_paf_rec>, std::vector<dp_group_meta>>> g_tjb_free;
}  // namespace
static bool tjbOn() {
    static const bool on = true;
    return on;
}
void TextJobBuffers::take(std::vector<dp_paf_rec>& recs, std::vector<dp_group_meta>& groups) {
    if (!tjbOn()) return;
    std::lock_guard<std::mutex> lk(g_tjb_mu);
    if (g_tjb_free.empty()) return;
    recs = std::move(g_tjb_free.back().first);
    groups = std::move(g_tjb_free.back().second);
    g_tjb_free.pop_back();
}
void TextJobBuffers::give(std::vector<dp_paf_rec>& recs, std::vector<dp_group_meta>& groups) {
    if (!tjbOn() || (recs.capacity() == 0 && groups.capacity() == 0)) return;
    recs.clear();
    groups.clear();
    std::lock_guard<std::mutex> lk(g_tjb_mu);
    if (g_tjb_free.size() < 32) g_tjb_free.emplace_back(std::move(recs), std::move(groups));
}

namespace {
std::mutex g_txt_mu;
std::vector<std::string> g_txt_free;
size_t g_txt_bytes = 0;  // sum of the kept strings' capacities
}  // namespace
static bool txtOn() {
    static const bool on = true;
    return on;
}
void TextJobBuffers::takeText(std::string& s) {
    if (!txtOn()) return;
    std::lock_guard<std::mutex> lk(g_txt_mu);
    if (g_txt_free.empty()) return;
    s = std::move(g_txt_free.back());
    g_txt_free.pop_back();
    g_txt_bytes -= std::min(g_txt_bytes, s.capacity());
    s.clear();
}
void TextJobBuffers::giveTexts(std::vector<std::string>& v) {
    if (!txtOn()) return;
    std::lock_guard<std::mutex> lk(g_txt_mu);
    // capped by bytes, not by entries: a long-lived embedder keeps at most that much until dph_release_caches().  512 MB: the text
    // of a config-2 job is 235 MB in 599 strings whose capacities add up to a little more - with a cap of 256 MB a tenth of the
    // rounds of every job allocated (and page-faulted) fresh strings on the committing thread, which a 20-job run showed as 0.15 ->
    // 0.18 ms per round (DPH_TEXT_POOL_MB: another cap)
    static const size_t CAP_BYTES = (size_t)std::max(1L, dph_tune("text_pool_mb", 512)) << 20;
    for (std::string& s : v)
        if (s.capacity() >= 65536 && g_txt_bytes + s.capacity() <= CAP_BYTES) {
            g_txt_bytes += s.capacity();
            g_txt_free.push_back(std::move(s));
        }
}
size_t TextJobBuffers::releaseAll() {
    size_t freed = 0;
    {
        std::lock_guard<std::mutex> lk(g_txt_mu);
        freed += g_txt_bytes;
        std::vector<std::string>().swap(g_txt_free);
        g_txt_bytes = 0;
    }
    std::lock_guard<std::mutex> lk(g_tjb_mu);
    for (auto& pr : g_tjb_free) freed += pr.first.capacity() * sizeof(dp_paf_rec) + pr.second.capacity() * sizeof(dp_group_meta);
    decltype(g_tjb_free)().swap(g_tjb_free);
    return freed;
}

void TextJob::format() {
    size_t nLines = 0, hostBytes = 0;
    for (const dp_group_meta& gm : groups) nLines += gm.n_lines;
    for (const std::string& hp : hostPaf) hostBytes += hp.size();
    const ReadSet& rs = *reads;
    const size_t lineCap = 2 * rs.maxNameLen + 7 * 21 + 24;  // two names + 7 numbers + fixed text
    // written into a buffer this thread keeps (sized for the worst case: no per-round allocation, zero fill or page faults of
    // a megabyte), then copied out at its real size
    static thread_local std::vector<char> textBuf;
    if (textBuf.size() < nLines * lineCap + hostBytes + 64) textBuf.resize((nLines * lineCap + hostBytes + 64) * 3 / 2);
    char* const w0 = textBuf.data();
    char* w = w0;
    for (size_t g = 0; g < groups.size(); g++) {
        const dp_group_meta& gm = groups[g];
        if (gm.flag) {
            const std::string& hp = hostPaf[hostOf[g]];
            memcpy(w, hp.data(), hp.size());
            w += hp.size();
            continue;
        }
        for (uint32_t j = 0; j < gm.n_lines; j++) {
            const dp_paf_rec& r = recs[gm.slot + j];
            if (j + 4 < gm.n_lines) __builtin_prefetch(&rs.names[recs[gm.slot + j + 4].t_read], 0, 1);  // names are hit at random
            const std::string& qName = rs.names[r.q_read];
            const std::string& tName = rs.names[r.t_read];
            memcpy(w, qName.data(), qName.size());
            w += qName.size();
            *w++ = '\t';
            w = putInt(w, r.q_len);
            *w++ = '\t';
            w = putInt(w, r.q_start);
            *w++ = '\t';
            w = putInt(w, r.q_end);
            *w++ = '\t';
            *w++ = r.minus ? '-' : '+';
            *w++ = '\t';
            memcpy(w, tName.data(), tName.size());
            w += tName.size();
            *w++ = '\t';
            w = putInt(w, r.t_len);
            *w++ = '\t';
            w = putInt(w, r.t_start);
            *w++ = '\t';
            w = putInt(w, r.t_end);
            *w++ = '\t';
            w = putInt(w, r.ident);
            memcpy(w, "\t0\t255\n", 7);
            w += 7;
        }
    }
    TextJobBuffers::takeText(text);
    text.assign(w0, (size_t)(w - w0));
}

TextPool::TextPool(int nThreads) {
    for (int i = 0; i < std::max(1, nThreads); i++) th_.emplace_back([this] { loop(); });
}
TextPool::~TextPool() {
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
}
void TextPool::submit(std::shared_ptr<TextJob> job) {
    {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push_back(std::move(job));
    }
    cv_.notify_one();
}
void TextPool::loop() {
    struct Reg {
        Reg() { sampleProfRegister("text"); }
        ~Reg() { sampleProfUnregister(); }
    } reg;
    for (;;) {
        std::shared_ptr<TextJob> job;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;  // (stop requested and nothing left)
            job = std::move(q_.front());
            q_.pop_front();
        }
        const double tf = now();
        job->format();
        g_prof.formatUs += (long long)((now() - tf) * 1e6);
        job->finish();
    }
}

// finalCheckWorker commands/overlap.go:197-233 (+ collation :158-173).  Queries are independent (the reference runs
// num_workers finalCheckWorkers); they are spread over host threads here, and the PAF text and SetIgnore effects are
// applied in query order afterwards, so the result is the canonical single-worker output.
static void finalCheckEmit(SeedContig* contig, const SeedIndex& index, const ReadSet& reads, i64 overlapSize, std::string& paf,
                           std::vector<int>& ignoreIds, FinalCheckStats& fs);

static void finalCheckOne(Arena& arena, const SeedIndex& index, const ReadSet& reads, std::vector<SeedMatch*>& results,
                          i64 overlapSize, std::string& paf, std::vector<int>& ignoreIds, FinalCheckStats& fs) {
    finalCheckEmit(buildConsensus(arena, index, results, &fs.badBack), index, reads, overlapSize, paf, ignoreIds, fs);
}

// the PAF lines and SetIgnore decisions of one contig (commands/overlap.go:199-231)
static void finalCheckEmit(SeedContig* contig, const SeedIndex& index, const ReadSet& reads, i64 overlapSize, std::string& paf,
                           std::vector<int>& ignoreIds, FinalCheckStats& fs) {
    const int k = index.k;
    if (!contig || contig->Parts.size() <= 1) return;
    FINE(7);
    if (contig->SeqLengths[0] <= overlapSize * 2) ignoreIds.push_back(contig->Parts[0]);
    const i64 queryStart = contig->Offsets[0], queryEnd = queryStart + contig->Lengths[0];
    // one line = 12 tab-separated fields (commands/overlap.go:223-228): written into the string's own storage in one go
    auto put = [](char* w, i64 v) -> char* {  // %d
        char num[24];
        char* e = num + sizeof num;
        char* p = e;
        uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        do {
            *--p = (char)('0' + u % 10);
            u /= 10;
        } while (u);
        if (v < 0) *--p = '-';
        memcpy(w, p, (size_t)(e - p));
        return w + (e - p);
    };
    const std::string& qName = reads.names[(size_t)contig->Parts[0]];
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
        const std::string& tName = reads.names[(size_t)part];
        const size_t old = paf.size(), need = qName.size() + tName.size() + 7 * 21 + 24;
        if (paf.capacity() < old + need) paf.reserve(std::max(old + need, paf.capacity() * 2));
        paf.resize(old + need);
        char* w = &paf[old];
        memcpy(w, qName.data(), qName.size());
        w += qName.size();
        *w++ = '\t';
        w = put(w, contig->SeqLengths[0]);
        *w++ = '\t';
        w = put(w, queryStart);
        *w++ = '\t';
        w = put(w, queryEnd);
        *w++ = '\t';
        *w++ = rcs[0];
        *w++ = '\t';
        memcpy(w, tName.data(), tName.size());
        w += tName.size();
        *w++ = '\t';
        w = put(w, contig->SeqLengths[id]);
        *w++ = '\t';
        w = put(w, start);
        *w++ = '\t';
        w = put(w, end);
        *w++ = '\t';
        w = put(w, ident);
        memcpy(w, "\t0\t255\n", 7);
        w += 7;
        paf.resize((size_t)(w - paf.data()));
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

int finalCheck(Arena& arena, const SeedIndex& index, ReadSet& reads, const std::vector<SeedMatch*>& matches, i64 numQuerySeqs,
               i64 overlapSize, std::string& paf, FinalCheckStats& fs, std::vector<int>* ignoreOut, dp_ctx* ctx,
               std::vector<ConsJob>* jobs, std::string* errOut, RoundStats* st) {
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
    // DP_DEVICE_CONSENSUS=1: run the seed-space alignment in the middle of BuildConsensus on the device
    // (dp_consensus_align).  Off by default: measured on config 2 it takes 1.7 core-ms per round off the host (of 13.6)
    // but adds a 0.22 ms kernel and a second fork/join to every round, a net loss (5.6 M -> 4.6 M overlaps/s) until the
    // trimming before it and trimToBestSeed after it move to the device too.  Read per call so tests can toggle it.
    const char* devCons = getenv("DP_DEVICE_CONSENSUS");
    if (ctx && jobs && devCons && devCons[0] == '1' && nw > 0) {
        // ---- phase 1 (pool): trim the matched targets and reduce them, per query window
        if (jobs->size() < nw) jobs->resize(nw);
        parallelFor(nw, [&](size_t w) {
            const double tw = g_prof.on ? threadCpuNow() : 0;
            consensusPrepare((*jobs)[w], index, queryResults[work[w]]);
            if (g_prof.on) cpuUs += (long long)((threadCpuNow() - tw) * 1e6);
        });
        // ---- the seed-space alignment of every window in one device call
        static thread_local std::vector<int32_t> segs;
        static thread_local std::vector<uint64_t> seqOff;
        static thread_local std::vector<uint32_t> groupOff;
        segs.clear();
        seqOff.assign(1, 0);
        groupOff.assign(1, 0);
        for (size_t w = 0; w < nw; w++) {
            ConsJob& job = (*jobs)[w];
            if (!job.aligned) continue;
            job.group = (uint32_t)groupOff.size() - 1;
            job.firstSeq = (uint32_t)seqOff.size() - 1;
            for (SeedSeq* r : job.red) {
                if (r) segs.insert(segs.end(), r->seg, r->seg + r->n);
                seqOff.push_back(segs.size());
            }
            groupOff.push_back((uint32_t)seqOff.size() - 1);
        }
        dp_consensus_batch batch;
        memset(&batch, 0, sizeof batch);
        const uint32_t nGroups = (uint32_t)groupOff.size() - 1;
        if (nGroups) {
            const int rc = dp_consensus_align(ctx, segs.data(), seqOff.data(), groupOff.data(), nGroups, index.k, &batch);
            if (rc != 0) {
                if (errOut) *errOut = dp_last_error(ctx);
                return rc;
            }
            if (st) st->k_cons_ms += batch.kernel_ms;
        }
        // ---- phase 2 (pool): trimToBestSeed, contig, PAF
        const uint64_t* seqOffPtr = seqOff.data();  // (the vector is thread_local: name it here, not inside the workers)
        const dp_consensus_batch* batchPtr = nGroups ? &batch : nullptr;
        parallelFor(nw, [&, seqOffPtr, batchPtr](size_t w) {
            const double tw = g_prof.on ? threadCpuNow() : 0;
            std::string pafLocal;
            std::vector<int> ignLocal;
            FinalCheckStats fsLocal;
            pafLocal.reserve(2048);
            SeedContig* contig = consensusFinish((*jobs)[w], index, batchPtr, seqOffPtr, &fsLocal.badBack);
            finalCheckEmit(contig, index, reads, overlapSize, pafLocal, ignLocal, fsLocal);
            outs[w] = std::move(pafLocal);
            ign[w] = std::move(ignLocal);
            tfs[w] = fsLocal;
            if (g_prof.on) cpuUs += (long long)((threadCpuNow() - tw) * 1e6);
        });
    } else {
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
    }
    g_prof.consensusCpuUs += cpuUs.load();
    const double tf2 = now();
    g_prof.add(12, tf2 - tf1);
    for (size_t w = 0; w < nw; w++) {
        paf += outs[w];
        for (int id : ign[w]) {
            if (ignoreOut) ignoreOut->push_back(id);
            else flagStore(&reads.ignore[(size_t)id], 1);
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
    return 0;
}

}  // namespace dph
