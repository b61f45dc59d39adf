// ORACLE — test infrastructure only (see oracle.hpp).  Flat C entry points so that tests/ (ctypes),
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive the restatement.
#include <cstring>
#include <map>
#include <stdexcept>

#include "oracle.hpp"

using namespace dpo;

namespace {
thread_local std::string g_err;
template <class F>
int guard(F f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
SeedSequence* mkSeq(Arena& a, const i64* seg, i64 n) {
    SeedSequence* s = a.make();
    s->store = std::make_shared<std::vector<i64>>(seg, seg + n);
    s->lo = 0;
    s->n = (size_t)n;
    return s;
}
IntSet seedSetLikeIndex(const SeedSequence* s) {  // SeedIndex.AddSequence seeds.go:272-285
    i64 mx = s->getMaxSeed();
    IntSet st(mx + 1);
    for (i64 i = 0; i < s->numSeeds(); i++) st.add((u64)s->getSeed(i));
    return st;
}
void flatten(const std::vector<SeedMatch>& ms, i64* outCounts, i64* outA, i64* outB, i64 cap, i64* nMatches) {
    i64 pos = 0;
    *nMatches = (i64)ms.size();
    for (size_t i = 0; i < ms.size(); i++) {
        outCounts[i] = (i64)ms[i].MatchA.size();
        for (size_t j = 0; j < ms[i].MatchA.size(); j++) {
            if (pos >= cap) throw std::runtime_error("oracle capi: output capacity");
            outA[pos] = ms[i].MatchA[j];
            outB[pos] = ms[i].MatchB[j];
            pos++;
        }
    }
}
}  // namespace

extern "C" {

const char* dpo_last_error() { return g_err.c_str(); }

// ---- sequences -------------------------------------------------------------------------------
void* dpo_seq_new(const char* s, int64_t n) { return new PackedSeq(newPackedSequence(0, std::string(s, (size_t)n), nullptr)); }
void dpo_seq_free(void* h) { delete (PackedSeq*)h; }
void* dpo_seq_sub(void* h, int64_t start, int64_t end) { return new PackedSeq(((PackedSeq*)h)->subSequence(start, end)); }
void* dpo_seq_rc(void* h) { return new PackedSeq(((PackedSeq*)h)->reverseComplement()); }
int64_t dpo_seq_str(void* h, char* out, int64_t cap) {
    std::string s = ((PackedSeq*)h)->str();
    if ((int64_t)s.size() > cap) return -1;
    memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
}
// out: nbytes, firstLen, finalLen, offset, inset, length
void dpo_seq_meta(void* h, int64_t* out) {
    PackedSeq* p = (PackedSeq*)h;
    out[0] = (int64_t)p->nbytes();
    out[1] = p->firstLen;
    out[2] = p->finalLen;
    out[3] = p->offset;
    out[4] = p->inset;
    out[5] = p->length;
}
void dpo_seq_bytes(void* h, uint8_t* out) { memcpy(out, ((PackedSeq*)h)->data(), ((PackedSeq*)h)->nbytes()); }
int64_t dpo_seq_kmer_at(void* h, int64_t i, int k) { return ((PackedSeq*)h)->kmerAt(i, k); }
int64_t dpo_seq_next_kmer(void* h, int64_t cur, int64_t mask, int64_t idx) { return ((PackedSeq*)h)->nextKmer(cur, mask, idx); }
int64_t dpo_seq_count_kmers(void* h, int64_t upTo, int k, const uint8_t* seeds) { return ((PackedSeq*)h)->countKmers(upTo, k, seeds); }
int64_t dpo_seq_count_kmers_between(void* h, int64_t from, int64_t to, int64_t upTo, int k, const uint8_t* seeds) {
    return ((PackedSeq*)h)->countKmersBetween(from, to, upTo, k, seeds);
}
int64_t dpo_seq_write_segments(void* h, int k, const uint8_t* seeds, int64_t* out) {
    PackedSeq* p = (PackedSeq*)h;
    return packedWriteSegmentsAsm(p->data(), (i64)p->nbytes(), 4 - p->firstLen, 4 - p->finalLen, k, seeds, out);
}
int64_t dpo_byte_count_kmers(const char* s, int64_t n, int64_t upTo, int k, const uint8_t* seeds) {
    return byteCountKmers(std::string(s, (size_t)n), upTo, k, seeds);
}
int64_t dpo_byte_write_segments(const char* s, int64_t n, int k, const uint8_t* seeds, int64_t* out) {
    return byteWriteSegments(std::string(s, (size_t)n), k, seeds, out);
}
void dpo_pack_bytes(const uint8_t* s, int64_t n, uint8_t* out) { packBytesAsm(s, (size_t)n, out); }
uint64_t dpo_rc_kmer(uint64_t kmer, int k) { return reverseComplementKmer(kmer, k); }

// ---- IntSet ----------------------------------------------------------------------------------
void* dpo_set_new() { return new IntSet(); }
void* dpo_set_new_cap(int64_t cap) { return new IntSet(cap); }
void dpo_set_free(void* h) { delete (IntSet*)h; }
void dpo_set_add(void* h, uint64_t x) { ((IntSet*)h)->add(x); }
void dpo_set_clear(void* h) { ((IntSet*)h)->clear(); }
int dpo_set_contains(void* h, uint64_t x) { return ((IntSet*)h)->contains(x) ? 1 : 0; }
uint64_t dpo_set_size(void* h) { return ((IntSet*)h)->size(); }
// out: start, end, len(vs)
void dpo_set_window(void* h, uint64_t* out) {
    IntSet* s = (IntSet*)h;
    out[0] = s->start;
    out[1] = s->end;
    out[2] = (uint64_t)s->vs.size();
}
void dpo_set_words(void* h, uint64_t* out) { memcpy(out, ((IntSet*)h)->vs.data(), ((IntSet*)h)->vs.size() * 8); }
uint64_t dpo_set_count_intersection(void* a, void* b) { return ((IntSet*)a)->countIntersection(*(IntSet*)b); }
int64_t dpo_set_count_intersection_to(void* a, void* b, int64_t maxCount) {
    int64_t r = -1;
    guard([&] { r = (int64_t)((IntSet*)a)->countIntersectionTo(*(IntSet*)b, maxCount); });
    return r;
}
int64_t dpo_shared_ids(void** sets, int64_t n, int64_t minCount, int fast, uint64_t* out, int64_t cap) {
    std::vector<const IntSet*> v;
    for (int64_t i = 0; i < n; i++) v.push_back((IntSet*)sets[i]);
    std::vector<u64> ids = getSharedIDs(v, minCount, fast != 0);
    if ((int64_t)ids.size() > cap) return -1;
    memcpy(out, ids.data(), ids.size() * 8);
    return (int64_t)ids.size();
}
void dpo_soft_union(int which, const uint64_t* vs, int64_t n, uint64_t* out4) {
    if (which == 4) softUnion4(vs, n, out4);
    else if (which == 8) softUnion8(vs, n, out4);
    else softUnion16(vs, n, out4);
}
void dpo_gap_range(int64_t gap, int k, int64_t* out2) { gapRange(gap, k, &out2[0], &out2[1]); }

// ---- chaining on raw segment arrays ----------------------------------------------------------
// Calls PairwiseAlignments exactly as matchWorker does (overlap/overlap.go:351-365): aSet = seeds of a
// (NewIntSet), bSet = SeedIndex.AddSequence's seed set of b.  Matches come back in the reference's order.
int dpo_pairwise(const int64_t* aSeg, int64_t aN, const int64_t* bSeg, int64_t bN, int64_t minMatches, int k,
                 int64_t maxLength, int64_t* outCounts, int64_t* outA, int64_t* outB, int64_t cap, int64_t* nMatches) {
    return guard([&] {
        Arena ar;
        SeedSequence* a = mkSeq(ar, aSeg, aN);
        SeedSequence* b = mkSeq(ar, bSeg, bN);
        IntSet aSet;
        for (i64 i = 0; i < a->numSeeds(); i++) aSet.add((u64)a->getSeed(i));
        IntSet bSet = seedSetLikeIndex(b);
        SeedAligner al(maxLength);
        std::vector<SeedMatch> ms = al.pairwiseAlignments(a, b, aSet, bSet, minMatches, k);
        flatten(ms, outCounts, outA, outB, cap, nMatches);
    });
}
// SeedSequence.Match as performMapping calls it (mapping/mapping.go:514-526): seq = target, query.
int dpo_match(const int64_t* seqSeg, int64_t sN, const int64_t* qSeg, int64_t qN, int64_t minMatch, int k,
              int64_t* outCounts, int64_t* outA, int64_t* outB, int64_t cap, int64_t* nMatches) {
    return guard([&] {
        Arena ar;
        SeedSequence* s = mkSeq(ar, seqSeg, sN);
        SeedSequence* q = mkSeq(ar, qSeg, qN);
        IntSet qSet = seedSetLikeIndex(q);
        IntSet sSet = seedSetLikeIndex(s);
        std::vector<SeedMatch> ms = ssMatch(ar, s, q, &qSet, &sSet, minMatch, k);
        flatten(ms, outCounts, outA, outB, cap, nMatches);
    });
}

// multiAligner.Consensus (seeds/alignment.go:23-268) on raw segment arrays: sequence i = segs[off[i] .. off[i + 1]).  cons_out receives the
// consensus' segments ([dist, seed, ..., 0]); kept[] the indices of the sequences whose match survived (>= 3 pairs) in the order
// the function returns them, their pairs one after another in outA / outB with outCounts[j] pairs each.
int dpo_hand_consensus(const int64_t* segs, const int64_t* off, int n_seqs, int k, int64_t* cons_out, int64_t cons_cap, int64_t* cons_n,
                       int* kept, int64_t* outCounts, int64_t* outA, int64_t* outB, int64_t cap, int64_t* nMatches) {
    return guard([&] {
        Arena ar;
        std::vector<SeedSequence*> seqs;
        for (int i = 0; i < n_seqs; i++) seqs.push_back(mkSeq(ar, segs + off[i], off[i + 1] - off[i]));
        std::vector<std::unique_ptr<SeedMatch>> ms;
        SeedSequence* cons = multiAlignerConsensus(ar, seqs, k, ms);
        if ((int64_t)cons->n > cons_cap) throw std::runtime_error("dpo_hand_consensus: consensus buffer too small");
        for (size_t i = 0; i < cons->n; i++) cons_out[i] = cons->seg()[i];
        *cons_n = (int64_t)cons->n;
        int64_t at = 0;
        *nMatches = (int64_t)ms.size();
        for (size_t j = 0; j < ms.size(); j++) {
            int which = -1;
            for (int i = 0; i < n_seqs; i++)
                if (ms[j]->SeqB == seqs[(size_t)i]) which = i;
            kept[j] = which;
            outCounts[j] = (int64_t)ms[j]->MatchA.size();
            for (size_t x = 0; x < ms[j]->MatchA.size(); x++) {
                if (at >= cap) throw std::runtime_error("dpo_hand_consensus: match buffer too small");
                outA[at] = ms[j]->MatchA[x];
                outB[at] = ms[j]->MatchB[x];
                at++;
            }
        }
    });
}

// SeedMatch.GetBasesCovered (seeds/sequence.go:830-858) on raw segment arrays and a list of matched seed indices; out2 = {countA, countB};
// returns 1 where the reference would panic (an index out of range)
int dpo_hand_bases_covered(const int64_t* aSeg, int64_t aN, const int64_t* bSeg, int64_t bN, const int64_t* matchA, const int64_t* matchB,
                           int64_t n, int k, int64_t* out2) {
    int panicked = 0;
    int rc = guard([&] {
        Arena ar;
        SeedMatch m;
        m.SeqA = mkSeq(ar, aSeg, aN);
        m.SeqB = mkSeq(ar, bSeg, bN);
        m.MatchA.assign(matchA, matchA + n);
        m.MatchB.assign(matchB, matchB + n);
        i64 a = 0, b = 0;
        if (!smGetBasesCovered(m, k, &a, &b)) panicked = 1;
        out2[0] = a;
        out2[1] = b;
    });
    return rc ? rc : panicked;
}

// ---- decision rules on bare numbers (tests/test_hand_known_answers.py: answers worked by hand from the Go text) ----------------
// chunkWorker (overlap/overlap.go:253-318) on one seed sequence; out = 5 values per piece handed to AddSequence: first seed, seeds,
// Len(), GetOffset(), GetInset()
int dpo_hand_chunks(const int64_t* seg, int64_t n, int64_t length, int64_t offset, int64_t inset, int64_t chunkSize, int64_t overlap,
                    int64_t minSeeds, int k, int64_t* out, int64_t cap, int64_t* nOut) {
    return guard([&] {
        Arena ar;
        SeedSequence* s = mkSeq(ar, seg, n);
        s->length = length;
        s->offset = offset;
        s->inset = inset;
        int64_t made = 0;
        chunkPieces(ar, s, chunkSize, overlap, minSeeds, k, [&](SeedSequence* piece) {
            if (made < cap) {
                out[5 * made] = ((int64_t)piece->lo - (int64_t)s->lo) / 2;
                out[5 * made + 1] = piece->numSeeds();
                out[5 * made + 2] = piece->length;
                out[5 * made + 3] = piece->offset;
                out[5 * made + 4] = piece->inset;
            }
            made++;
        });
        *nOut = made;
    });
}
// step 1 of trimToBestSeed (overlap/combine.go:24-58): match i's MatchA = matchA[off[i] .. off[i + 1]); out2 = {bestIndex, backIndex}
int dpo_hand_trim_indices(int64_t upto, const int64_t* matchA, const int64_t* off, int64_t nMatches, int64_t minMatch, int64_t length,
                          int64_t* out2) {
    return guard([&] {
        std::vector<SeedMatch> store((size_t)nMatches);
        std::vector<SeedMatch*> ms;
        for (int64_t i = 0; i < nMatches; i++) {
            store[(size_t)i].MatchA.assign(matchA + off[i], matchA + off[i + 1]);
            ms.push_back(&store[(size_t)i]);
        }
        trimBestIndices(upto, ms, minMatch, length, &out2[0], &out2[1]);
    });
}
// isConsistent (mapping/mapping.go:131-160): left5 = {RC, Query.Len(), QueryInset, Start, End}, right4 = {RC, QueryOffset, Start, End}
int dpo_hand_is_consistent(const int64_t* left5, const int64_t* right4, int circular, int64_t refLen) {
    Mapping L, R;
    L.RC = left5[0] != 0;
    L.QueryInset = left5[2];
    L.Start = left5[3];
    L.End = left5[4];
    R.RC = right4[0] != 0;
    R.QueryOffset = right4[1];
    R.Start = right4[2];
    R.End = right4[3];
    return mappingsConsistent(&L, left5[1], &R, circular != 0, refLen) ? 1 : 0;
}
// removeDominated (mapping.go:387-428): maps3 = n x {QueryOffset, QueryInset, ids}; kept[] = the survivors' indices in returned order
int dpo_hand_remove_dominated(const int64_t* maps3, int n, int64_t queryLen, int* kept) {
    std::vector<Mapping> store((size_t)n);
    std::vector<Mapping*> open;
    for (int i = 0; i < n; i++) {
        store[(size_t)i].QueryOffset = maps3[3 * i];
        store[(size_t)i].QueryInset = maps3[3 * i + 1];
        store[(size_t)i].ids = maps3[3 * i + 2];
        open.push_back(&store[(size_t)i]);
    }
    std::vector<Mapping*> out = removeDominated(open, nullptr, queryLen);
    for (size_t i = 0; i < out.size(); i++) kept[i] = (int)(out[i] - store.data());
    return (int)out.size();
}

// ---- k-mer value table -----------------------------------------------------------------------
struct ReadSetH {
    FastaSet set;
};
void* dpo_reads_from_fasta(const char* path, int64_t minLen, int himem) {
    ReadSetH* h = new ReadSetH{FastaSet::fromFile(path, minLen, himem != 0)};
    if (!h->set.error.empty()) {  // the reference's log.Fatal
        g_err = h->set.error;
        delete h;
        return nullptr;
    }
    return h;
}
// concatenated ASCII reads + offsets (n+1)
void* dpo_reads_from_arrays(const char* bases, const int64_t* off, int64_t n, int64_t minLen, int himem) {
    std::vector<std::string> names, seqs;
    char nm[32];
    for (int64_t i = 0; i < n; i++) {
        snprintf(nm, sizeof nm, "r%07lld", (long long)i);
        names.push_back(nm);
        seqs.emplace_back(bases + off[i], (size_t)(off[i + 1] - off[i]));
    }
    return new ReadSetH{FastaSet::fromReads(names, seqs, minLen, himem != 0)};
}
// the same with a quality string per read (raw FASTQ bytes, same offsets as the bases): the set behaves as if read from a FASTQ
void* dpo_reads_from_arrays_q(const char* bases, const char* quals, const int64_t* off, int64_t n, int64_t minLen, int himem) {
    std::vector<std::string> names, seqs, qs;
    char nm[32];
    for (int64_t i = 0; i < n; i++) {
        snprintf(nm, sizeof nm, "r%07lld", (long long)i);
        names.push_back(nm);
        seqs.emplace_back(bases + off[i], (size_t)(off[i + 1] - off[i]));
        qs.emplace_back(quals + off[i], (size_t)(off[i + 1] - off[i]));
    }
    ReadSetH* h = new ReadSetH{FastaSet::fromReads(names, seqs, minLen, himem != 0, &qs)};
    h->set.isFastq = true;
    return h;
}
void dpo_reads_free(void* h) { delete (ReadSetH*)h; }
// "name\tACGT...\n" per read (the 2-bit content spelled back): lets tests compare FASTA readers
const char* dpo_reads_dump(void* h, int64_t* n) {
    static thread_local std::string out;
    out.clear();
    FastaSet& s = ((ReadSetH*)h)->set;
    for (size_t i = 0; i < s.size(); i++) {
        out += s.names[i];
        out += '\t';
        out += s.cached[i].str();
        if (s.cached[i].qual) {  // FASTQ: the stored quality bytes (phred - 33), two hex digits each
            out += '\t';
            const PackedSeq& c = s.cached[i];
            for (i64 j = 0; j < c.length; j++) {
                const uint8_t q = (*c.qual)[c.qlo + (size_t)j];
                out += "0123456789abcdef"[q >> 4];
                out += "0123456789abcdef"[q & 15];
            }
        }
        out += '\n';
    }
    *n = (int64_t)out.size();
    return out.data();
}
int64_t dpo_reads_count(void* h) { return (int64_t)((ReadSetH*)h)->set.size(); }
void dpo_reads_reset_ignore(void* h) {
    auto& ig = ((ReadSetH*)h)->set.ignore;
    std::fill(ig.begin(), ig.end(), 0);
}
void dpo_reads_get_ignore(void* h, uint8_t* out) {
    auto& ig = ((ReadSetH*)h)->set.ignore;
    memcpy(out, ig.data(), ig.size());
}
// values must hold 4^k doubles
int dpo_kmer_values(void* h, int k, double* values) {
    return guard([&] {
        std::vector<u64> counts = kmerOccurrences(((ReadSetH*)h)->set.cached, k);
        std::vector<double> v = kmerValues(counts, k);
        memcpy(values, v.data(), v.size() * sizeof(double));
    });
}
int dpo_kmer_counts(void* h, int k, uint64_t* counts) {
    return guard([&] {
        std::vector<u64> c = kmerOccurrences(((ReadSetH*)h)->set.cached, k);
        memcpy(counts, c.data(), c.size() * 8);
    });
}

// AddSeeds (seeds/seeds.go:62-156) of each sequence into an EMPTY index: out receives the resulting seedMap (k-mers in
// seed-id order) per sequence, off[i]..off[i+1].  Test hook for the product's device-side selection.
int dpo_add_seeds_each(void** seqs, int64_t n, int k, int64_t numSeeds, const double* values, int64_t* out, int64_t cap,
                       int64_t* off) {
    return guard([&] {
        SeedIndex index(k);
        int64_t pos = 0;
        off[0] = 0;
        for (int64_t i = 0; i < n; i++) {
            index.addSeeds(*(PackedSeq*)seqs[i], numSeeds, values);
            for (i64 km : index.seedMap) {
                if (pos >= cap) throw std::runtime_error("dpo_add_seeds_each: output too small");
                out[pos++] = km;
            }
            off[i + 1] = pos;
            for (i64 km : index.seedMap) {  // back to the empty index
                index.kmers[(size_t)km] = 0;
                index.kmerMap[(size_t)km] = 0;
            }
            index.seedMap.clear();
            index.sequenceSets.clear();  // grown in lock step with seedMap (addSeedKmer)
            index.size = 0;
        }
    });
}

// ---- overlap command -------------------------------------------------------------------------
struct OverlapH {
    OverlapResult res;
    std::map<std::pair<int64_t, int>, std::vector<int64_t>> cache;
};
// params: overlapSize,k,numSeeds,seedBatchSize,chunkSize,queryBatchSize,himem ; minHits separately
void* dpo_overlap_run(void* reads, const int64_t* params, double minHits, const double* valuesOrNull, int64_t maxRounds,
                      int keepTraces) {
    OverlapH* h = new OverlapH();
    int rc = guard([&] {
        OverlapParams p;
        p.overlapSize = params[0];
        p.k = (int)params[1];
        p.numSeeds = params[2];
        p.seedBatchSize = params[3];
        p.chunkSize = params[4];
        p.queryBatchSize = params[5];
        p.himem = (params[6] & 1) != 0;
        p.queryType = (int)(params[6] >> 8) ? (int)(params[6] >> 8) : 1;  // bits 8.. of the himem word: overlap.Query* flags
        p.minHits = minHits;
        ((ReadSetH*)reads)->set.himem = p.himem;
        h->res = runOverlap(((ReadSetH*)reads)->set, p, valuesOrNull, maxRounds, keepTraces != 0);
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}
void dpo_overlap_free(void* h) { delete (OverlapH*)h; }
int64_t dpo_overlap_rounds(void* h) { return ((OverlapH*)h)->res.rounds; }
const char* dpo_overlap_paf(void* h, int64_t* n) {
    *n = (int64_t)((OverlapH*)h)->res.paf.size();
    return ((OverlapH*)h)->res.paf.data();
}
const char* dpo_overlap_err(void* h, int64_t* n) {
    *n = (int64_t)((OverlapH*)h)->res.err.size();
    return ((OverlapH*)h)->res.err.data();
}
// Trace fields.  Nested fields come as (flat, offsets) pairs: field f = flat data, f+100 = offsets.
//  0 seedKmers | 1 queryIDs | 2 querySeqIDs | 3 indexedIds | 4 indexedLength | 5 indexedOffset | 6 indexedInset
//  7 matchQueryIndex | 8 matchTarget | 9 newlyIgnored | 10 scalars {firstSequence,numQuerySeqs,hits,qHits}
//  20 querySegments | 21 indexedSegments | 22 candidates | 23 matchA | 24 matchB
static void flat(const std::vector<std::vector<i64>>& v, std::vector<i64>& data, std::vector<i64>& off) {
    off.push_back(0);
    for (auto& x : v) {
        data.insert(data.end(), x.begin(), x.end());
        off.push_back((i64)data.size());
    }
}
const int64_t* dpo_overlap_trace(void* hh, int64_t round, int field, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    if (round < 0 || round >= (int64_t)h->res.traces.size()) {
        *n = -1;
        return nullptr;
    }
    auto key = std::make_pair(round, field);
    auto it = h->cache.find(key);
    if (it == h->cache.end()) {
        const RoundTrace& t = h->res.traces[(size_t)round];
        std::vector<i64> d, o;
        int base = field >= 100 ? field - 100 : field;
        switch (base) {
            case 0: d = t.seedKmers; break;
            case 1: d = t.queryIDs; break;
            case 2: d = t.querySeqIDs; break;
            case 3: d = t.indexedIds; break;
            case 4: d = t.indexedLength; break;
            case 5: d = t.indexedOffset; break;
            case 6: d = t.indexedInset; break;
            case 7: d = t.matchQueryIndex; break;
            case 8: d = t.matchTarget; break;
            case 9: d = t.newlyIgnored; break;
            case 11: d = t.queryLength; break;
            case 12: d = t.queryOffset; break;
            case 13: d = t.queryInset; break;
            case 10: d = {t.firstSequence, t.numQuerySeqs, t.hits, t.qHits}; break;
            case 20: flat(t.querySegments, d, o); break;
            case 21: flat(t.indexedSegments, d, o); break;
            case 22: {
                std::vector<std::vector<i64>> c;
                for (auto& x : t.candidates) c.emplace_back(x.begin(), x.end());
                flat(c, d, o);
                break;
            }
            case 23: flat(t.matchA, d, o); break;
            case 24: flat(t.matchB, d, o); break;
            default: *n = -1; return nullptr;
        }
        h->cache[std::make_pair(round, base)] = d;
        h->cache[std::make_pair(round, base + 100)] = o;
        it = h->cache.find(key);
    }
    *n = (int64_t)it->second.size();
    return it->second.data();
}
const char* dpo_overlap_trace_paf(void* hh, int64_t round, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    const std::string& s = h->res.traces[(size_t)round].paf;
    *n = (int64_t)s.size();
    return s.data();
}

// ---- map command -----------------------------------------------------------------------------
struct MapH {
    MapResult res;
};
// params: circular,k,querySize,minLength(unused here: applied when the read set was loaded),chunkSize,seedRate
void* dpo_map_run(void* refSet, void* reads, const int64_t* params) {
    MapH* h = new MapH();
    int rc = guard([&] {
        MapParams p;
        p.circular = params[0] != 0;
        p.k = (int)params[1];
        p.querySize = params[2];
        p.minLength = params[3];
        p.chunkSize = params[4];
        p.seedRate = params[5];
        h->res = runMap(((ReadSetH*)refSet)->set, ((ReadSetH*)reads)->set, p);
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}
void dpo_map_free(void* h) { delete (MapH*)h; }
const char* dpo_map_paf(void* h, int64_t* n) {
    *n = (int64_t)((MapH*)h)->res.paf.size();
    return ((MapH*)h)->res.paf.data();
}
const char* dpo_map_err(void* h, int64_t* n) {
    *n = (int64_t)((MapH*)h)->res.err.size();
    return ((MapH*)h)->res.err.data();
}

}  // extern "C"
