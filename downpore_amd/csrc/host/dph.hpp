// Host side of the product: a C++ mirror of the reference's Go layers that sit ABOVE the C ABI
// (include/downpore_hip.h).  Go is not available in this image, so what would be a cgo-backed Go implementation of
// overlap.Overlapper / mapping.Mapper is written here in C++ with the same names, argument meaning and order of
// operations (see INTEGRATION.md for the Go-side stub).  Everything data-parallel goes through the C ABI to the GPU;
// what stays on the host is the sequential / control logic SURVEY §8(a) marks "host": FASTA rules (A23), value table
// arithmetic (A22, float64), seed selection (A9), query preparation (A15), chunking (A12), consensus (A16), PAF (A17).
//
// This code does not include, link or call anything under oracle/.
#pragma once
#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <cstring>
#include <new>
#include <algorithm>
#include <deque>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "downpore_hip.h"

namespace dph {

// The SetIgnore flags are bytes that planner lanes and executor slots read without a lock while a commit sets them (what a stale
// read may cause is caught by the epoch / flag-round validation): relaxed atomic accesses, so that the concurrency is defined
// behaviour (and ThreadSanitizer-clean: `make host-tsan`, tests/test_planner_epoch.py).
inline uint8_t flagLoad(const uint8_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
inline void flagStore(uint8_t* p, uint8_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }

typedef int64_t i64;

// ---- read set (sequence/seqio.go fastaSequenceSet, FASTA subset) -------------------------------------------------
struct ReadSet {
    std::vector<std::string> names;
    std::vector<i64> off;          // offsets into bases (n+1)
    std::string bases;             // concatenated ASCII of the kept reads
    // FASTQ (sequence/seqio.go:106-276): phred byte - 33 per base at the same offsets as `bases` (empty for a FASTA);
    // hasQual[r] = the record's quality line had exactly the read's length (:229-238), else the read carries none
    std::string qual;
    std::vector<uint8_t> hasQual;
    bool isFastq = false;
    const uint8_t* quality(size_t r) const { return (!qual.empty() && hasQual[r]) ? (const uint8_t*)qual.data() + off[r] : nullptr; }
    std::vector<uint8_t> ignore;   // SetIgnore flags (seqio.go:375); planner lanes read them while a commit sets them: flagLoad / flagStore
    bool himem = true;             // cached views (seqio.go:115) vs top-level re-reads (:158)
    size_t maxNameLen = 0;         // longest name (PAF line buffers are sized from it)
    size_t size() const { return names.size(); }
    i64 length(size_t r) const { return off[r + 1] - off[r]; }
    const char* seq(size_t r) const { return bases.data() + off[r]; }
    void addLine(const std::string& lastName, const char* line, size_t len, i64 minLen, const char* qualLine = nullptr, size_t qualLen = 0);
    static bool fromFile(const std::string& path, i64 minLen, bool himem, ReadSet& out, std::string& err);
    static ReadSet fromArrays(const char* bases, const i64* off, size_t n, i64 minLen, bool himem, const char* quals = nullptr);
    // k-mers the reference's scan examines for the view a later pass receives
    i64 scanKmers(size_t r, int k) const;
    // GetInset() of the served view: SubSequence's inset is one too large (sequence.go:365)
    i64 servedInset() const { return himem ? 1 : 0; }
};

static inline uint32_t baseCode(unsigned char b) { return ((b >> 1) ^ ((b & 4) >> 2)) & 3; }
uint32_t reverseComplementKmer(uint32_t kmer, int k);  // seeds/sequence.go:125-132
void packBases(const char* src, size_t n, uint8_t* dst, bool scalarOnly = false);  // packBytes of a whole read (host_seq.cpp); dst: ceil(n / 4) bytes

// ---- value table (commands/overlap.go:39-94, util/sequtil/kmers.go:87-112) ---------------------------------------
std::vector<double> kmerValuesFromCounts(std::vector<uint64_t>& counts, int k);

// ---- seed space ---------------------------------------------------------------------------------------------------
// SeedSequence (seeds/sequence.go:10-20).  Segments live in a shared int32 pool owned by the round.
struct SeedSeq {
    const int32_t* seg = nullptr;  // [gap, seed, gap, ..., gap]
    int n = 0;                     // number of ints
    int id = 0;
    i64 length = 0, offset = 0, inset = 0;
    bool rc = false;
    SeedSeq* parent = nullptr;
    SeedSeq* reverseComplement = nullptr;
    int numSeeds() const { return n / 2; }
    i64 seedOffset(int index, int k) const;         // GetSeedOffset        :1239
    i64 seedOffsetFromEnd(int index, int k) const;  // GetSeedOffsetFromEnd :1269
    i64 nextSeedOffset(int index, int k) const { return (i64)seg[index * 2 + 2] + k; }
    int maxSeed() const;
};

struct SeedMatch {  // seeds/sequence.go:24-32
    std::vector<int32_t> MatchA, MatchB;
    SeedSeq* SeqA = nullptr;
    SeedSeq* SeqB = nullptr;
    int QueryID = 0;
    bool ReverseComplementQuery = false;
    // from the device (dp_match_batch.target_anchor), in the FORWARD target's coordinates: GetSeedOffset(MatchB[0]) and
    // GetSeedOffsetFromEnd(MatchB.back()) as the chain came back; -1 = unknown (Trimmed() then sums the gaps itself)
    i64 anchorFirstB = -1, anchorLastFromEndB = -1;
};

// per-round arena (Go GC stand-in): bump allocation out of slabs that are kept across clear()
struct Arena {
    struct Slab {
        std::unique_ptr<char[]> p;
        size_t cap = 0, used = 0;
    };
    std::vector<Slab> slabs;
    size_t cur = 0;
    size_t slabBytes = (size_t)1 << 20;  // minimum size of a new slab
    void* raw(size_t bytes, size_t align) {
        for (;;) {
            if (cur < slabs.size()) {
                Slab& s = slabs[cur];
                const size_t at = (s.used + align - 1) & ~(align - 1);
                if (at + bytes <= s.cap) {
                    s.used = at + bytes;
                    return s.p.get() + at;
                }
                cur++;
                continue;
            }
            Slab s;
            s.cap = std::max<size_t>(bytes + align, slabBytes);
            s.p.reset(new char[s.cap]);
            slabs.push_back(std::move(s));
        }
    }
    SeedSeq* make() { return new (raw(sizeof(SeedSeq), alignof(SeedSeq))) SeedSeq(); }
    int32_t* alloc(size_t n) {
        int32_t* p = (int32_t*)raw(n * sizeof(int32_t) + 4, alignof(int32_t));
        memset(p, 0, n * sizeof(int32_t));
        return p;
    }
    void clear() {
        for (Slab& s : slabs) s.used = 0;
        cur = 0;
    }
};

// The k-mer value table (commands/overlap.go:55-93) as the selection loops read it: 4^k doubles, or - a value being a function
// of the k-mer's own count and the total count alone - one 2-byte code per k-mer (its count; 0 = value 0) with the 65536
// values the codes stand for (dp_values_download_codes).  Both forms give the same double for every k-mer.
struct ValueView {
    const double* full = nullptr;
    const uint16_t* codes = nullptr;
    const uint8_t* codes8 = nullptr;  // one byte per k-mer where every valued k-mer counts below 255 (dp_values_download_codes8)
    const double* lut = nullptr;
    ValueView() {}
    ValueView(const double* f) : full(f) {}
    ValueView(const uint16_t* c, const double* l) : codes(c), lut(l) {}
    ValueView(const uint8_t* c, const double* l) : codes8(c), lut(l) {}
    double at(uint32_t kmer) const { return full ? full[kmer] : codes8 ? lut[codes8[kmer]] : lut[codes[kmer]]; }
    void prefetch(uint32_t kmer) const {
        if (full) __builtin_prefetch(&full[kmer], 0, 0);
        else if (codes8) __builtin_prefetch(&codes8[kmer], 0, 0);
        else __builtin_prefetch(&codes[kmer], 0, 0);
    }
};

// Host mirror of seeds.SeedIndex (seeds/seeds.go:11-21): the seed <-> k-mer maps and the list of indexed sequences.
// The posting sets / seed sets themselves live on the GPU (dp_index_build).
struct SeedIndex {
    int k;
    // k-mer -> seed id: a small open-addressing table (<= ~20k seeds per round) that stays cache resident, instead of the
    // reference's 4^k-entry kmers/kmerMap arrays (seeds/seeds.go:13,17)
    std::vector<uint32_t> hkeys;                    // 0xffffffff = empty
    std::vector<int32_t> hvals;
    uint32_t hmask = 0;
    std::vector<uint32_t> seedMap;                  // seed id -> k-mer
    // adopt(): an executor slot takes a finished plan's seed list and reverse-complement table as they are - it never asks
    // "is this k-mer a seed" - and the hash is only rebuilt should somebody ask after all (hashValid)
    bool hashValid = true;
    void adopt(const std::vector<uint32_t>& seeds, const std::vector<int32_t>& rcTable);
    void rebuildHash();
    std::vector<SeedSeq*> sequences;                // indexed sequences (chunks), index == GPU sequence index
    std::vector<dp_seq_ref> refs;                   // their views into the device-resident scan output
    Arena arena;
    explicit SeedIndex(int k_, int preBits = 18);
    void reset();
    int size() const { return (int)seedMap.size(); }
    static uint32_t hash(uint32_t x) { return (x * 2654435761u) >> 7; }
    int32_t find(uint32_t kmer) const {             // seed id or -1
        if (__builtin_expect(!hashValid, 0)) const_cast<SeedIndex*>(this)->rebuildHash();
        uint32_t h = hash(kmer) & hmask;
        for (;;) {
            const uint32_t kk = hkeys[h];
            if (kk == kmer) return hvals[h];
            if (kk == 0xffffffffu) return -1;
            h = (h + 1) & hmask;
        }
    }
    // bit pre-filter in front of the hash (a miss is the common case during seed selection; with ~20 k seeds and 2^21 bits only
    // one probe in a hundred goes on to the hash)
    // (2^21 bits for the planner's index, whose touch test probes ~170 k k-mers per plan; 2^18 - 32 KiB, cleared every round - for
    // the executor slots' indexes, which only ever insert)
    uint32_t preShift = 32 - 18;
    std::vector<uint32_t> pre;
    uint32_t preHash(uint32_t x) const { return (x * 2246822519u) >> preShift; }
    bool isSeed(uint32_t kmer) const {
        if (__builtin_expect(!hashValid, 0)) const_cast<SeedIndex*>(this)->rebuildHash();
        const uint32_t b = preHash(kmer);
        if (!((pre[b >> 5] >> (b & 31)) & 1)) return false;
        return find(kmer) >= 0;
    }
    void grow();
    void addSeedKmer(uint32_t kmer);                                     // seeds.go:132-141
    int32_t addSeedKmerId(uint32_t kmer);                                // ... returning the seed's id (new or existing)
    std::vector<int32_t> rcPair;                                         // commitSeeds: id of each seed's reverse complement
    void addSeeds(const char* s, i64 len, int minSeeds, ValueView ranks);  // AddSeeds :62-156
    // q (may be null): the window's quality bytes (seq.Quality(), seeds.go:73): value *= q[nextIndex - k/2] (:99-101)
    void selectSeeds(const char* s, i64 len, int minSeeds, ValueView ranks, uint32_t* topN, bool checkIndex,
                     const uint8_t* q = nullptr) const;
    bool touchesSeed(const char* s, i64 len) const;
    // the same test on the window's evaluated k-mers (0xffffffff = unused slot): sixteen pre-filter probes per step where
    // the CPU has AVX-512 (eight with AVX2), the hash only for the probes that pass
    bool touchesSeed(const uint32_t* kmers, uint32_t n) const;
    bool touchesSeedWith(int isa, const uint32_t* kmers, uint32_t n) const;  // tests: 0 scalar, 1 AVX2, 2 AVX-512; false if the CPU lacks it
    void commitSeeds(const uint32_t* topN, int n);
    int32_t seedOfRcKmer(int32_t seed) const;                            // kmerMap[rc(seedMap[seed])]
    // seed id -> seed id of its reverse complement for the complete seed set (call once all seeds of the round are in)
    void buildRcTable();
    std::vector<int32_t> rcOf;
    std::vector<Arena> chunkArenas;  // one per block of survivors when chunking runs on the worker pool (kept across rounds)
};

SeedSeq* seqReverseComplement(Arena& a, SeedSeq* s, const SeedIndex& ix);  // seeds/sequence.go:134-159
SeedSeq* seqSubSequence(Arena& a, SeedSeq* s, int start, int end, i64 length, i64 offset, i64 inset);  // :46
SeedSeq* seqTrimmed(Arena& a, SeedSeq* s, i64 startOffset, int startSeed, i64 endOffset, int endSeed, int k, i64 anchorStart = -1,
                    i64 anchorEnd = -1);  // :54
SeedSeq* seqReduced(Arena& a, SeedSeq* s, const std::vector<uint64_t>& whitelist, int k, int minSeeds,
                    std::vector<int>* index);                                                              // :85
void matchReverseComplement(Arena& a, SeedMatch& m, const SeedIndex& ix);   // :800
void matchBasesCovered(const SeedMatch& m, int k, i64* a, i64* b, bool* wouldPanic);  // :830
void matchBaseIndex(const SeedMatch& m, int aIndex, int k, i64* index, i64* bases, i64* distance);  // :1190
void gapRange(i64 gap, int k, i64* mn, i64* mx);                          // seeds/alignment.go:411

// seeds/alignment.go:23-268
SeedSeq* multiAlignerConsensus(Arena& a, std::vector<SeedSeq*>& seqs, int k, std::vector<SeedMatch*>& out);  // (the matches live in the arena)

// overlap/combine.go
struct SeedContig {
    std::vector<int> Parts;
    std::vector<i64> Offsets, Lengths, SeqLengths;
    std::vector<uint8_t> ReverseComplement, Approximate;
    std::vector<SeedMatch*> Matches;
};
// Returns a contig in per-thread storage (valid until the calling thread's next buildConsensus) or nullptr.
SeedContig* buildConsensus(Arena& a, const SeedIndex& sg, std::vector<SeedMatch*>& overlaps, i64* badBack);

// BuildConsensus in two halves so that the seed-space alignment in the middle can run on the device for all query
// windows of a round at once (dp_consensus_align).  consensusPrepare: un-RC, trim the matched targets, Reduced() to the
// seeds shared by >= 2 of them.  consensusFinish: consensus + per-sequence matches (from the device batch, or computed
// here when `batch` is null / the group was flagged) -> trimToBestSeed -> contig.
struct ConsJob {
    ConsJob() { arena.slabBytes = (size_t)1 << 16; }
    Arena arena;                          // everything the job allocates; cleared by consensusPrepare
    std::vector<SeedSeq*> seqs, red;      // trimmed targets and their reduced forms (null = no shared seed)
    std::vector<std::vector<int>> seedMap;  // reduced index -> index in seqs[i]
    bool aligned = false;                 // seqs.size() > 1: takes part in the alignment
    uint32_t group = 0, firstSeq = 0;     // position in the flattened device batch
};
void consensusPrepare(ConsJob& job, const SeedIndex& sg, std::vector<SeedMatch*>& overlaps);
SeedContig* consensusFinish(ConsJob& job, const SeedIndex& sg, const dp_consensus_batch* batch, const uint64_t* seqOff, i64* badBack);

// ---- overlap.Overlapper (overlap/overlap.go:24-29) ----------------------------------------------------------------
struct SeedQuery {  // overlap/overlap.go:10-16
    int ID = 0, SequenceID = 0;
    SeedSeq* Query = nullptr;
    bool AtStart = true, ReverseComplement = false;
};

struct OverlapParams {  // flag table commands/overlap.go:24-25
    i64 overlapSize = 1000;
    int k = 10;
    int numSeeds = 15;
    i64 seedBatchSize = 10000, chunkSize = 10000, queryBatchSize = 20000;
    double minHits = 0.25;
    int numWorkers = 4;  // accepted for compatibility; the GPU path is batch-parallel
    bool himem = true;
    int queryType = 1;   // overlap.QueryEdges | QueryCentre | QueryAll (+ WeightEdges), overlap.go:18-21
};

struct RoundStats {
    double t_prepare = 0, t_scan = 0, t_index = 0, t_query = 0, t_consensus = 0;  // host wall seconds
    double k_scan_ms = 0, k_query_ms = 0, k_chain_ms = 0;                         // device kernel ms
    double k_count_ms = 0, k_write_ms = 0;                                        // scan passes
    double k_index_ms = 0;                                                        // index build (chunks on the device)
    double k_cons_ms = 0;                                                         // consensus alignment kernel
    uint64_t count_bytes = 0;                                                     // algorithmic bytes of the count pass
    uint64_t scan_bases = 0, scan_items = 0, scan_bytes = 0, query_bytes = 0;
    uint64_t n_queries = 0, n_indexed = 0, n_hits = 0, n_matches = 0, n_paf = 0, n_seeds = 0;
    uint64_t chain_bytes = 0;               // algorithmic bytes of the prefilter + chaining kernel
    uint64_t cons_bytes = 0;                // algorithmic bytes of the consensus kernel (sum of its windows' group records)
    uint64_t idx_rounds = 0, idx_hits = 0;  // rounds served by the resident k-mer position index, and their seed occurrences
    uint64_t timed_rounds = 0;              // rounds whose kernels were bracketed by timing events (dp_set_kernel_timing): the k_*_ms are theirs
    uint64_t gang_members = 0;              // (always 1 per round since round 6: the field keeps the statistics block's layout) ; 1 without a gang
    void add(const RoundStats& o) {
        t_prepare += o.t_prepare, t_scan += o.t_scan, t_index += o.t_index, t_query += o.t_query, t_consensus += o.t_consensus;
        k_scan_ms += o.k_scan_ms, k_query_ms += o.k_query_ms, k_chain_ms += o.k_chain_ms, k_count_ms += o.k_count_ms;
        k_index_ms += o.k_index_ms;
        k_write_ms += o.k_write_ms, k_cons_ms += o.k_cons_ms, count_bytes += o.count_bytes, scan_bases += o.scan_bases;
        scan_items += o.scan_items, scan_bytes += o.scan_bytes, query_bytes += o.query_bytes, n_queries += o.n_queries;
        n_indexed += o.n_indexed, n_hits += o.n_hits, n_matches += o.n_matches, n_paf += o.n_paf, n_seeds += o.n_seeds;
        chain_bytes += o.chain_bytes, idx_rounds += o.idx_rounds, idx_hits += o.idx_hits, timed_rounds += o.timed_rounds;
        gang_members += o.gang_members;
        cons_bytes += o.cons_bytes;
    }
};

// Survivor exchange hook for multi-GPU runs: after the local scan, the caller may replace the local
// survivor list by the rank-ordered concatenation over all ranks.
struct Survivors {
    std::vector<uint32_t> read;      // read id per survivor (ascending)
    std::vector<uint32_t> n_seeds;
    std::vector<uint64_t> seg_off;   // n+1
    std::vector<int32_t> segs;       // owned copy (after a multi-GPU exchange) ...
    const int32_t* segsView = nullptr;  // ... or a view of the producing context's pinned scan output (valid until that
    uint64_t segsViewLen = 0;           // context scans again): the dense-seed regime moves ~80 MB per round here
    bool deviceResident = false;     // the device scan buffer of the producing context holds exactly these ints at the same offsets
    bool segsOnHost = true;          // false: the scan left the survivors' segments on the device (dp_scan_fetch_mode): segData() is not valid
    const void* chunkCtx = nullptr;  // the context whose device-side survivor list describes exactly this set (its own scan, or the
                                     // set dp_allgather_survivors installed): dp_index_build_chunked can chunk it where it lies
    const int32_t* segData() const { return segsView ? segsView : segs.data(); }
    uint64_t segCount() const { return segsView ? segsViewLen : (uint64_t)segs.size(); }
};

struct TextJob;
class TextPool;
struct FinalCheckStats {
    i64 badBack = 0, emptyMatch = 0;
    uint64_t lines = 0, hits = 0, qHits = 0;
};

// The query windows of the `overlap` command (getEdges, overlap.go:55-89: the first and last overlap_size bases of every
// read, the whole read when it is shorter than two of them) with everything about them that does not depend on the round:
// the selection AddSeeds makes when no evaluated k-mer is a seed yet, and the evaluated k-mers themselves
// (dp_select_windows).  Produced on the device in read order by a thread that runs ahead of the planner; the plan chain
// then never waits for the GPU and its "did the speculation hold" test probes resident k-mers.
class WindowCache {
   public:
    struct Win {
        uint32_t read, start, len;
    };
    // ctx == nullptr: the producer selects on the host with hostValues (tests of the plan chain without a GPU)
    WindowCache(dp_ctx* ctx, const ReadSet& reads, i64 overlap, int k, int numSeeds, ValueView hostValues = ValueView((const double*)nullptr));
    ~WindowCache();
    std::vector<uint32_t> first;  // windows of read r: [first[r], first[r + 1])
    std::vector<Win> wins;
    uint32_t stride = 0;          // evaluated k-mers per window (0xffffffff = unused slot)
    int numSeeds = 0;
    // Blocks until window w has been produced.  The pointers stay valid until release() passes w's read.
    bool get(uint32_t w, const uint32_t** spec, const uint32_t** kmers, std::string* err);
    // the same, and how far the block of windows goes that lies behind these pointers (window w + i: spec + i * numSeeds, kmers + i *
    // stride, for w + i < *wEnd): a plan walks ~670 consecutive windows and asked - under the cache's lock - for every one of them
    bool getRun(uint32_t w, const uint32_t** spec, const uint32_t** kmers, uint32_t* wEnd, std::string* err);
    void release(size_t belowRead);  // reads below are committed: their windows will not be asked for again
    static size_t releaseSpares();   // dph_release_caches: the chunk buffers kept for a handle's next job (up to 16 x 11.5 MB)
    // Seeds window w adds to an index none of its k-mers touches (its cached selection and the reverse complements, without
    // repeats), or -1 while w has not been produced: what a planner lane needs to guess where the plan in front of it ends.
    int seedsOf(uint32_t w) const { return w < producedWins.load(std::memory_order_acquire) ? (int)seedCount[w] : -1; }

   private:
    std::vector<uint16_t> seedCount;
    std::atomic<uint32_t> producedWins{0};
    void producer();
    struct Impl;
    std::unique_ptr<Impl> d;
};

class Overlapper {
   public:
    Overlapper(dp_ctx* ctx, ReadSet& reads, SeedIndex& index, i64 chunkSize, int numWorkers, i64 overlap, int minSeeds,
               double hitFraction);
    // PrepareQueries (:157): seed selection over the query windows; returns the windows (queries are completed by
    // AddSequences, which scans them on the GPU together with the reads)
    int PrepareQueries(int numSeeds, i64 seedLimit, ValueView kmerValues, i64 firstSequence, i64 maxSeqs, int queryType = 1);
    // AddSequences (:217): GPU scan of every non-ignored read in [shardLo, shardHi) + all query windows
    int ScanLocal(size_t shardLo, size_t shardHi, Survivors& local, RoundStats& st);
    // chunkWorker (:253) + IndexSequences on the GPU, from the (possibly all-gathered) survivors
    int IndexSurvivors(const Survivors& all, RoundStats& st);
    // FindOverlaps (:320): Matches + prefilter + chaining + ratchet on the GPU
    // `pool` is storage reused across rounds (its SeedMatch objects keep their vector capacity); out points into it
    int FindOverlaps(std::vector<SeedMatch>& pool, std::vector<SeedMatch*>& out, RoundStats& st);
    // The same with everything after the chaining kept on the device (dp_find_overlaps without download + dp_consensus_paf):
    // FindOverlaps + the collation and finalCheckWorker of commands/overlap.go:158-233 for all query windows of the round.
    // Windows the device flags (they do not fit its layout) are done by the host path on fetched matches.
    // textOut != nullptr and a text pool set: the PAF text is left to a formatter thread (*textOut; `paf` stays untouched)
    int FindOverlapsAndFinalCheck(std::vector<SeedMatch>& pool, i64 overlapSize, std::string& paf, FinalCheckStats& fs,
                                  std::vector<int>* ignoreOut, RoundStats& st, std::shared_ptr<TextJob>* textOut = nullptr);
    void SetOverlapSize(i64 size) { overlap_ = size; }
    std::vector<SeedQuery> queries;
    std::string err;
    struct Window {
        uint32_t read, start, len;
    };
    const std::vector<Window>& windows() const { return windows_; }
    void setWindows(const std::vector<Window>& w) { windows_ = w; }
    // which reads count as ignored for PrepareQueries / ScanLocal (default: the read set's live flags)
    void setIgnoreView(const uint8_t* ig, uint64_t epoch = 0) {
        ignore_ = ig;
        ignoreEpoch_ = epoch;
    }
    void setWindowCache(WindowCache* c) { cache_ = c; }
    // multi-GPU: replaces the survivors of ScanLocal (this rank's read range) by the rank-ordered concatenation over all
    // ranks, exchanged device to device inside the library (dp_allgather_survivors)
    int ExchangeSurvivors(dp_comm* comm, Survivors& all);
    void setTextPool(TextPool* tp) { textPool_ = tp; }
    // chunkWorker + index build on the device (dp_index_build_chunked) for the rounds scanned by this object's own context: the
    // survivors' segments never come to the host (set before ScanLocal; IndexSurvivors falls back to the host chunking for
    // survivors it did not scan itself - the gathered ones of the multi-GPU scan-shard mode)
    void setDeviceChunking(bool on) { deviceChunkWanted_ = on; }
    void setPrechain(bool on) { prechainOk_ = on; }  // the survivors of ScanLocal are the ones IndexSurvivors will chunk (no exchange in between)
    i64 indexedSequences() const { return chunksOnDevice_ ? (i64)nIndexedExact_ : (i64)index_.sequences.size(); }

   private:
    bool deviceChunkWanted_ = false, chunksOnDevice_ = false, prechainOk_ = false;
    uint32_t nIndexedCap_ = 0, nIndexedExact_ = 0;
    int materializeChunks();  // the device-made chunks as host SeedSeq objects (index_.sequences), for the host consensus path
    void buildQueries(RoundStats& st);
    const Survivors* lastLocal_ = nullptr;  // the survivor set ScanLocal filled last (what this context's scan buffer holds)
    void chunkAndAdd(SeedSeq* s, uint64_t segBase, Arena& ar, std::vector<SeedSeq*>& seqOut, std::vector<dp_seq_ref>& refOut);
    const uint8_t* ignore_ = nullptr;
    uint64_t ignoreEpoch_ = 0;
    WindowCache* cache_ = nullptr;
    int prepareFromCache(int numSeeds, i64 seedLimit, ValueView values, i64 firstSequence, i64 maxSeqs);
    dp_ctx* ctx_;
    ReadSet& reads_;
    SeedIndex& index_;
    i64 chunkSize_, overlap_;
    int minSeeds_;
    double hitFraction_;
    std::vector<Window> windows_;
    void assembleQueries();
    bool assembled_ = false;  // querySegs_ / queryOff_ hold the current `queries`
    std::vector<int32_t> querySegs_;       // fwd/rc query segments (host)
    std::vector<uint64_t> queryOff_;
    TextPool* textPool_ = nullptr;
    dp_survivor_batch lastScan_;           // what dp_scan_reads returned for this round (valid until the context scans again)
    std::vector<int32_t> winSegs_;         // scan output of the windows
    std::vector<uint64_t> winOff_;
    const int32_t* allSegs_ = nullptr;     // survivors' segments on the host (device copy is what the index references)
};

unsigned hostThreads();  // DP_HOST_THREADS or hardware_concurrency (<= 96): size of the shared worker pool
void setHostThreadShare(unsigned concurrentUsers);
void profilePrint();  // DPH_PROFILE counters to stderr
void sampleProfStart();  // DPH_SAMPLE_PROF=1: CPU-time sampling by HIP API entry (host_pool.cpp), between start and stop
void sampleProfStop();
void sampleProfRegister(const char* role);  // a pipeline thread announces itself to the sampler
void sampleProfUnregister();
// Runs fn(i) for every i in [0, n) on the process-wide worker pool plus the calling thread and returns when all calls
// have finished.  Concurrent callers (executor slots, the planner) share the pool; jobs are served oldest first.
void parallelFor(size_t n, const std::function<void(size_t)>& fn);
// finalCheckWorker (commands/overlap.go:197-233) over the collated matches of a round: consensus, SetIgnore, PAF text.
// SetIgnore calls are returned in ignoreOut (query order) when it is non-null, otherwise applied to reads.ignore.
// ctx + jobs (optional): run the consensus alignment of all query windows on the device (dp_consensus_align); jobs is
// per-executor scratch kept across rounds.  Returns 0, or the device call's error code (text in *errOut).
int finalCheck(Arena& arena, const SeedIndex& index, ReadSet& reads, const std::vector<SeedMatch*>& matches, i64 numQuerySeqs,
               i64 overlapSize, std::string& paf, FinalCheckStats& fs, std::vector<int>* ignoreOut = nullptr,
               dp_ctx* ctx = nullptr, std::vector<ConsJob>* jobs = nullptr, std::string* errOut = nullptr, RoundStats* st = nullptr);
void finalCheck(Arena& arena, const SeedIndex& index, ReadSet& reads, std::vector<std::unique_ptr<SeedMatch>>& matches,
                i64 numQuerySeqs, i64 overlapSize, std::string& paf, FinalCheckStats& fs, std::vector<int>* ignoreOut = nullptr);

// ---- command driver (commands/overlap.go:96-233) ------------------------------------------------------------------
// One round's query batch, produced by seed selection alone (PrepareQueries): it depends on the read set, the value
// table, the ignore flags of reads >= firstIn and on nothing the GPU computes, so it can run ahead of the execution.
struct RoundPlan {
    i64 round = 0;
    i64 firstIn = 0, firstOut = 0;  // firstSequence before / after this round (commands/overlap.go:135-142)
    bool empty = true;              // len(queries) == 0 -> the command ends (:130)
    std::vector<Overlapper::Window> windows;
    std::vector<uint32_t> seedMap;  // seed id -> k-mer
    std::vector<int32_t> rcOf;      // seed id -> seed id of its reverse complement (SeedIndex::buildRcTable)
    bool failed = false;            // a device call of the planner failed; `error` holds the text
    std::string error;
};

// Runs the PrepareQueries chain ahead of the executing rounds on its own thread.  Plans are speculative with respect
// to ignore flags set by rounds that have not been committed yet; applyIgnores() discards every cached plan that a
// newly ignored read could have influenced (any plan whose firstIn <= that read id), so get() always returns what the
// sequential command would compute.
class Planner {
   public:
    // selCtx (may be null): device context whose resident reads + value table serve the speculative seed selection
    Planner(ReadSet& reads, const OverlapParams& p, ValueView values, bool threaded, dp_ctx* selCtx = nullptr,
            WindowCache* cache = nullptr);
    ~Planner();
    std::shared_ptr<const RoundPlan> get(i64 round);
    // commit-time: set the flags; returns the first round whose cached plan was discarded (or -1)
    i64 applyIgnores(const std::vector<int>& ids, i64 committedRound);
    uint64_t ignoreEpoch();  // bumped whenever a flag is set
    // rounds < round are committed; `round` starts at firstInOfRound (the committed firstSequence)
    void dropBefore(i64 round, i64 firstInOfRound);
    // planner lanes (threads that compute consecutive plans concurrently, each from a guess of where its predecessor ends)
    static int lanesFor(int world, int slots);
    static int lanesMax(int world, int slots);  // what OverlapRun::step may grow the lanes to while slots wait for plans
    int lanes() const;
    void setLanes(int n);  // grows only
    // Round-parallel runs (rounds dealt to `world` ranks): this planner computes the plans of the rounds its rank executes and
    // nothing else - where the rounds in between end is GUESSED (predictFirstOut: the window cache's seed counts, a microsecond
    // per round), so a rank's planner does 1 / world of the chain's work instead of all of it.  get(round) of an owned round then
    // returns a plan that starts where the guesses say the round starts; whether that is where the committed chain arrives is
    // checked where every speculative round is checked - at the commit (firstIn == firstSequence) - and a round whose guess
    // failed is planned again from the committed truth (dropBefore).  Needs the window cache (the guesses are its counts);
    // without it every rank walks the whole chain as before.
    void setOwnership(int rank, int world);
    bool sparse() const;

   private:
    std::shared_ptr<RoundPlan> compute(i64 round, i64 firstIn, SeedIndex& index);
    void laneMain(size_t lane);
    bool chainHead(i64* round, i64* firstIn) const;
    void promote();
    bool nextWork(i64* round, i64* firstIn) const;
    i64 predictFirstOut(i64 firstIn) const;
    i64 predictMemo(i64 firstIn) const;
    bool sparseNext(i64* round, i64* firstIn) const;
    int sparseEstimate(i64 round, i64* firstIn, std::shared_ptr<RoundPlan>* ended) const;
    struct Impl;
    std::unique_ptr<Impl> d;
};

// The PAF text of a round, written after the executor slot has gone on to its next round: the slot hands the device's 40-byte
// records over (copied out of the context's pinned buffer) and a formatter thread of the run turns them into lines
// (commands/overlap.go:223-228); whoever needs the text (the commit, the result exchange of the round-parallel mode) waits.
// (a round's 6-14 k records are a few hundred KB: buffers of that size come from mmap and go back with munmap - page faults on
// one side, an interrupt to every thread of the process on the other - so a job's two arrays are kept for the next job)
struct TextJobBuffers {
    static void take(std::vector<dp_paf_rec>& recs, std::vector<dp_group_meta>& groups);
    static void give(std::vector<dp_paf_rec>& recs, std::vector<dp_group_meta>& groups);
    // the text itself: a round's ~400 KB string lives until the job's text is given up (dph_overlap_reset), then serves a round of the
    // handle's next job (no mmap, no page faults, no munmap storm at the reset)
    static void takeText(std::string& s);
    static void giveTexts(std::vector<std::string>& v);
    static size_t releaseAll();  // dph_release_caches: every kept buffer goes back to the allocator; returns the bytes
};
struct TextJob {
    std::vector<dp_paf_rec> recs;
    std::vector<dp_group_meta> groups;
    TextJob() { TextJobBuffers::take(recs, groups); }
    ~TextJob() { TextJobBuffers::give(recs, groups); }
    std::vector<std::string> hostPaf;  // text of the windows the host consensus path did, indexed by hostOf[group]
    std::vector<uint32_t> hostOf;
    const ReadSet* reads = nullptr;
    std::string text;
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    void format();  // fills `text`
    void finish() {
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
};
// formatter threads of a run (two: a round's text is ~0.3 ms of one thread)
class TextPool {
  public:
    explicit TextPool(int nThreads);
    ~TextPool();
    void submit(std::shared_ptr<TextJob> job);

  private:
    void loop();
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<TextJob>> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

struct RoundResult {
    i64 round = 0;
    bool empty = true;
    i64 firstIn = 0, firstOut = 0, numQuerySeqs = 0;
    std::string paf;
    std::shared_ptr<TextJob> text;       // pending text of this round (appended to `paf` by takeText())
    void takeText() {
        if (!text) return;
        text->wait();
        if (paf.empty()) paf = std::move(text->text);
        else paf += text->text;
        text.reset();
    }
    std::vector<int> ignores;            // SetIgnore calls of this round, in order
    std::vector<uint32_t> indexedReads;  // read ids that entered the index (for speculation checks)
    std::vector<uint32_t> queryReads;    // read ids of the query windows
    i64 snapshot = 0;                    // rounds committed when the execution started (pipeline mode)
    i64 planRound = -2;                  // round of the plan this result was executed from (an empty result may come from
                                         // an EARLIER empty plan of a speculative chain); -2 = no plan at all
    FinalCheckStats fs;
    RoundStats st;
};

// Per-executor state: one GPU context (stream + per-round device buffers) and the host mirror of the running round.
struct ExecSlot {
    dp_ctx* ctx = nullptr;
    bool ownsCtx = false;
    std::unique_ptr<SeedIndex> index;  // executor-side seed maps of the running round
    std::unique_ptr<Overlapper> lap;
    Survivors local;
    int slotNo = 0;                    // position among the run's slots
    dp_comm* comm = nullptr;           // scan-shard mode: this slot's communicator (owned by the caller of OverlapRun)
    Survivors gathered;                // ... and the survivors of all ranks of the running round
    std::vector<SeedMatch> matchPool;  // reused across rounds
    std::vector<SeedMatch*> matches;
    std::vector<ConsJob> consJobs;     // consensus scratch per query window, reused across rounds
    std::string error;
};

struct OverlapRun {
    dp_ctx* ctx = nullptr;
    ReadSet* reads = nullptr;
    OverlapParams p;
    // k-mer value table (4^k doubles, 512 MiB at k=13).  Seed selection probes it at random, so it sits on transparent
    // huge pages (TLB reach) when the kernel allows it.
    struct HugeTable {
        double* p = nullptr;
        size_t n = 0, bytes = 0, mapped_ = 0;
        void* base_ = nullptr;
        void assign(const double* src, size_t count);
        double* reserve(size_t count);  // uninitialised table of `count` entries (filled by the caller)
        void clear();
        ~HugeTable() { clear(); }
        const double* data() const { return p; }
        size_t size() const { return n; }
    } values;
    // ... or, when the table was computed on the device, its 2-byte form (see ValueView); `values` is then filled on demand
    HugeTable valueCodes;            // 4^k uint16 (or uint8: codes8_) codes (the table's storage reused: bytes, not doubles)
    bool codes8_ = false;
    std::vector<double> valueLut;    // value of every code
    ValueView valueView() const {
        return valueLut.empty() ? ValueView(values.data())
                                : codes8_ ? ValueView((const uint8_t*)valueCodes.data(), valueLut.data())
                                          : ValueView((const uint16_t*)valueCodes.data(), valueLut.data());
    }
    const double* fullValues();      // the 4^k doubles (expands the 2-byte form the first time it is asked for)
    std::unique_ptr<TextPool> textPool;            // formatter threads (PAF text off the executor slots' critical path)
    std::vector<std::unique_ptr<ExecSlot>> slots;  // slot 0 drives `ctx`; further slots use contexts that borrow its reads
    std::unique_ptr<WindowCache> winCache;  // QueryEdges: the windows' round-independent part, produced ahead of the planner
    std::unique_ptr<Planner> planner;
    dp_ctx* plannerCtx = nullptr;  // borrows the reads and the value table of `ctx`
    dp_ctx* qualityCtx = nullptr;  // context whose resident reads carry their FASTQ quality bytes already (dp_quality_upload)
    i64 firstSequence = 0;
    i64 round = 0;  // next round to commit
    i64 numQuerySeqs = 0;
    i64 badBack = 0, emptyMatch = 0;
    bool done = false;
    std::string paf;      // PAF text of the last committed round(s)
    int adaptRounds_ = 0;           // step(): rounds since the planner's lanes were last looked at
    double adaptT_ = 0;
    double initEnd_ = 0;  // when init() returned (DPH_START_TRACE prints a job's first rounds against it)
    long long adaptWait_ = 0;
    std::atomic<long long> planWaitUs_{0};  // microseconds this handle's slots waited for plans (executeRoundOnImpl)
    int adaptLanes_ = 0;            // lanes the last job grew to (0: none yet)
    i64 pafLines = 0;     // ... and its number of lines
    std::string errText;  // stderr progress lines accumulated
    std::string error;    // failure text
    RoundStats last;
    RoundStats total;     // sums over every committed round of this job
    std::shared_ptr<const RoundPlan> curPlan;
    RoundResult cur;
    // shard of reads this process scans in scan-shard mode: [shardLo, shardHi)
    size_t shardLo = 0, shardHi = 0;

    int init(dp_ctx* c, ReadSet* r, const OverlapParams& params, const double* valuesOrNull, int nSlots = 1);
    ~OverlapRun();
    void shutdown(bool keepContexts = false);
    Survivors& local() { return slots[0]->local; }
    // ---- whole rounds on this process: plans (prefetched) -> execute (one round per slot, concurrently) -> commit in order.
    // Returns the number of rounds committed, 0 = finished, <0 error
    int step();
    // discards everything the executor pipeline has in flight (bench: start the timed region from an empty pipeline)
    void drain();
    // ---- pipelined round-parallel mode (several ranks): this rank's executor pipeline works on the rounds r with
    // r % world == rank; per superstep every rank contributes its next owned round (waitOwned), the results are
    // exchanged and commitGathered commits them in round order on every rank
    void setRanks(int rank, int world);
    // blocks until the owned round in [round, round+world) is ready; adds the finished owned rounds after it (maxRounds in all)
    int waitOwned(std::vector<RoundResult>& outs, int maxRounds = 1);
    int commitGathered(std::vector<RoundResult>& results);  // rounds committed; rejected owned rounds are re-queued
    // executes rounds[i] on slot i concurrently (host threads); outs[i] receives the result
    int executeRounds(const std::vector<i64>& rounds, std::vector<RoundResult>& outs);
    // ---- scan-shard mode (survivor all-gather between the two halves)
    int roundPrepareAndScan();
    int roundFinish(const Survivors& all);
    // ... with the exchange inside the library (RCCL or in-process peers): one whole round; 1 ran, 0 finished, < 0 error
    dp_comm* comm = nullptr;  // owned by the caller
    int roundSharded();
    // ... with executor slots: slot i has its own communicator (slotComms[i], owned by the caller; every rank creates them in
    // the same order).  One call executes the next slots.size() rounds concurrently - slot i takes round + i: its shard's scan,
    // the exchange on its communicator, then index, query, consensus - and commits them in order with the speculation check of
    // commitResults (the rounds ran against the flags committed before the batch).  Every rank sees the same results and
    // therefore commits the same prefix: the collectives of the ranks stay matched without further agreement.
    // Returns the number of rounds committed, 0 = finished, <0 error.
    std::vector<dp_comm*> slotComms;
    int roundsShardedBatch();
    // The slots of a batch exchange on their own communicators, all on this rank's one device: RCCL wants collectives of several
    // communicators that share a device issued in the same order on every rank, so the exchanges of a batch take turns in slot
    // order (slot i after slot i - 1, on every rank alike); everything else of the rounds still runs concurrently.
    std::mutex exchangeMu_;
    std::condition_variable exchangeCv_;
    int exchangeTurn_ = 0, resultTurn_ = 0;
    bool exchangeOrdered_ = false;
    // scan-shard with per-slot communicators: the query windows of a round are dealt to the ranks as well ("queries
    // are partitioned by QueryID") - every rank builds the whole index from the gathered survivors but queries, chains and
    // builds the consensus for its own contiguous share of the windows only; the ranks' PAF text, SetIgnore ids and counters are
    // all-gathered (dp_allgather_blobs on the slot's communicator) and joined in rank order = query order.  DPH_SHARD_QUERIES=0:
    // every rank does every window (round 2's behaviour).
    bool shardQueries = true;
    void abortComms();
    // ---- round-parallel mode: execute round `r` speculatively against the current flags, commit gathered results
    int executeRound(i64 r, RoundResult& out);
    // returns the number of rounds committed from `results` (in order, all consecutive from `round`), stopping at the
    // first one invalidated by an earlier round's ignores; sets done when an empty round is reached
    int commitResults(std::vector<RoundResult>& results);

   private:
    int executeRoundOn(ExecSlot& s, i64 r, RoundResult& out);
    int executeRoundOnImpl(ExecSlot& s, i64 r, RoundResult& out);
    int beginRound(ExecSlot& s, const RoundPlan& plan);
    int finishRound(ExecSlot& s, const Survivors& all, RoundResult& out);
    void commitOne(RoundResult& r);
    void commitText(RoundResult& r);   // the half of a commit nobody else looks at (text, counters)
    void commitState(RoundResult& r);  // the half the executor slots and the planner see (under the pipeline's lock)
    // executor pipeline behind step(): every slot has a worker thread that keeps executing the next uncommitted round
    // speculatively; step() commits finished rounds in order and re-queues the ones a later-arriving ignore flag
    // invalidated
    void startWorkers();
    void workerMain(size_t slot);
    bool resultValid(const RoundResult& r) const;
    bool emptyResultValid(const RoundResult& r) const;
    std::vector<std::thread> workers_;
    std::mutex pmu_;
    std::condition_variable cvWork_, cvDone_;
    std::map<i64, RoundResult> ready_;
    std::deque<i64> redo_;
    i64 nextIssue_ = 0;
    int rank_ = 0, world_ = 1;
   public:
    i64 roundLimit = -1;  // step() commits no round >= this (-1: no limit)
   private:
    int inflight_ = 0, workerRc_ = 0;
    bool stopWorkers_ = false, issueEnd_ = false, draining_ = false;
    std::string workerErr_;
    std::vector<int32_t> flagRound_;  // round whose commit flagged each read, -1 = not flagged by this run
    Survivors gathered_;              // scan-shard mode: the survivors of all ranks
};

// ---- mapping.Mapper / `downpore map` (mapping/mapping.go, commands/map.go) ----------------------------------------
struct MapParams {  // flag table commands/map.go:19-20
    bool circular = true;
    int k = 11;
    i64 querySize = 1000, minLength = 500, chunkSize = 10000, seedRate = 40;
    int numWorkers = 4;
};
struct MapStats {
    uint64_t n_chunks = 0, n_seeds = 0, n_windows = 0, n_chains = 0, n_batches = 0;
    double k_scan_ms = 0, k_map_ms = 0;
    double map_bytes = 0, scan_bytes = 0;  // algorithmic bytes: dp_map_windows' query + prefilter + chaining; packed bases of the windows scanned
    double t_setup_s = 0, t_scan_s = 0, t_chain_s = 0, t_host_s = 0;  // wall: indexing the reference / window scans / dp_map_windows / coroutines
};
// Runs the whole command on HIP device `device`: reference = first sequence of refSet (top-level, cache=false), reads
// top-level.  paf receives the PAF lines (read order), errText the reference's stderr lines.
size_t releaseMapStaging();  // dph_release_caches: the staging block kept for the process's next map command
int runMap(const ReadSet& refSet, const ReadSet& reads, const MapParams& p, int device, std::string& paf, std::string& errText,
           MapStats* stats, std::string& error);

// ---- flag table helpers (commands/command.go:18-74, downpore.go:34-51) -------------------------------------------
struct ArgTable {
    std::vector<std::string> names, defaults, descriptions;
    std::unordered_map<std::string, std::string> args, alias;
    void make(const std::vector<std::string>& n, const std::vector<std::string>& d, const std::vector<std::string>& desc);
    bool parse(int argc, char** argv, std::string& err);
};

// test hooks (host_capi.cpp: dph_hand_*)
void trimBestIndices(int upto, const std::vector<SeedMatch*>& ms, int minMatch, int length, int* bestOut, int* backOut);  // host_seq.cpp
long coroSelfTest(int nTasks, int yields);  // host_map.cpp: the read tasks' stack switch by itself
bool handIsConsistent(const i64* l, const i64* r, bool circular, i64 refLen);                                                // host_map.cpp
int handRemoveDominated(const i64* maps, int n, i64 queryLen, int* kept);                                                    // host_map.cpp
}  // namespace dph
