// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (single-threaded, deterministic C++17) of the reference's seed-index +
// seed-chaining overlap/map path.  Nothing in the product (downpore_amd/, include/) may include,
// link or execute this code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// use it, and only as the checker.
//
// Parity status: the reference is Go + Plan-9 amd64 assembly; no Go toolchain exists in the build
// container, so the reference itself cannot be executed here.  The oracle is pinned against every
// known-answer vector the reference's own unit tests hold (sequence/sequence_test.go,
// util/bitset_test.go — see tests/test_oracle_known_answers.py).  Everything above the primitives
// (seed selection, chunking, Matches, both chainers, consensus, mapping, PAF text) has NO reference
// test or golden output: for those rows the oracle is "PARITY UNPINNED" — its fidelity rests on the
// line-by-line citations below.
//
// Canonical semantics chosen where the reference is racy / toolchain dependent:
//   * single worker, fully synchronous: a query window's seeds are committed before the seed-budget
//     test for the next read (overlap/overlap.go:58); sequences are indexed in file order; queries are
//     [fwd0, rc0, fwd1, rc1, ...]; matches are emitted in query order, candidates ascending.
//   * sort.Sort (Go pdqsort) tie order: slices of <=12 elements are insertion sorted (that IS what
//     Go >=1.19 does, and it is stable); longer slices use a stable sort, i.e. "ties by original
//     index" (documented divergence risk).  TopOccurrences' 4^k-element sort: ascending by
//     (count, kmer id); top-N = the last N.
//   * getSoftUnion8Asm with n<=5 (registers v1..v4 uninitialised in the reference,
//     util/asm_amd64.s:200-207) is restated zero-initialised.
//   * overlap/combine.go:93-102 prints a "Bad back:" diagnostic into the PAF stream whose text contains
//     Go pointer values; it is suppressed (counted in OverlapResult.badBack).  A part whose match
//     becomes empty there — or whose (off-by-one, commands/overlap.go:224) match indexes past the trimmed
//     consensus — makes GetBasesCovered panic with an index out of range; canonically that PAF line is still
//     printed, with ident = 0 (counted in emptyMatchPanics).
//   * 8-byte loads that run past the end of a read's packed bytes see zero bytes (the reference
//     reads whatever follows on the Go heap).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstddef>
#include <memory>
#include <string>
#include <functional>
#include <vector>

namespace dpo {

typedef int64_t i64;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// sequence/sequence.go:43-53 packedSequence (a view into shared packed bytes)
struct PackedSeq {
    std::shared_ptr<std::vector<uint8_t>> buf;  // parent bytes + >=8 bytes of zero padding
    size_t lo = 0, hi = 0;                      // data = buf[lo:hi]
    i64 id = 0;
    i64 offset = 0, inset = 0;
    std::shared_ptr<std::string> name;          // nil => decimal id
    i64 length = 0;
    int firstLen = 4, finalLen = 0;
    // FASTQ quality (phred byte - 33, sequence/seqio.go:169-173,231-236); nil unless the record's quality line had exactly
    // the sequence's length.  A view shares its parent's array (sequence.go:366-368): quality of base i = (*qual)[qlo + i].
    std::shared_ptr<std::vector<uint8_t>> qual;
    size_t qlo = 0;

    size_t nbytes() const { return hi - lo; }
    const uint8_t* data() const { return buf->data() + lo; }
    std::string getName() const;
    std::string str() const;                                    // sequence.go:242-276
    PackedSeq subSequence(i64 start, i64 end) const;            // sequence.go:353-370
    PackedSeq reverseComplement() const;                        // sequence.go:179-198
    PackedSeq append(i64 id, const PackedSeq& other) const;     // sequence.go:164-177 (name nil)
    i64 kmerAt(i64 index, int k) const;                         // sequence.go:440-442 + asm:3-30
    i64 nextKmer(i64 cur, i64 mask, i64 nextBaseIndex) const;   // sequence.go:447-453
    i64 countKmers(i64 upTo, int k, const uint8_t* seeds) const;                        // :329
    i64 countKmersBetween(i64 from, i64 to, i64 upTo, int k, const uint8_t* seeds) const;  // :332
    void writeSegments(i64* segments, int k, const uint8_t* seeds) const;               // :338
};
uint8_t baseCode(uint8_t b);                                    // ((b>>1)^((b&4)>>2))&3
PackedSeq newPackedSequence(i64 id, const std::string& seq, std::shared_ptr<std::string> name);  // :67-93
// asm restatements (sequence/asm_amd64.s), exposed for the known-answer tests
void packBytesAsm(const uint8_t* seq, size_t n, uint8_t* out);                       // asm:33-78
i64 packedCountKmersAsm(const uint8_t* data, i64 nbytes, i64 upTo, i64 skipFront, i64 skipBack, int k,
                        const uint8_t* seeds);                                       // asm:81-203
i64 packedWriteSegmentsAsm(const uint8_t* data, i64 nbytes, i64 skipFront, i64 skipBack, int k,
                           const uint8_t* seeds, i64* segments);                     // asm:206-394 (returns #values written)

// byteSequence reference implementation used by the reference's differential tests
// (sequence/sequence.go:278-324).  Only what the known-answer tests need.
i64 byteCountKmers(const std::string& s, i64 upTo, int k, const uint8_t* seeds);
i64 byteWriteSegments(const std::string& s, int k, const uint8_t* seeds, i64* segments);

// ---------------------------------------------------------------------------------------------
// util/bitset.go:13-153
struct IntSet {
    std::vector<u64> vs;
    u64 start = 1, end = 0, count = 0;
    IntSet() : vs(50, 0) {}                                     // NewIntSet :20
    explicit IntSet(i64 capacity) : vs((size_t)(capacity / 64 + 1), 0) {}  // NewIntSetCapacity :25
    static IntSet fromUInts(const std::vector<u64>& values);    // :43-57
    bool contains(u64 x) const;                                 // :65
    void add(u64 x);                                            // :74
    void clear();                                               // :145
    bool isEmpty() const { return start > end; }
    u64 size() const { return count; }
    u64 countIntersectionTo(const IntSet& other, i64 maxCount) const;  // :179 + asm:14-117
    u64 countIntersection(const IntSet& other) const;           // :163
};
std::vector<u64> getSharedIDs(const std::vector<const IntSet*>& sets, i64 minCount, bool fast);  // :308-411
void softUnion4(const u64* vs, i64 n, u64 out[4]);              // asm:121-193
void softUnion8(const u64* vs, i64 n, u64 out[4]);              // asm:196-314 (v5..v8)
void softUnion16(const u64* vs, i64 n, u64 out[4]);             // asm:317-509 (v13..v16), incl. step-8 defect

// ---------------------------------------------------------------------------------------------
struct SeedIndex;
// seeds/sequence.go:10-20
struct SeedSequence {
    std::shared_ptr<std::vector<i64>> store;  // backing array (shared by sub-sequence views)
    size_t lo = 0, n = 0;                     // segments = store[lo : lo+n]
    i64 id = 0;
    std::shared_ptr<std::string> name;
    i64 length = 0, offset = 0, inset = 0;
    SeedSequence* reverseComplement = nullptr;
    bool rc = false;
    SeedSequence* Parent = nullptr;

    i64* seg() { return store->data() + lo; }
    const i64* seg() const { return store->data() + lo; }
    i64 numSeeds() const { return (i64)n / 2; }
    i64 getSeed(i64 i) const { return seg()[i * 2 + 1]; }
    i64 getSeedOffset(i64 index, int k) const;        // :1239
    i64 getSeedOffsetFromEnd(i64 index, int k) const; // :1269
    i64 getNextSeedOffset(i64 index, int k) const { return seg()[index * 2 + 2] + k; }  // :1278
    i64 getMaxSeed() const;                           // :1286
};

// seeds/sequence.go:24-32
struct SeedMatch {
    std::vector<i64> MatchA, MatchB;
    SeedSequence* SeqA = nullptr;
    SeedSequence* SeqB = nullptr;
    i64 QueryID = 0;
    bool ReverseComplementQuery = false;
};

// Arena that owns every SeedSequence created in a round (Go GC stand-in).
struct Arena {
    std::vector<std::unique_ptr<SeedSequence>> seqs;
    SeedSequence* make() { seqs.emplace_back(new SeedSequence()); return seqs.back().get(); }
};

// seeds/seeds.go:11-21
struct SeedIndex {
    int seedSize;
    std::vector<uint8_t> kmers;        // []bool
    std::vector<SeedSequence*> sequences;
    std::vector<IntSet> sequenceSets;  // seed -> set of sequence indices
    std::vector<IntSet> seedSets;      // sequence -> set of seeds
    std::vector<int32_t> kmerMap;
    std::vector<i64> seedMap;
    i64 size = 0;
    Arena arena;

    explicit SeedIndex(int k);
    SeedSequence* newSeedSequence(const PackedSeq& seq);                       // :33-50
    std::shared_ptr<std::vector<i64>> scanSegments(const PackedSeq& seq) const;  // its CountKmers+WriteSegments half
    SeedSequence* newSeedSequence(const PackedSeq& seq, std::shared_ptr<std::vector<i64>> store);
    void addSeeds(const PackedSeq& seq, i64 minSeeds, const double* ranks);    // :62-156
    void addSingleSeeds(const PackedSeq& seq, i64 seedRate, const double* ranks);  // :160-200
    void addSequence(SeedSequence* s);                                         // :272-290
    void indexSequences();                                                     // :292-305,372-384
    std::vector<u64> matches(const SeedSequence* q, double hitFraction) const; // :335-353
    void addSeedKmer(i64 kmer);  // the "if !g.kmers[kmer] {...}" block :132-141
};
u64 reverseComplementKmer(u64 seed, int k);                                    // sequence.go:125-132
SeedSequence* ssReverseComplement(SeedSequence* s, int k, SeedIndex& index);   // sequence.go:134-159
SeedSequence* ssSubSequence(Arena& a, SeedSequence* s, i64 start, i64 end, i64 length, i64 offset, i64 inset);  // :46
SeedSequence* ssTrimmed(Arena& a, SeedSequence* s, i64 startOffset, i64 startSeed, i64 endOffset, i64 endSeed,
                        int k, i64* startSeedOut);                             // :54-82
SeedSequence* ssReduced(Arena& a, SeedSequence* s, const IntSet& whitelist, int k, i64 minSeeds,
                        std::vector<i64>* index);                              // :85-123
void smReverseComplement(SeedMatch& m, int k, SeedIndex& index);               // :800-816
bool smGetBasesCovered(const SeedMatch& m, int k, i64* a, i64* b);             // :830-858; false = reference panics
void smGetBaseIndex(const SeedMatch& m, i64 aIndex, int k, i64* index, i64* bases, i64* distance);  // :1190-1237

// seeds/alignment.go:274-616
struct SeedAligner {
    i64 maxLength;
    explicit SeedAligner(i64 maxLen) : maxLength(maxLen) {}
    // returns matches in the reference's (reversed) order; empty == nil
    std::vector<SeedMatch> pairwiseAlignments(SeedSequence* a, SeedSequence* b, const IntSet& aSet,
                                              const IntSet& bSet, i64 minMatches, int k);
};
void gapRange(i64 gap, int k, i64* minGap, i64* maxGap);                       // alignment.go:411-424

// seeds/sequence.go:361-576 (map chaining)
std::vector<SeedMatch> ssMatch(Arena& a, SeedSequence* seq, SeedSequence* query, const IntSet* querySet,
                               const IntSet* seqSet, i64 minMatch, int k);

// seeds/alignment.go:23-268
SeedSequence* multiAlignerConsensus(Arena& a, std::vector<SeedSequence*>& seqs, int k,
                                    std::vector<std::unique_ptr<SeedMatch>>& matchesOut);

// ---------------------------------------------------------------------------------------------
// sequence/seqio.go fastaSequenceSet (FASTA / FASTQ, one line per read)
struct FastaSet {
    std::vector<PackedSeq> cached;      // top-level sequences in file order
    std::vector<uint8_t> ignore;
    std::vector<std::string> names;
    std::vector<i64> lengths;
    bool himem = true;
    i64 bases = 0;
    static FastaSet fromFile(const std::string& path, i64 minLen, bool himem);        // readFasta :106
    static FastaSet fromReads(const std::vector<std::string>& names, const std::vector<std::string>& seqs,
                              i64 minLen, bool himem, const std::vector<std::string>* quals = nullptr);
    // qualLine (FASTQ): the record's quality line as ReadBytes returned it, or nullptr
    void addLine(const std::string& lastName, const std::string& line, i64 minLen, const std::string* qualLine = nullptr);
    bool isFastq = false;
    std::string error;  // "Invalid fastq format ..." (the reference calls log.Fatal)
    // the sequence object a later pass would receive (:115 cached view, or :158 top-level re-read)
    PackedSeq served(size_t id) const;
    size_t size() const { return cached.size(); }
};

// util/sequtil/kmers.go:34-112 + commands/overlap.go:39-94 / commands/map.go:45-71
std::vector<u64> kmerOccurrences(const std::vector<PackedSeq>& seqs, int k);
std::vector<double> kmerValues(std::vector<u64>& counts, int k);  // counts are modified like the reference (:90-96)

// ---------------------------------------------------------------------------------------------
// overlap/overlap.go
struct SeedQuery {
    i64 ID = 0, SequenceID = 0;
    SeedSequence* Query = nullptr;
    bool AtStart = true, ReverseComplement = false;
};
struct OverlapParams {
    i64 overlapSize = 1000;
    int k = 10;
    i64 numSeeds = 15, seedBatchSize = 10000, chunkSize = 10000, queryBatchSize = 20000;
    double minHits = 0.25;
    bool himem = true;
    int queryType = 1;  // overlap.QueryEdges (overlap.go:18-21): 1 edges, 2 centre, 4 all, +8 weight edges
};
struct Overlapper {
    SeedIndex& index;
    i64 chunkSize, overlap, minSeeds;
    double hitFraction;
    Overlapper(SeedIndex& ix, i64 chunk, i64 ov, i64 minS, double hf)
        : index(ix), chunkSize(chunk), overlap(ov), minSeeds(minS), hitFraction(hf) {}
    std::vector<SeedQuery> prepareQueries(i64 numSeeds, i64 seedLimit, const double* values,
                                          const std::vector<PackedSeq>& seqs, int queryType = 1);   // :157
    void chunkAndAdd(SeedSequence* s);                                          // chunkWorker :253-318
    void addSequences(const std::vector<PackedSeq>& seqs);                      // :217
    std::vector<std::unique_ptr<SeedMatch>> findOverlaps(const std::vector<SeedQuery>& queries);  // :320 + matchWorker :346
};

// pieces of overlap.go / combine.go / mapping.go that tests/test_hand_known_answers.py calls on bare numbers (capi.cpp: dpo_hand_*)
void chunkPieces(Arena& ar, SeedSequence* s, i64 chunkSize, i64 overlap, i64 minSeeds, int k, const std::function<void(SeedSequence*)>& add);  // overlap.go:253-318
void trimBestIndices(i64 upto, const std::vector<SeedMatch*>& ms, i64 minMatch, i64 length, i64* bestOut, i64* backOut);                    // combine.go:24-58

// overlap/combine.go
struct SeedContig {
    SeedSequence* Combined = nullptr;
    std::vector<i64> Parts, Offsets, Lengths, SeqLengths;
    std::vector<uint8_t> ReverseComplement, Approximate;
    std::vector<SeedMatch*> Matches;
    std::vector<std::unique_ptr<SeedMatch>> owned;
};
// badBack counts the "Bad back:" events (combine.go:93-102) — see the canonical-semantics note above.
std::unique_ptr<SeedContig> buildConsensus(SeedIndex& sg, std::vector<SeedMatch*>& overlaps, i64* badBack);

// Per-round trace a test can compare stage by stage.
struct RoundTrace {
    std::vector<i64> seedKmers;                    // seedMap after PrepareQueries
    i64 firstSequence = 0, numQuerySeqs = 0;
    std::vector<std::vector<i64>> querySegments;   // per SeedQuery
    std::vector<i64> queryIDs, querySeqIDs, queryLength, queryOffset, queryInset;
    std::vector<std::vector<i64>> indexedSegments; // per indexed sequence (after chunking)
    std::vector<i64> indexedIds, indexedLength, indexedOffset, indexedInset;
    std::vector<std::vector<u64>> candidates;      // per query: Matches() output
    // matches in emission order
    std::vector<i64> matchQueryIndex, matchTarget;
    std::vector<std::vector<i64>> matchA, matchB;
    i64 hits = 0, qHits = 0;
    std::string paf;
    std::vector<i64> newlyIgnored;
};
struct OverlapResult {
    std::string paf;      // everything the reference would print to stdout
    std::string err;      // stderr progress lines
    i64 rounds = 0;
    i64 badBack = 0;           // "Bad back:" diagnostics suppressed
    i64 emptyMatchPanics = 0;  // PAF lines where the reference would have panicked (ident printed as 0)
    std::vector<RoundTrace> traces;  // filled when keepTraces
};
OverlapResult runOverlap(FastaSet& set, const OverlapParams& p, const double* valuesOrNull, i64 maxRounds,
                         bool keepTraces);

// ---------------------------------------------------------------------------------------------
// mapping/mapping.go
struct MapParams {
    bool circular = true;
    int k = 11;
    i64 querySize = 1000, minLength = 500, chunkSize = 10000, seedRate = 40;
};
struct Mapping {
    const PackedSeq* Query = nullptr;
    i64 Start = 0, End = 0, QueryOffset = 0, QueryInset = 0;
    bool RC = false;
    i64 ids = 0;
};
bool mappingsConsistent(const Mapping* left, i64 leftQueryLen, const Mapping* right, bool circular, i64 referenceLength);  // mapping.go:131-160
std::vector<Mapping*> removeDominated(std::vector<Mapping*> open, const std::vector<Mapping*>* extendedIn, i64 queryLen);  // mapping.go:387-428
struct Mapper {
    SeedIndex index;
    PackedSeq reference;
    i64 edgeSize;
    bool circular;
    std::vector<std::unique_ptr<Mapping>> pool;
    Mapper(const PackedSeq& ref, bool circ, int k, const double* values, i64 seedRate, i64 edge, i64 chunk);  // :67
    std::vector<Mapping*> map(const PackedSeq& query);                     // :430
    std::vector<Mapping*> performMapping(const PackedSeq& query);          // :489
    std::string asString(const Mapping& m) const;                          // :112
    Mapping* mk();
    bool isConsistent(const Mapping* l, const Mapping* r) const;           // :131
    void matchPairs(std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& matched,
                    bool& matchedNil);                                     // :174
    void findSplitPoint(const PackedSeq& query, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB,
                        i64 left, i64 right);                              // :207
    void mapNext(const PackedSeq& query, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB,
                 std::vector<Mapping*>& newA, std::vector<Mapping*>& newB, std::vector<Mapping*>& matched,
                 bool& matchedNil);                                        // :305
};
struct MapResult { std::string paf, err; };
MapResult runMap(FastaSet& refSet, FastaSet& reads, const MapParams& p);

// Go sort.Sort stand-in (see header note): insertion sort for n<=12, stable sort otherwise.
template <class T, class Less>
void goSort(std::vector<T>& v, Less less) {
    size_t n = v.size();
    if (n <= 12) {
        for (size_t i = 1; i < n; i++)
            for (size_t j = i; j > 0 && less(v[j], v[j - 1]); j--) std::swap(v[j], v[j - 1]);
    } else {
        std::stable_sort(v.begin(), v.end(), less);
    }
}

}  // namespace dpo
