// Host-side seed space for the product (see dph.hpp): read set, value table, seed selection, SeedSequence
// operations, seed-space consensus and contig building.  File:line citations are into the reference.
#include <immintrin.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "host_util.hpp"

namespace dph {

// ---------------------------------------------------------------------------------------------------------------
// sequence/seqio.go readFasta :106-276 (FASTA subset: single-line sequences)

static std::string trimSpace(const std::string& s) {
    size_t a = 0, b = s.size();
    auto sp = [](unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (a < b && sp((unsigned char)s[a])) a++;
    while (b > a && sp((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// `line` is one ReadBytes('\n') result including its last byte; kept iff len(line) >= minLen; the stored sequence
// drops the last byte (seqio.go:210-219).
void ReadSet::addLine(const std::string& lastName, const char* line, size_t len, i64 minLen, const char* qualLine, size_t qualLen) {
    if ((i64)len >= minLen) {
        if (off.empty()) off.push_back(0);
        names.push_back(trimSpace(lastName));
        maxNameLen = std::max(maxNameLen, names.back().size());
        bases.append(line, len - 1);
        off.push_back((i64)bases.size());
        ignore.push_back(0);
        if (isFastq) {
            // :229-238 the quality line counts only when it is exactly one byte longer than the sequence; every byte has 33
            // subtracted as a Go byte (modulo 256)
            const bool ok = qualLine && qualLen == len;
            qual.resize(bases.size() - (len - 1), 0);
            if (ok)
                for (size_t i = 0; i + 1 < len; i++) qual.push_back((char)(uint8_t)((unsigned char)qualLine[i] - 33));
            else
                qual.resize(bases.size(), 0);
            hasQual.resize(names.size() - 1, 0);
            hasQual.push_back(ok ? 1 : 0);
        }
    }
}

bool ReadSet::fromFile(const std::string& path, i64 minLen, bool himem, ReadSet& f, std::string& err) {
    f = ReadSet();
    f.himem = himem;
    f.off.push_back(0);
    // the file is mapped, not copied: one pass of memchr over the page cache, one copy of the bases into the read set
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) {
        err = "cannot open " + path;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        err = "cannot stat " + path;
        return false;
    }
    const size_t size = (size_t)st.st_size;
    if (size == 0) {
        close(fd);
        return true;
    }
    void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        err = "cannot map " + path;
        return false;
    }
    madvise(m, size, MADV_SEQUENTIAL);
    const char* all = (const char*)m;
    f.bases.reserve(size);
    size_t pos = 0;
    auto next = [&](size_t& b, size_t& e) -> bool {
        if (pos >= size) return false;
        const void* nl = memchr(all + pos, '\n', size - pos);
        b = pos;
        e = nl ? (size_t)((const char*)nl - all) + 1 : size;
        pos = e;
        return true;
    };
    size_t b, e;
    if (next(b, e) && all[e - 1] == '\n') {  // :191-196 first line is always a name line; '@' makes the file a FASTQ
        if (all[b] == '@') f.isFastq = true;
        std::string lastName(all + b + 1, e - b - 1);
        while (next(b, e)) {
            bool eof = all[e - 1] != '\n';
            const unsigned char c = (unsigned char)all[b];
            if (c >= 'A' && c <= 'T') {
                if (f.isFastq) {  // :222-238 / :246-255 the '+' line and the quality line follow, kept read or not
                    size_t pb = 0, pe = 0, qb = 0, qe = 0;
                    const bool gotPlus = next(pb, pe);
                    if (!gotPlus || all[pe - 1] != '\n' || all[pb] != '+') {
                        err = "Invalid fastq format (on + line):" + (gotPlus ? std::string(all + pb, pe - pb) : std::string());
                        munmap(m, size);
                        return false;  // the reference calls log.Fatal
                    }
                    const bool gotQual = next(qb, qe);
                    f.addLine(lastName, all + b, e - b, minLen, gotQual ? all + qb : nullptr, gotQual ? qe - qb : 0);
                    eof = false;  // the loop's `err` is the '+' line's from here on (:224), and that line was complete
                } else {
                    f.addLine(lastName, all + b, e - b, minLen);
                }
            } else if (c == '@') {
                f.isFastq = true;
                lastName.assign(all + b + 1, e - b - 1);
            } else {
                lastName.assign(all + b + 1, e - b - 1);
            }
            if (eof) break;
        }
    }
    if (f.isFastq) {  // reads kept before the first '@' line was seen (none in a well-formed file) carry no quality
        f.qual.resize(f.bases.size(), 0);
        f.hasQual.resize(f.names.size(), 0);
    }
    munmap(m, size);
    return true;
}

ReadSet ReadSet::fromArrays(const char* bases, const i64* off, size_t n, i64 minLen, bool himem, const char* quals) {
    ReadSet f;
    f.himem = himem;
    f.isFastq = quals != nullptr;
    f.off.push_back(0);
    char nm[32];
    std::string line, ql;
    for (size_t i = 0; i < n; i++) {
        snprintf(nm, sizeof nm, "r%07zu\n", i);
        line.assign(bases + off[i], (size_t)(off[i + 1] - off[i]));
        line.push_back('\n');
        if (quals) {
            ql.assign(quals + off[i], (size_t)(off[i + 1] - off[i]));
            ql.push_back('\n');
            f.addLine(nm, line.data(), line.size(), minLen, ql.data(), ql.size());
        } else {
            f.addLine(nm, line.data(), line.size(), minLen);
        }
    }
    return f;
}

i64 ReadSet::scanKmers(size_t r, int k) const {
    i64 L = length(r);
    i64 n = L - k + 1;
    if (!himem && (L % 4) == 0) n -= 4;  // top-level re-read: finalLen == 0 (sequence.go:70,88; asm:88-96)
    return n < 0 ? 0 : n;
}

uint32_t reverseComplementKmer(uint32_t seed, int k) {  // complement every 2-bit code, reverse their order (k <= 16)
    uint32_t x = ~seed;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);
    return k >= 16 ? x : (x >> (32 - 2 * k));
}

// ---------------------------------------------------------------------------------------------------------------
// commands/overlap.go:55-93 + util/sequtil/kmers.go:87-112.  float64, no contraction (built with -ffp-contract=off).
// Tie rule for the unstable 4^k-element sort: ascending (count, k-mer id); top-N = last N.

std::vector<double> kmerValuesFromCounts(std::vector<uint64_t>& counts, int k) {
    const size_t n = counts.size();
    std::vector<double> values(n, 0.0);
    uint64_t tot = 0, maxCount = 0;
    for (uint64_t c : counts) {
        tot += c;
        maxCount = std::max(maxCount, c);
    }
    const double tf = (double)tot;
    const double targetFreq = 0.000005;
    auto valueOf = [&](uint64_t count) -> double {
        const double freq = (double)count / tf;
        if (count < 3) return 0.0;
        if (freq <= targetFreq) return 1.0 - (targetFreq - freq);
        return 1.0 - (freq - targetFreq);
    };
    // the value is a function of the count alone: one exact evaluation per distinct count, 4^k table lookups
    const uint64_t kLut = (uint64_t)1 << 22;
    if (maxCount < kLut) {
        std::vector<double> lut((size_t)maxCount + 1);
        for (uint64_t c = 0; c <= maxCount; c++) lut[(size_t)c] = valueOf(c);
        for (size_t i = 0; i < n; i++) values[i] = lut[(size_t)counts[i]];
    } else {
        for (size_t i = 0; i < n; i++) values[i] = valueOf(counts[i]);
    }
    for (size_t i = 0; i < n; i++) {  // in-place fwd+rc merge, sequential semantics (kmers.go:90-96)
        const size_t rc = reverseComplementKmer((uint32_t)i, k);
        const uint64_t c = counts[i] + counts[rc];
        counts[i] = c;
        counts[rc] = c;
    }
    const size_t topN = n / 100;
    if (topN > 0) {
        // T = the (n - topN)-th smallest merged count (what nth_element would put there)
        uint64_t T = 0, maxMerged = 0;
        for (uint64_t c : counts) maxMerged = std::max(maxMerged, c);
        if (maxMerged < kLut) {
            std::vector<uint64_t> hist((size_t)maxMerged + 1, 0);
            for (uint64_t c : counts) hist[(size_t)c]++;
            uint64_t cum = 0;
            for (uint64_t v = 0; v <= maxMerged; v++) {
                cum += hist[(size_t)v];
                if (cum > (uint64_t)(n - topN)) {
                    T = v;
                    break;
                }
            }
        } else {
            std::vector<uint64_t> tmp(counts);
            std::nth_element(tmp.begin(), tmp.begin() + (n - topN), tmp.end());
            T = tmp[n - topN];
        }
        size_t above = 0;
        for (uint64_t c : counts) above += c > T;
        size_t ties = topN - above;
        for (size_t i = n; i-- > 0;) {
            if (counts[i] > T) values[i] = 0;
            else if (counts[i] == T && ties > 0) {
                values[i] = 0;
                ties--;
            }
        }
    }
    values[0] = 0;
    return values;
}

// ---------------------------------------------------------------------------------------------------------------
// seeds.SeedIndex host mirror

SeedIndex::SeedIndex(int k_, int preBits) : k(k_), preShift((uint32_t)(32 - preBits)) {
    hkeys.assign(1u << 16, 0xffffffffu);
    hvals.assign(1u << 16, -1);
    hmask = (1u << 16) - 1;
    pre.assign(((size_t)1 << preBits) / 32, 0);
}

void SeedIndex::reset() {
    hashValid = true;
    rcPair.clear();
    std::fill(hkeys.begin(), hkeys.end(), 0xffffffffu);
    std::fill(pre.begin(), pre.end(), 0);
    seedMap.clear();
    rcOf.clear();
    sequences.clear();
    refs.clear();
    arena.clear();
    for (Arena& a : chunkArenas) a.clear();
}

void SeedIndex::adopt(const std::vector<uint32_t>& seeds, const std::vector<int32_t>& rcTable) {
    seedMap = seeds;
    rcOf = rcTable;
    rcPair.clear();
    hashValid = false;
    sequences.clear();
    refs.clear();
    arena.clear();
    for (Arena& a : chunkArenas) a.clear();
}

void SeedIndex::rebuildHash() {
    std::vector<uint32_t> seeds;
    seeds.swap(seedMap);
    std::vector<int32_t> rc;
    rc.swap(rcOf);
    std::fill(hkeys.begin(), hkeys.end(), 0xffffffffu);
    std::fill(pre.begin(), pre.end(), 0);
    hashValid = true;
    for (uint32_t km : seeds) addSeedKmer(km);
    rcOf.swap(rc);
}

void SeedIndex::grow() {
    const size_t ncap = hkeys.size() * 2;
    hkeys.assign(ncap, 0xffffffffu);
    hvals.assign(ncap, -1);
    hmask = (uint32_t)ncap - 1;
    for (size_t i = 0; i < seedMap.size(); i++) {
        uint32_t h = hash(seedMap[i]) & hmask;
        while (hkeys[h] != 0xffffffffu) h = (h + 1) & hmask;
        hkeys[h] = seedMap[i];
        hvals[h] = (int32_t)i;
    }
}

int32_t SeedIndex::addSeedKmerId(uint32_t kmer) {
    if (!hashValid) rebuildHash();
    if (!rcOf.empty()) rcOf.clear();
    const uint32_t b = preHash(kmer);
    if ((pre[b >> 5] >> (b & 31)) & 1) {
        const int32_t id = find(kmer);
        if (id >= 0) return id;
    }
    if ((seedMap.size() + 1) * 2 > hkeys.size()) grow();
    uint32_t h = hash(kmer) & hmask;
    while (hkeys[h] != 0xffffffffu) h = (h + 1) & hmask;
    hkeys[h] = kmer;
    const int32_t id = (int32_t)seedMap.size();
    hvals[h] = id;
    seedMap.push_back(kmer);
    pre[b >> 5] |= 1u << (b & 31);
    return id;
}
void SeedIndex::addSeedKmer(uint32_t kmer) { addSeedKmerId(kmer); }

// ---- touchesSeed on a window's evaluated k-mers
namespace {
typedef bool (*TouchFn)(const SeedIndex&, const uint32_t*, uint32_t);

bool touchScalar(const SeedIndex& ix, const uint32_t* kmers, uint32_t n) {
    for (uint32_t i = 0; i < n; i++)
        if (kmers[i] != 0xffffffffu && ix.isSeed(kmers[i])) return true;
    return false;
}

__attribute__((target("avx2"))) bool touchAvx2(const SeedIndex& ix, const uint32_t* kmers, uint32_t n) {
    const int* pre = (const int*)ix.pre.data();
    const __m256i mul = _mm256_set1_epi32((int)2246822519u), one = _mm256_set1_epi32(1), m31 = _mm256_set1_epi32(31),
                  ones = _mm256_set1_epi32(-1);
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256i km = _mm256_loadu_si256((const __m256i*)(kmers + i));
        const __m256i h = _mm256_srl_epi32(_mm256_mullo_epi32(km, mul), _mm_cvtsi32_si128((int)ix.preShift));
        const __m256i w = _mm256_i32gather_epi32(pre, _mm256_srli_epi32(h, 5), 4);
        const __m256i bit = _mm256_sllv_epi32(one, _mm256_and_si256(h, m31));
        const __m256i hit = _mm256_andnot_si256(_mm256_cmpeq_epi32(km, ones), _mm256_cmpeq_epi32(_mm256_and_si256(w, bit), bit));
        unsigned m = (unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(hit));
        while (m) {
            const int l = __builtin_ctz(m);
            m &= m - 1;
            if (ix.find(kmers[i + l]) >= 0) return true;
        }
    }
    return touchScalar(ix, kmers + i, n - i);
}

__attribute__((target("avx512f"))) bool touchAvx512(const SeedIndex& ix, const uint32_t* kmers, uint32_t n) {
    const int* pre = (const int*)ix.pre.data();
    const __m512i mul = _mm512_set1_epi32((int)2246822519u), one = _mm512_set1_epi32(1), m31 = _mm512_set1_epi32(31),
                  ones = _mm512_set1_epi32(-1);
    uint32_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i km = _mm512_loadu_si512((const void*)(kmers + i));
        const __m512i h = _mm512_srl_epi32(_mm512_mullo_epi32(km, mul), _mm_cvtsi32_si128((int)ix.preShift));
        const __m512i w = _mm512_i32gather_epi32(_mm512_srli_epi32(h, 5), pre, 4);
        const __m512i bit = _mm512_sllv_epi32(one, _mm512_and_si512(h, m31));
        unsigned m = (unsigned)(_mm512_test_epi32_mask(w, bit) & _mm512_cmpneq_epi32_mask(km, ones));
        while (m) {
            const int l = __builtin_ctz(m);
            m &= m - 1;
            if (ix.find(kmers[i + l]) >= 0) return true;
        }
    }
    return touchScalar(ix, kmers + i, n - i);
}

TouchFn pickTouch() {
    if (const long want = dph_tune("touch_isa", -1); want >= 0) {  // tests: 0 scalar, 1 AVX2, 2 AVX-512 (only what the CPU has)
        if (want >= 2 && __builtin_cpu_supports("avx512f")) return touchAvx512;
        if (want >= 1 && __builtin_cpu_supports("avx2")) return touchAvx2;
        return touchScalar;
    }
    if (__builtin_cpu_supports("avx512f")) return touchAvx512;
    if (__builtin_cpu_supports("avx2")) return touchAvx2;
    return touchScalar;
}
}  // namespace

bool SeedIndex::touchesSeed(const uint32_t* kmers, uint32_t n) const {
    static const TouchFn fn = pickTouch();
    return fn(*this, kmers, n);
}

bool SeedIndex::touchesSeedWith(int isa, const uint32_t* kmers, uint32_t n) const {
    if (isa == 2) return __builtin_cpu_supports("avx512f") ? touchAvx512(*this, kmers, n) : false;
    if (isa == 1) return __builtin_cpu_supports("avx2") ? touchAvx2(*this, kmers, n) : false;
    return touchScalar(*this, kmers, n);
}

void SeedIndex::buildRcTable() {
    if (rcPair.size() == seedMap.size() && hashValid) {  // (all seeds came in through commitSeeds)
        bool complete = true;
        for (int32_t v : rcPair)
            if (v < 0) {
                complete = false;
                break;
            }
        if (complete) {
            rcOf = rcPair;
            return;
        }
    }
    rcOf.clear();
    std::vector<int32_t> t(seedMap.size());
    for (size_t i = 0; i < seedMap.size(); i++) t[i] = seedOfRcKmer((int32_t)i);
    rcOf.swap(t);
}

int32_t SeedIndex::seedOfRcKmer(int32_t seed) const {
    if ((size_t)seed < rcOf.size()) return rcOf[(size_t)seed];
    const int32_t id = find(reverseComplementKmer(seedMap[(size_t)seed], k));
    return id < 0 ? 0 : id;  // kmerMap[] of a non-seed is the zero value (seeds.go:17)
}

// AddSeeds seeds/seeds.go:62-156 on an ASCII window (firstLen == 4 views behave like plain strings).
// selectSeeds() is the selection loop; with checkIndex=false it assumes no evaluated k-mer is a seed yet (the
// speculative, thread-parallel form used by Overlapper::PrepareQueries), touchesSeed() tests exactly that assumption.
template <bool CHECK>
static void selectSeedsT(const SeedIndex& ix, const char* s, i64 L, int minSeeds, ValueView ranks, uint32_t* topN, const uint8_t* q) {
    const int k = ix.k;
    const uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * k)) - 1);
    double topVbuf[64];
    std::vector<double> topVdyn;
    double* topV = topVbuf;
    if (minSeeds > 64) {
        topVdyn.assign((size_t)minSeeds, 0.0);
        topV = topVdyn.data();
    }
    for (int i = 0; i < minSeeds; i++) {
        topN[i] = 0;
        topV[i] = 0.0;
    }
    auto kmerAt = [&](i64 p) {
        uint32_t v = 0;
        for (int j = 0; j < k; j++) v = (v << 2) | baseCode((unsigned char)s[p + j]);
        return v;
    };
    uint32_t kmer = kmerAt(0);
    i64 nextIndex = k;
    bool prefetched = false;  // the lookups of the block about to be walked were issued during the previous block
    while (nextIndex < L - k) {
        bool reset = false;
        double bestValue = 0.0;
        uint32_t bestSeed = 0;
        // the block's k value lookups hit a 4^k-entry table at random.  They are prefetched one block ahead (the next block is
        // known unless this one meets a seed): by the time a block is walked its lines have had a block's worth of time to arrive
        if (!prefetched) {
            uint32_t pk = kmer;
            for (i64 pi = nextIndex, i = 0; pi < L && i < k; i++, pi++) {
                pk = ((pk << 2) | baseCode((unsigned char)s[pi])) & mask;
                ranks.prefetch(pk);
            }
        }
        prefetched = false;
        if (nextIndex + 2 * (i64)k < L - k) {  // the block after this one exists (same test as the loop's) if nothing resets
            const i64 nb = nextIndex + 2 * (i64)k;  // its initial k-mer starts here
            uint32_t pk = kmerAt(nb);
            for (i64 pi = nb + k, i = 0; pi < L && i < k; i++, pi++) {
                pk = ((pk << 2) | baseCode((unsigned char)s[pi])) & mask;
                ranks.prefetch(pk);
            }
            prefetched = true;
        }
        for (int i = 0; nextIndex < L && i < k; i++) {
            kmer = ((kmer << 2) | baseCode((unsigned char)s[nextIndex])) & mask;
            nextIndex++;
            if (CHECK && ix.isSeed(kmer)) {
                reset = true;
                prefetched = false;  // (the walk goes on from a different offset than the one prefetched for)
                break;
            }
            double value = ranks.at(kmer);
            if (q) value *= (double)q[nextIndex - k / 2];  // seeds.go:99-101
            if (value > bestValue) {
                bestValue = value;
                bestSeed = kmer;
            }
        }
        if (!reset) {
            int n = 0;
            for (; n < minSeeds && topV[n] < bestValue; n++) {
                if (n > 0) {
                    topV[n - 1] = topV[n];
                    topN[n - 1] = topN[n];
                }
            }
            if (n > 0) {
                topV[n - 1] = bestValue;
                topN[n - 1] = bestSeed;
            }
        }
        nextIndex += k;
        if (nextIndex < L - k) kmer = kmerAt(nextIndex);
        nextIndex += k;
    }
}

void SeedIndex::selectSeeds(const char* s, i64 L, int minSeeds, ValueView ranks, uint32_t* topN, bool checkIndex,
                            const uint8_t* q) const {
    if (checkIndex) selectSeedsT<true>(*this, s, L, minSeeds, ranks, topN, q);
    else selectSeedsT<false>(*this, s, L, minSeeds, ranks, topN, q);
}

// true iff one of the k-mers AddSeeds would evaluate (with no reset so far) is already a seed
bool SeedIndex::touchesSeed(const char* s, i64 L) const {
    const uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * k)) - 1);
    auto kmerAt = [&](i64 p) {
        uint32_t v = 0;
        for (int j = 0; j < k; j++) v = (v << 2) | baseCode((unsigned char)s[p + j]);
        return v;
    };
    uint32_t kmer = kmerAt(0);
    i64 nextIndex = k;
    while (nextIndex < L - k) {
        for (int i = 0; nextIndex < L && i < k; i++) {
            kmer = ((kmer << 2) | baseCode((unsigned char)s[nextIndex])) & mask;
            nextIndex++;
            if (isSeed(kmer)) return true;
        }
        nextIndex += k;
        if (nextIndex < L - k) kmer = kmerAt(nextIndex);
        nextIndex += k;
    }
    return false;
}

void SeedIndex::commitSeeds(const uint32_t* topN, int n) {  // seeds.go:130-154
    // every seed enters together with its reverse complement: the seed -> reverse-complement-seed table falls out of the ids
    for (int i = 0; i < n; i++) {
        const int32_t a = addSeedKmerId(topN[i]);
        const int32_t b = addSeedKmerId(reverseComplementKmer(topN[i], k));
        if (rcPair.size() < seedMap.size()) rcPair.resize(seedMap.size(), -1);
        rcPair[(size_t)a] = b;
        rcPair[(size_t)b] = a;
    }
}

void SeedIndex::addSeeds(const char* s, i64 L, int minSeeds, ValueView ranks) {
    std::vector<uint32_t> topN((size_t)minSeeds, 0);
    selectSeeds(s, L, minSeeds, ranks, topN.data(), true);
    commitSeeds(topN.data(), minSeeds);
}

// ---------------------------------------------------------------------------------------------------------------
// SeedSequence operations (seeds/sequence.go)

i64 SeedSeq::seedOffset(int index, int k) const {
    index = index * 2 + 1;
    i64 o = seg[0];
    for (int i = 2; i < index; i += 2) o += seg[i] + k;
    return o;
}
i64 SeedSeq::seedOffsetFromEnd(int index, int k) const {
    index = index * 2 + 1;
    i64 o = seg[n - 1];
    for (int i = n - 3; i > index; i -= 2) o += seg[i] + k;
    return o;
}
int SeedSeq::maxSeed() const {
    int m = 0;
    for (int i = 1; i < n; i += 2) m = std::max(m, (int)seg[i]);
    return m;
}

SeedSeq* seqReverseComplement(Arena& a, SeedSeq* s, const SeedIndex& ix) {
    if (s->reverseComplement) return s->reverseComplement;
    int32_t* d = a.alloc((size_t)s->n);
    for (int i = 0; i < s->n; i++) d[s->n - 1 - i] = (i & 1) ? ix.seedOfRcKmer(s->seg[i]) : s->seg[i];
    SeedSeq* r = a.make();
    r->seg = d;
    r->n = s->n;
    r->id = s->id;
    r->length = s->length;
    r->offset = s->offset;
    r->inset = s->inset;
    r->reverseComplement = s;
    r->rc = !s->rc;
    r->parent = s->parent;
    return r;
}

SeedSeq* seqSubSequence(Arena& a, SeedSeq* s, int start, int end, i64 length, i64 offset, i64 inset) {
    SeedSeq* r = a.make();
    r->seg = s->seg + start * 2;
    r->n = end * 2 + 3 - start * 2;
    r->length = length;
    r->offset = offset;
    r->inset = inset;
    r->rc = s->rc;
    r->id = s->id;
    r->parent = s;
    return r;
}

// anchorStart / anchorEnd: seedOffset(startSeed) / seedOffsetFromEnd(endSeed) of the seeds passed in, when the caller knows
// them (>= 0); they follow the two walks below, so the whole-sequence sums are not needed
SeedSeq* seqTrimmed(Arena& a, SeedSeq* s, i64 startOffset, int startSeed, i64 endOffset, int endSeed, int k, i64 anchorStart,
                    i64 anchorEnd) {
    while (startSeed > 0 && startOffset >= s->seg[startSeed * 2] + k) {
        startOffset -= s->seg[startSeed * 2] + k;
        anchorStart -= s->seg[startSeed * 2] + k;
        startSeed--;
    }
    const int numSeeds = s->n / 2;
    while (endSeed < numSeeds - 1 && endOffset >= s->seg[endSeed * 2 + 2] + k) {
        endOffset -= s->seg[endSeed * 2 + 2] + k;
        anchorEnd -= s->seg[endSeed * 2 + 2] + k;
        endSeed++;
    }
    const bool known = anchorStart >= 0 && anchorEnd >= 0;
    const i64 offset = (known ? anchorStart : s->seedOffset(startSeed, k)) - startOffset;
    const i64 inset = (known ? anchorEnd : s->seedOffsetFromEnd(endSeed, k)) - endOffset;
    SeedSeq* t = s->rc ? seqSubSequence(a, s, startSeed, endSeed, s->length - offset - inset, s->offset + inset, s->inset + offset)
                       : seqSubSequence(a, s, startSeed, endSeed, s->length - offset - inset, s->offset + offset, s->inset + inset);
    int32_t* d = a.alloc((size_t)t->n);
    memcpy(d, t->seg, (size_t)t->n * 4);
    d[0] = (int32_t)startOffset;
    d[t->n - 1] = (int32_t)endOffset;
    t->seg = d;
    return t;
}

static inline bool wlContains(const std::vector<uint64_t>& w, int x) {
    size_t i = (size_t)x >> 6;
    return i < w.size() && ((w[i] >> (x & 63)) & 1);
}

SeedSeq* seqReduced(Arena& a, SeedSeq* s, const std::vector<uint64_t>& whitelist, int k, int minSeeds, std::vector<int>* index) {
    int count = 0, prev = -1;
    for (int i = 1; i < s->n; i += 2) {
        int next = s->seg[i];
        if (next != prev && wlContains(whitelist, next)) {
            count++;
            prev = next;
        }
    }
    if (count < minSeeds) return nullptr;
    int32_t* d = a.alloc((size_t)count * 2 + 1);
    i64 offset = s->seg[0];
    if (index) index->assign((size_t)count, 0);
    prev = -1;
    int j = 0;
    for (int i = 1; i < s->n; i += 2) {
        int seed = s->seg[i];
        if (prev != seed && wlContains(whitelist, seed)) {
            d[j] = (int32_t)offset;
            d[j + 1] = seed;
            if (index) (*index)[(size_t)(j / 2)] = i / 2;
            j += 2;
            offset = s->seg[i + 1];
            prev = seed;
        } else {
            offset += s->seg[i + 1] + k;
        }
    }
    d[j] = (int32_t)offset;
    SeedSeq* r = a.make();
    r->seg = d;
    r->n = count * 2 + 1;
    r->length = s->length;
    r->offset = s->offset;
    r->inset = s->inset;
    r->rc = s->rc;
    r->id = s->id;
    r->parent = s;
    return r;
}

void matchReverseComplement(Arena& a, SeedMatch& m, const SeedIndex& ix) {
    m.SeqA = seqReverseComplement(a, m.SeqA, ix);
    m.SeqB = seqReverseComplement(a, m.SeqB, ix);
    const int lengthA = m.SeqA->n / 2 - 1, lengthB = m.SeqB->n / 2 - 1;
    std::reverse(m.MatchA.begin(), m.MatchA.end());
    std::reverse(m.MatchB.begin(), m.MatchB.end());
    for (size_t i = 0; i < m.MatchA.size(); i++) {
        m.MatchA[i] = lengthA - m.MatchA[i];
        m.MatchB[i] = lengthB - m.MatchB[i];
    }
}

void matchBasesCovered(const SeedMatch& m, int k, i64* a, i64* b, bool* wouldPanic) {
    if (m.MatchA.empty()) {  // the reference indexes MatchA[0] and panics
        if (wouldPanic) *wouldPanic = true;
        *a = *b = 0;
        return;
    }
    i64 countA = (i64)m.MatchA.size() * k, countB = countA;
    int prevA = m.MatchA[0], prevB = m.MatchB[0];
    const int32_t* sa = m.SeqA->seg;
    const int32_t* sb = m.SeqB->seg;
    const int na = m.SeqA->n, nb = m.SeqB->n;
    for (size_t i = 1; i < m.MatchA.size(); i++) {
        const int s = m.MatchA[i], s2 = m.MatchB[i];
        if (s * 2 >= na || s2 * 2 >= nb || prevA * 2 + 2 >= na || prevB * 2 + 2 >= nb || prevA < 0 || prevB < 0) {
            if (wouldPanic) *wouldPanic = true;  // index out of range in the reference
            *a = *b = 0;
            return;
        }
        i64 d1 = sa[prevA * 2 + 2], d2 = sb[prevB * 2 + 2];
        for (int j = prevA + 2; j <= s; j++) d1 += sa[j * 2] + k;
        for (int j = prevB + 2; j <= s2; j++) d2 += sb[j * 2] + k;
        if (d1 < 0) countA += d1;
        if (d2 < 0) countB += d2;
        prevB = s2;
        prevA = s;
    }
    *a = countA;
    *b = countB;
}

void matchBaseIndex(const SeedMatch& m, int aIndex, int k, i64* indexOut, i64* basesOut, i64* distOut) {
    const int32_t* sa = m.SeqA->seg;
    const int32_t* sb = m.SeqB->seg;
    const int nb = m.SeqB->n;
    int before = 0;
    while (before < (int)m.MatchA.size() && m.MatchA[(size_t)before] <= aIndex) before++;
    if (before == 0) {
        i64 offset = 0;
        for (int i = m.MatchA[0]; i > aIndex; i--) offset += sa[i * 2] + k;
        int bIndex = m.MatchB[0];
        i64 distance = 0;
        for (int i = bIndex * 2; i > 0 && offset > 0; i -= 2) {
            offset -= sb[i] + k;
            distance += sb[i] + k;
            bIndex--;
        }
        *indexOut = bIndex == 0 ? 0 : bIndex;
        *basesOut = -offset;
        *distOut = bIndex == 0 ? distance + offset : distance;
        return;
    }
    before--;
    int bIndex = m.MatchB[(size_t)before];
    if (aIndex == m.MatchA[(size_t)before]) {
        *indexOut = bIndex;
        *basesOut = 0;
        *distOut = 0;
        return;
    }
    i64 offset = 0;
    for (int i = m.MatchA[(size_t)before] + 1; i <= aIndex; i++) offset += sa[i * 2] + k;
    i64 distance = 0;
    for (int i = bIndex * 2 + 2; i < nb && offset >= sb[i]; i += 2) {
        offset -= sb[i] + k;
        distance += sb[i] + k;
        bIndex++;
    }
    *indexOut = bIndex >= nb / 2 ? bIndex - 1 : bIndex;
    *basesOut = offset;
    *distOut = distance + offset;
}

void gapRange(i64 gap, int k, i64* mn, i64* mx) {
    i64 minGap = (gap * 2) / 3 - k, maxGap = (gap * 3) / 2 + k + 1;
    if (minGap < 0) {
        minGap = -k;
        if (maxGap < 0) maxGap = 0;
    } else if (maxGap < 20) {
        maxGap = 20;
        minGap = 0;
    }
    *mn = minGap;
    *mx = maxGap;
}

// ---------------------------------------------------------------------------------------------------------------
// util.GetSharedIDs(sets, 2, true) as multiAligner.Consensus uses it (alignment.go:45): seeds present in >= 2 of
// the sequences.  With minCount == 2 the 4-ladder's v2 is exact and order independent, and the early-return rule
// (bitset.go:338-342) fires when fewer than 2 sets reach a word — equivalent to "count >= 2" word by word.
static const std::vector<uint64_t>& seedsSharedByTwo(const std::vector<SeedSeq*>& seqs) {
    int maxSeed = 100;
    for (auto* s : seqs) maxSeed = std::max(maxSeed, s->maxSeed());
    const size_t W = (size_t)maxSeed / 64 + 1;
    // sparse form of the word-wise ladder: a seed is shared once a SECOND sequence shows it.  Per seed id one 32-bit
    // stamp (sequence number that touched it last, in this call) and a flag "already shared"; the stamps are never
    // cleared, a call counter makes old ones stale.
    static thread_local std::vector<uint64_t> v2;
    static thread_local std::vector<uint32_t> stamp;  // (call << 8 | sequence + 1) truncated: see below
    static thread_local uint32_t call = 0;
    v2.assign(W, 0);
    if (stamp.size() < W * 64) stamp.resize(W * 64, 0);
    const size_t ns = seqs.size();
    if (ns >= 0xffff || ++call >= 0xffff) {  // stamps hold 16 bits of call and 16 bits of sequence number
        std::fill(stamp.begin(), stamp.end(), 0u);
        call = 1;
    }
    if (ns >= 0xffff) {  // (never in practice) fall back to the three-bitset form
        static thread_local std::vector<uint64_t> v1, row;
        v1.assign(W, 0);
        row.assign(W, 0);
        for (auto* s : seqs) {
            for (int j = 1; j < s->n; j += 2) {
                const uint32_t sd = (uint32_t)s->seg[j];
                const uint64_t bit = 1ull << (sd & 63);
                if (row[sd >> 6] & bit) continue;
                row[sd >> 6] |= bit;
                if (v1[sd >> 6] & bit) v2[sd >> 6] |= bit;
                else v1[sd >> 6] |= bit;
            }
            for (int j = 1; j < s->n; j += 2) row[(uint32_t)s->seg[j] >> 6] = 0;
        }
        return v2;
    }
    const uint32_t base = call << 16;
    uint32_t q = 0;
    for (auto* s : seqs) {
        q++;
        const uint32_t mine = base | q;
        for (int j = 1; j < s->n; j += 2) {
            const uint32_t sd = (uint32_t)s->seg[j];
            const uint32_t st = stamp[sd];
            if (st == mine) continue;                 // seen in this sequence already
            if ((st >> 16) == call) v2[sd >> 6] |= 1ull << (sd & 63);  // an earlier sequence of this call showed it
            stamp[sd] = mine;
        }
    }
    return v2;
}

// One support-scan pass of multiAligner.Consensus over all other sequences j for a proposing sequence (alignment.go:
// 101-131), restricted to what the first candidate seed of j decides: found there (count/sum), nothing in the window,
// or `slow` (the reference's walk has to advance: done by the caller with the scalar code).  32-bit, branch free.
// The proposing sequence itself is NOT excluded here (the caller subtracts it), so one pass serves every proposer with
// the same (d, seed, window) — in a clean pile-up that is all of them.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx2", "default")))
#endif
static void pairScan(int ns, const int32_t* __restrict__ ok, const int32_t* __restrict__ od, const int32_t* __restrict__ sd,
                     const int32_t* __restrict__ gp, int32_t d, int32_t minD, int32_t maxD, int32_t nextSeed, int k,
                     uint8_t* __restrict__ slow, uint8_t* __restrict__ fnd, int* cntOut, i64* sumOut) {
    int cnt = 0;
    int32_t sum = 0;
    for (int j = 0; j < ns; j++) {
        const int32_t t = d + gp[j];
        const int32_t m0 = (t * 2) / 3 - k, x0 = (t * 3) / 2 + k + 1;  // gapRange :411-424
        const bool neg = m0 < 0, small = !neg && x0 < 20;
        int32_t mx = neg ? (x0 < 0 ? 0 : x0) : (small ? 20 : x0);
        int32_t mn = neg ? -k : (small ? 0 : m0);
        mn = mn > minD ? minD : mn;
        mx = mx < maxD ? maxD : mx;
        const int32_t o = od[j];
        const bool valid = ok[j] != 0;
        const bool inWin = o >= mn && o < mx;
        const bool found = valid && inWin && sd[j] == nextSeed;
        const bool walk = valid && !found && o < mx;  // below the window, or inside it on another seed
        cnt += found ? 1 : 0;
        sum += found ? o : 0;
        slow[j] = walk ? 1 : 0;
        fnd[j] = found ? 1 : 0;
    }
    *cntOut = cnt;
    *sumOut = sum;
}

// The same pass when every other sequence has gaps[j] == 0 (all in step, the common state): the window is one pair of
// constants and the loop is compares only.
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx2", "default")))
#endif
static void pairScanInStep(int ns, const int32_t* __restrict__ ok, const int32_t* __restrict__ od, const int32_t* __restrict__ sd,
                           int32_t mn, int32_t mx, int32_t nextSeed, uint8_t* __restrict__ slow, uint8_t* __restrict__ fnd,
                           int* cntOut, i64* sumOut) {
    int cnt = 0;
    int32_t sum = 0;
    for (int j = 0; j < ns; j++) {
        const int32_t o = od[j];
        const bool valid = ok[j] != 0;
        const bool found = valid && o >= mn && o < mx && sd[j] == nextSeed;
        const bool walk = valid && !found && o < mx;
        cnt += found ? 1 : 0;
        sum += found ? o : 0;
        slow[j] = walk ? 1 : 0;
        fnd[j] = found ? 1 : 0;
    }
    *cntOut = cnt;
    *sumOut = sum;
}

// seeds/alignment.go:23-268
// matchesOut receives pointers into a per-thread pool of SeedMatch objects (their vectors keep their capacity between
// calls); they stay valid until the calling thread's next call.
// Reduced() of every sequence to the seeds shared by >= 2 of them (alignment.go:45-50)
static void reduceForConsensus(Arena& arena, std::vector<SeedSeq*>& seqs, int k, std::vector<SeedSeq*>& red,
                               std::vector<std::vector<int>>& seedMap) {
    const size_t ns = seqs.size();
    const std::vector<uint64_t>* shared;
    {
        FINE(2);
        shared = &seedsSharedByTwo(seqs);
    }
    const std::vector<uint64_t>& useSeeds = *shared;
    FINE(3);
    if (seedMap.size() < ns) seedMap.resize(ns);
    red.assign(ns, nullptr);
    for (size_t i = 0; i < ns; i++) red[i] = seqReduced(arena, seqs[i], useSeeds, k, 1, &seedMap[i]);
}

static SeedSeq* multiAlignerCore(Arena& arena, std::vector<SeedSeq*>& seqs, std::vector<SeedSeq*>& red,
                                 std::vector<std::vector<int>>& seedMap, int k, std::vector<SeedMatch*>& matchesOut);

SeedSeq* multiAlignerConsensus(Arena& arena, std::vector<SeedSeq*>& seqs, int k, std::vector<SeedMatch*>& matchesOut) {
    // scratch that keeps its capacity between calls (one set per worker thread)
    static thread_local std::vector<std::vector<int>> seedMap;
    static thread_local std::vector<SeedSeq*> red;
    reduceForConsensus(arena, seqs, k, red, seedMap);
    return multiAlignerCore(arena, seqs, red, seedMap, k, matchesOut);
}

// per-thread pool of SeedMatch objects for the consensus matches (their vectors keep their capacity)
static std::vector<SeedMatch>& consensusMatchPool(size_t ns) {
    static thread_local std::vector<SeedMatch> matchPool;
    if (matchPool.size() < ns) matchPool.resize(ns);
    return matchPool;
}

static SeedSeq* multiAlignerCore(Arena& arena, std::vector<SeedSeq*>& seqs, std::vector<SeedSeq*>& red,
                                 std::vector<std::vector<int>>& seedMap, int k, std::vector<SeedMatch*>& matchesOut) {
    FINE(4);
    const size_t ns = seqs.size();
    static thread_local std::vector<i64> pos, offs, gaps, supported, dist;
    static thread_local std::vector<int32_t> consensus, okv, odv, sdv, gpv;
    static thread_local std::vector<uint8_t> slowv, fndv;
    auto S = [&](size_t i) -> const int32_t* { return red[i] ? red[i]->seg : nullptr; };
    auto N = [&](size_t i) -> i64 { return red[i] ? red[i]->n : 0; };
    pos.assign(ns, -1);
    offs.assign(ns, 0);
    gaps.assign(ns, 50);
    supported.assign(ns, 0);
    dist.assign(ns, 0);
    consensus.clear();
    std::vector<SeedMatch>& matchPool = consensusMatchPool(ns);
    static thread_local std::vector<SeedMatch*> matches;
    matches.assign(ns, nullptr);
    for (size_t i = 0; i < ns; i++)
        if (red[i]) {
            SeedMatch* m = &matchPool[i];
            m->MatchA.clear();
            m->MatchB.clear();
            m->SeqA = nullptr;
            m->SeqB = seqs[i];
            m->QueryID = 0;
            m->ReverseComplementQuery = false;
            matches[i] = m;
        }
    bool finished = false;
    static const bool uniformFast = true;
    const i64 kNarrow = (i64)1 << 28;  // all quantities below this: 32-bit arithmetic is exact
    okv.assign(ns, 0);
    odv.assign(ns, 0);
    sdv.assign(ns, 0);
    gpv.assign(ns, 0);
    slowv.assign(ns, 0);
    fndv.assign(ns, 0);
    while (!finished) {
        i64 fCount = 0, near = 100000;
        bool memoOk = false, memoSlow = false;  // last support scan of this step: (d, seed, window) -> totals
        i64 memoD = 0, memoMin = 0, memoMax = 0, memoSum = 0;
        int32_t memoSeed = 0;
        int memoCnt = 0;
        bool narrow = true, inStep = true, uniform = true;
        i64 nValid = 0, o0 = 0;
        int32_t sd0 = -1;
        for (size_t j = 0; j < ns; j++) {  // state of every sequence's next seed (constant during the support scan)
            const int32_t* s2 = S(j);
            const i64 p2 = pos[j] + 1;
            const bool ok = s2 && p2 < N(j) / 2;
            const i64 o = ok ? (i64)s2[p2 * 2] - offs[j] : 0;
            if (o >= kNarrow || o <= -kNarrow || gaps[j] >= kNarrow || gaps[j] <= -kNarrow) narrow = false;
            if (ok && gaps[j] != 0) inStep = false;
            okv[j] = ok ? 1 : 0;
            odv[j] = (int32_t)o;
            sdv[j] = ok ? s2[p2 * 2 + 1] : -1;
            gpv[j] = (int32_t)gaps[j];
            if (ok) {
                if (nValid == 0) {
                    o0 = o;
                    sd0 = s2[p2 * 2 + 1];
                } else if (o != o0 || s2[p2 * 2 + 1] != sd0) {
                    uniform = false;
                }
                nValid++;
            }
        }
#ifdef DPH_FINE
        g_fine.cyc[8] += 1;                                                 // steps
        if (uniform && inStep && nValid >= 2) g_fine.cyc[9] += 1;            // ... taken by the fast path
        if (!inStep) g_fine.cyc[10] += 1;                                   // ... with a sequence out of step
        else if (!uniform) g_fine.cyc[11] += 1;                             // ... in step but disagreeing
        else if (nValid < 2) g_fine.cyc[12] += 1;                           // ... fewer than 2 live sequences
#endif
        if (uniformFast && uniform && inStep && narrow && nValid >= 2 && o0 > -k && o0 < 100000) {
            // Every sequence that still has a seed is in step (gap 0) and shows the same seed at the same distance.  Then
            // the general code below does nothing but agree: each of them proposes (d = o0 < near, which only drops to
            // maxD(o0) > o0), finds the seed in every other one at once (o0 lies inside gapRange(o0) for every -k < o0),
            // so supported = nValid >= 2 and dist/supported = o0 exactly; the first proposer wins the selection, and the
            // update loop finds the seed at pos+1 of every sequence.  Same state, same consensus, same matches.
            consensus.push_back((int32_t)o0);
            consensus.push_back(sd0);
            const int32_t ci = (int32_t)(consensus.size() / 2 - 1);
            for (size_t i = 0; i < ns; i++)
                if (okv[i]) {
                    const i64 md = pos[i] + 1;
                    pos[i] = md;
                    offs[i] = 0;
                    dist[i] = nValid * o0;  // what the support scan leaves behind (a later step may read it once
                                            // the sequence has ended: the reference's distances persist, :170-176)
                    matches[i]->MatchA.push_back(ci);
                    matches[i]->MatchB.push_back(seedMap[i][(size_t)md]);
                }
            continue;  // finished would be false: nValid >= 2 sequences go on
        }
        for (size_t i = 0; i < ns; i++) {
            const int32_t* segment = S(i);
            const i64 sl = N(i), p = pos[i];
            supported[i] = 0;
            if (!segment || p >= (sl - 1) / 2 - 1) {
                fCount++;
                continue;
            }
            const i64 d = segment[p * 2 + 2] - offs[i];
            dist[i] = d;
            if (d < near && d > -k) {
                const int32_t nextSeed = segment[p * 2 + 3];
                i64 minD, maxD;
                gapRange(d + gaps[i], k, &minD, &maxD);
                minD -= gaps[i];
                maxD -= gaps[i];
                if (near > maxD) near = maxD;
                supported[i] = 1;
                auto scanOne = [&](size_t j) {  // alignment.go:101-131 for one other sequence
                    const int32_t* s2 = S(j);
                    const i64 sl2 = N(j);
                    if (!s2 || j == i) return;
                    i64 p2 = pos[j] + 1;
                    if (p2 < sl2 / 2) {
                        i64 min2, max2;
                        gapRange(d + gaps[j], k, &min2, &max2);
                        if (min2 > minD) min2 = minD;
                        if (max2 < maxD) max2 = maxD;
                        i64 otherD = s2[p2 * 2] - offs[j];
                        while (otherD < min2 && p2 < sl2 / 2) {
                            p2++;
                            otherD += s2[p2 * 2] + k;
                        }
                        while (otherD < max2 && p2 < sl2 / 2) {
                            if (s2[p2 * 2 + 1] == nextSeed) {
                                supported[i]++;
                                dist[i] += otherD;
                                break;
                            }
                            p2++;
                            otherD += s2[p2 * 2] + k;
                        }
                    }
                };
                if (narrow && d < kNarrow && d > -kNarrow && minD > -kNarrow && maxD < kNarrow) {
                    // the other sequences' next seeds were snapshotted for this step: one branch-free (vectorised)
                    // pass settles every pair whose answer is decided by that first seed; the rest take scanOne
                    if (!(memoOk && memoD == d && memoSeed == nextSeed && memoMin == minD && memoMax == maxD)) {
                        if (inStep) {
                            i64 mn0, mx0;
                            gapRange(d, k, &mn0, &mx0);
                            if (mn0 > minD) mn0 = minD;
                            if (mx0 < maxD) mx0 = maxD;
                            pairScanInStep((int)ns, okv.data(), odv.data(), sdv.data(), (int32_t)mn0, (int32_t)mx0, nextSeed,
                                           slowv.data(), fndv.data(), &memoCnt, &memoSum);
                        } else {
                            pairScan((int)ns, okv.data(), odv.data(), sdv.data(), gpv.data(), (int32_t)d, (int32_t)minD,
                                     (int32_t)maxD, nextSeed, k, slowv.data(), fndv.data(), &memoCnt, &memoSum);
                        }
                        memoOk = true;
                        memoD = d;
                        memoSeed = nextSeed;
                        memoMin = minD;
                        memoMax = maxD;
                        memoSlow = false;
                        for (size_t j = 0; j < ns; j++) memoSlow |= slowv[j] != 0;
                    }
                    supported[i] += memoCnt - fndv[i];
                    dist[i] += memoSum - (fndv[i] ? odv[i] : 0);
                    if (memoSlow)
                        for (size_t j = 0; j < ns; j++)
                            if (slowv[j] && j != i) scanOne(j);
                } else {
                    for (size_t j = 0; j < ns; j++) scanOne(j);
                }
            }
        }
        if (fCount >= (i64)ns) break;
        i64 minseed = -1, mindist = 0, minsup = 0, minD = 0, maxD = 0;
        for (size_t i = 0; i < ns; i++) {
            if (supported[i] > 1) {
                const i64 d = dist[i] / supported[i];
                const i64 seed = S(i)[pos[i] * 2 + 3];
                if (minseed == -1 || (minseed == seed && supported[i] > minsup) || (minseed != seed && mindist > d)) {
                    minsup = supported[i];
                    mindist = d;
                    minseed = seed;
                    gapRange(d + gaps[i], k, &minD, &maxD);
                    minD -= gaps[i];
                    maxD -= gaps[i];
                }
            }
        }
        if (minseed == -1) {
            i64 minIndex = -1, minDist = 100000;
            for (size_t i = 0; i < ns; i++) {
                i64 d = dist[i];
                if (supported[i] > 1) d = d / supported[i];
                if (S(i) && pos[i] < (i64)ns / 2 && d < minDist) {  // len(segments)/2 is len(seqs)/2 (:170)
                    minDist = d;
                    minIndex = (i64)i;
                }
            }
            if (minIndex == -1) break;
            for (size_t i = 0; i < ns; i++)
                if (S(i)) {
                    gaps[i] += minDist;
                    offs[i] += minDist;
                }
            gaps[(size_t)minIndex] = 0;
            offs[(size_t)minIndex] = 0;
            pos[(size_t)minIndex]++;
            continue;
        }
        consensus.push_back((int32_t)mindist);
        consensus.push_back((int32_t)minseed);
        fCount = 0;
        bool cWinOk = false;
        i64 cMin = 0, cMax = 0;
        for (size_t i = 0; i < ns; i++) {
            const int32_t* segment = S(i);
            const i64 sl = N(i);
            if (!segment) {
                fCount++;
                continue;
            }
            i64 matchDex = pos[i] + 1;
            if (matchDex < sl / 2) {
                i64 min2, max2;
                if (gaps[i] == 0) {  // in step: one window for all such sequences
                    if (!cWinOk) {
                        gapRange(mindist, k, &cMin, &cMax);
                        if (cMin > minD) cMin = minD;
                        if (cMax < maxD) cMax = maxD;
                        cWinOk = true;
                    }
                    min2 = cMin;
                    max2 = cMax;
                } else {
                    gapRange(mindist + gaps[i], k, &min2, &max2);
                    if (min2 > minD) min2 = minD;
                    if (max2 < maxD) max2 = maxD;
                }
                i64 otherD = segment[matchDex * 2] - offs[i];
                while (otherD < min2 && matchDex < sl / 2) {
                    matchDex++;
                    otherD += segment[matchDex * 2] + k;
                }
                bool found = false;
                while (otherD < max2 && matchDex < sl / 2) {
                    if (segment[matchDex * 2 + 1] == minseed) {
                        pos[i] = matchDex;
                        offs[i] = 0;
                        gaps[i] = 0;
                        matches[i]->MatchA.push_back((int32_t)(consensus.size() / 2 - 1));
                        matches[i]->MatchB.push_back(seedMap[i][(size_t)matchDex]);
                        found = true;
                        break;
                    }
                    matchDex++;
                    otherD += segment[matchDex * 2] + k;
                }
                if (!found) {
                    gaps[i] += mindist;
                    offs[i] += mindist;
                    i64 p = pos[i];
                    while (p < sl / 2 && offs[i] > segment[p * 2 + 2] + 50) {
                        offs[i] -= segment[p * 2 + 2] + k;
                        p++;
                        pos[i]++;
                    }
                    if (p >= sl / 2) fCount++;
                }
            } else {
                fCount++;
            }
        }
        finished = fCount >= (i64)ns;
    }
    consensus.push_back(0);
    SeedSeq* cons = arena.make();
    int32_t* d = arena.alloc(consensus.size());
    memcpy(d, consensus.data(), consensus.size() * 4);
    cons->seg = d;
    cons->n = (int)consensus.size();
    cons->length = -k;
    for (size_t i = 0; i < consensus.size(); i += 2) cons->length += consensus[i] + k;  // LoadSequence :35-42
    for (i64 i = (i64)matches.size() - 1; i >= 0; i--) {
        SeedMatch* m = matches[(size_t)i];
        if (!m || m->MatchA.size() < 3) {
            matches[(size_t)i] = matches.back();
            matches.pop_back();
        } else {
            m->SeqA = cons;
        }
    }
    matchesOut.assign(matches.begin(), matches.end());
    return cons;
}

// ---------------------------------------------------------------------------------------------------------------
// overlap/combine.go

// step 1 of trimToBestSeed (:24-58): the best front and back seeds of the consensus (also what tests/test_hand_known_answers.py holds to
// answers worked from the Go text: dph_hand_trim_indices)
void trimBestIndices(int upto, const std::vector<SeedMatch*>& ms, int minMatch, int length, int* bestOut, int* backOut) {
    int bestCount = 0, bestScore = 0, bestIndex = upto, backCount = 0, backScore = 0;
    int backIndex = length - upto - 1;
    // count[i] = matches whose MatchA holds consensus seed i; bCount[i] = ... holds seed length-1-i at a position j >= 1
    // (the reference's backward walk stops before j = 0, :40-46).  MatchA is strictly ascending, so one pass over the
    // first / last `upto` consensus seeds of every match gives all counts the per-i walks of :30-47 produce.
    static thread_local std::vector<int> frontCount, backCountV;
    frontCount.assign((size_t)std::max(upto, 0), 0);
    backCountV.assign((size_t)std::max(upto, 0), 0);
    bool ascending = true;
    for (SeedMatch* match : ms) {
        const std::vector<int32_t>& A = match->MatchA;
        for (size_t j = 1; j < A.size() && ascending; j++) ascending = A[j] > A[j - 1];
    }
    if (ascending) {
        for (SeedMatch* match : ms) {
            const std::vector<int32_t>& A = match->MatchA;
            for (size_t j = 0; j < A.size() && A[j] < upto; j++)
                if (A[j] >= 0) frontCount[(size_t)A[j]]++;
            for (size_t j = A.size(); j-- > 1;) {
                const int back = length - 1 - A[j];
                if (back >= upto) break;
                if (back >= 0) backCountV[(size_t)back]++;
            }
        }
    }
    for (int i = 0; i < upto; i++) {
        int count = 0, bCount = 0;
        if (ascending) {
            count = frontCount[(size_t)i];
            bCount = backCountV[(size_t)i];
        } else {  // (never seen: the literal walks)
            for (SeedMatch* match : ms) {
                for (int index : match->MatchA) {
                    if (index == i) count++;
                    if (index >= i) break;
                }
                for (int j = (int)match->MatchA.size() - 1; j > 0; j--) {
                    const int index = match->MatchA[(size_t)j];
                    if (index == length - 1 - i) bCount++;
                    if (index <= length - 1 - i) break;
                }
            }
        }
        if (count - i >= bestScore || (bestCount < minMatch && count >= minMatch)) {
            bestCount = count;
            bestScore = count - i;
            bestIndex = i;
        }
        if (bCount - i >= backScore || (backCount < minMatch && bCount >= minMatch)) {
            backCount = bCount;
            backScore = bCount - i;
            backIndex = length - 1 - i;
        }
    }
    *bestOut = bestIndex;
    *backOut = backIndex;
}

static void trimToBestSeed(Arena& ar, int upto, std::vector<SeedMatch*>& ms, int minMatch, int k, std::vector<SeedSeq*>& parts,
                           std::vector<uint8_t>& cantTrim, i64* badBack) {  // :21-111
    parts.assign(ms.size(), nullptr);
    cantTrim.assign(ms.size(), 0);
    int bestIndex = 0, backIndex = 0;
    trimBestIndices(upto, ms, minMatch, ms[0]->SeqA->numSeeds(), &bestIndex, &backIndex);
    SeedSeq* consensus = seqTrimmed(ar, ms[0]->SeqA, 0, bestIndex, 0, backIndex, k);
    for (size_t j = 0; j < ms.size(); j++) {
        SeedMatch* match = ms[j];
        i64 index, bases, frontDistance, bIndex, backBases, backDistance;
        matchBaseIndex(*match, bestIndex, k, &index, &bases, &frontDistance);
        matchBaseIndex(*match, backIndex, k, &bIndex, &backBases, &backDistance);
        cantTrim[j] = frontDistance > 50 || frontDistance < -50 || backDistance > 50 || backDistance < -50;
        if (bases > -k && index < match->SeqB->numSeeds() - 1) {
            bases = match->SeqB->nextSeedOffset((int)index, k) - bases;
            index++;
        } else if (bases < 0) {
            bases = -bases + k;
        }
        parts[j] = seqTrimmed(ar, match->SeqB, bases, (int)index, backBases, (int)bIndex, k);
        match->SeqB = parts[j];
        match->SeqA = consensus;
        int front = 0;
        while (front < (int)match->MatchB.size() && match->MatchB[(size_t)front] < index) front++;
        int back = (int)match->MatchB.size() - 1;
        while (back >= 0 && match->MatchB[(size_t)back] > bIndex) back--;
        if (front < 0 || back + 1 > (int)match->MatchA.size() || back < front) {
            // "Bad back:" diagnostic (:93-102) is suppressed (it prints Go pointers); the match becomes empty.
            if (badBack) (*badBack)++;
            if (back + 1 < front) back = front - 1;
        }
        // keep [front, back] in place (no reallocation: this runs for every part of every query)
        match->MatchA.erase(match->MatchA.begin() + back + 1, match->MatchA.end());
        match->MatchA.erase(match->MatchA.begin(), match->MatchA.begin() + front);
        match->MatchB.erase(match->MatchB.begin() + back + 1, match->MatchB.end());
        match->MatchB.erase(match->MatchB.begin(), match->MatchB.begin() + front);
        for (size_t n = 0; n < match->MatchB.size(); n++) {
            match->MatchA[n] -= bestIndex;
            match->MatchB[n] -= (int32_t)index;
        }
    }
}

static SeedContig* newSeedContig(Arena& ar, std::vector<SeedMatch*>& ms, int k, i64* badBack) {  // :113-133
    const int minMatch = ms.size() < 5 ? (int)ms.size() : 5;
    static thread_local std::vector<SeedSeq*> parts;
    static thread_local std::vector<uint8_t> trimFailed;
    static thread_local SeedContig contig;  // one contig per worker thread at a time
    {
        FINE(5);
        trimToBestSeed(ar, ms[0]->SeqA->numSeeds() / 4, ms, minMatch, k, parts, trimFailed, badBack);
    }
    FINE(6);
    SeedContig* c = &contig;
    const size_t n = ms.size();
    c->Parts.assign(n, 0);
    c->ReverseComplement.assign(n, 0);
    c->Offsets.assign(n, 0);
    c->Lengths.assign(n, 0);
    c->SeqLengths.assign(n, 0);
    c->Approximate = trimFailed;
    c->Matches = ms;
    for (size_t i = 0; i < n; i++) {
        SeedSeq* part = parts[i];
        c->Parts[i] = part->id;
        c->ReverseComplement[i] = part->rc;
        SeedSeq* parent = part;
        while (parent->parent) parent = parent->parent;
        c->SeqLengths[i] = parent->length;
        c->Offsets[i] = part->offset;
        c->Lengths[i] = parent->length - part->offset - part->inset;
    }
    return c;
}

// BuildConsensus :163-182: un-RC the rc-query matches, drop matches covering < 25 bases, trim each target to the
// query-aligned span.  Fills `seqs`.
// What matchReverseComplement + Trimmed() leave behind for a match of the reverse-complemented query, without spelling the
// whole reverse complement of the target out: a target is a chunk of up to chunk_size bases (~770 seeds at k=13) of which
// a match touches ~75, and the full copy with its seed-id translation was a sixth of the consensus stage.  `S` is the
// forward target, the match's index lists are already reversed.  R = rc(S) is created with only the slice
// [2*startSeed, 2*endSeed+2] of its segments filled in (Trimmed()'s final range, which contains every matched seed);
// offsets come from the forward sequence: R.seedOffset(i) = S.seedOffsetFromEnd(ns-1-i) and vice versa.
// anchorStart / anchorEnd (>= 0 when known): R.seedOffset(startSeed) / R.seedOffsetFromEnd(endSeed) of the seeds passed in,
// i.e. the forward target's GetSeedOffsetFromEnd(last matched) / GetSeedOffset(first matched) from the device.
static SeedSeq* trimmedRcTarget(Arena& a, SeedSeq* S, const SeedIndex& ix, i64 startOffset, int startSeed, i64 endOffset,
                                int endSeed, int k, int fillLo, int fillHi, i64 anchorStart, i64 anchorEnd, SeedSeq** rcOut) {
    const int n = S->n, numSeeds = n / 2;
    auto gapR = [&](int t) -> i64 { return S->seg[2 * (numSeeds - t)]; };  // R.seg[2t]
    while (startSeed > 0 && startOffset >= gapR(startSeed) + k) {
        startOffset -= gapR(startSeed) + k;
        anchorStart -= gapR(startSeed) + k;
        startSeed--;
    }
    while (endSeed < numSeeds - 1 && endOffset >= gapR(endSeed + 1) + k) {
        endOffset -= gapR(endSeed + 1) + k;
        anchorEnd -= gapR(endSeed + 1) + k;
        endSeed++;
    }
    const bool known = anchorStart >= 0 && anchorEnd >= 0;
    const i64 offset = (known ? anchorStart : S->seedOffsetFromEnd(numSeeds - 1 - startSeed, k)) - startOffset;  // R.seedOffset(startSeed)
    const i64 inset = (known ? anchorEnd : S->seedOffset(numSeeds - 1 - endSeed, k)) - endOffset;                // R.seedOffsetFromEnd(endSeed)
    SeedSeq* R = a.make();
    int32_t* d = a.alloc((size_t)n);
    const int lo = std::max(0, std::min(2 * startSeed, 2 * fillLo)), hi = std::min(n - 1, std::max(2 * endSeed + 2, 2 * fillHi + 2));
    for (int j = lo; j <= hi; j++) d[j] = (j & 1) ? ix.seedOfRcKmer(S->seg[n - 1 - j]) : S->seg[n - 1 - j];
    R->seg = d;
    R->n = n;
    R->id = S->id;
    R->length = S->length;
    R->offset = S->offset;
    R->inset = S->inset;
    R->reverseComplement = S;
    R->rc = !S->rc;
    R->parent = S->parent;
    *rcOut = R;
    SeedSeq* t = R->rc ? seqSubSequence(a, R, startSeed, endSeed, R->length - offset - inset, R->offset + inset, R->inset + offset)
                       : seqSubSequence(a, R, startSeed, endSeed, R->length - offset - inset, R->offset + offset, R->inset + inset);
    int32_t* c = a.alloc((size_t)t->n);
    memcpy(c, t->seg, (size_t)t->n * 4);
    c[0] = (int32_t)startOffset;
    c[t->n - 1] = (int32_t)endOffset;
    t->seg = c;
    return t;
}

static void consensusTrimTargets(Arena& ar, const SeedIndex& sg, std::vector<SeedMatch*>& overlaps, std::vector<SeedSeq*>& seqs) {
    const int k = sg.k;
    seqs.clear();
    static const bool fullRc = false;
    static const bool useAnchors = !fullRc;
    {
        FINE(0);
        for (SeedMatch* lap : overlaps) {
            if (!lap->ReverseComplementQuery) continue;
            if (fullRc || lap->MatchA.empty()) {
                matchReverseComplement(ar, *lap, sg);
                lap->ReverseComplementQuery = false;  // (done: the second pass takes the forward path)
                continue;
            }
            // the index lists and the query now; the target in the second pass, once the query of overlaps[0] is forward
            lap->SeqA = seqReverseComplement(ar, lap->SeqA, sg);
            const int lengthA = lap->SeqA->n / 2 - 1, lengthB = lap->SeqB->n / 2 - 1;
            std::reverse(lap->MatchA.begin(), lap->MatchA.end());
            std::reverse(lap->MatchB.begin(), lap->MatchB.end());
            for (size_t i = 0; i < lap->MatchA.size(); i++) {
                lap->MatchA[i] = lengthA - lap->MatchA[i];
                lap->MatchB[i] = lengthB - lap->MatchB[i];
            }
        }
    }
    FINE(1);
    for (SeedMatch* lap : overlaps) {
        if (lap->ReverseComplementQuery) {
            int fillLo = lap->MatchB[0], fillHi = lap->MatchB[0];
            for (int32_t b : lap->MatchB) {
                fillLo = std::min(fillLo, (int)b);
                fillHi = std::max(fillHi, (int)b);
            }
            SeedSeq* R = nullptr;
            SeedSeq* t = trimmedRcTarget(ar, lap->SeqB, sg, overlaps[0]->SeqA->seedOffset(lap->MatchA[0], k), lap->MatchB[0],
                                         overlaps[0]->SeqA->seedOffsetFromEnd(lap->MatchA.back(), k), lap->MatchB.back(), k, fillLo,
                                         fillHi, useAnchors ? lap->anchorLastFromEndB : -1, useAnchors ? lap->anchorFirstB : -1, &R);
            lap->SeqB = R;
            i64 ca, cb;
            matchBasesCovered(*lap, k, &ca, &cb, nullptr);
            if (ca < 25 || cb < 25) continue;
            seqs.push_back(t);
            continue;
        }
        i64 ca, cb;
        matchBasesCovered(*lap, k, &ca, &cb, nullptr);
        if (ca < 25 || cb < 25) continue;
        seqs.push_back(seqTrimmed(ar, lap->SeqB, overlaps[0]->SeqA->seedOffset(lap->MatchA[0], k), lap->MatchB[0],
                                  overlaps[0]->SeqA->seedOffsetFromEnd(lap->MatchA.back(), k), lap->MatchB.back(), k,
                                  useAnchors ? lap->anchorFirstB : -1, useAnchors ? lap->anchorLastFromEndB : -1));
    }
}

SeedContig* buildConsensus(Arena& ar, const SeedIndex& sg, std::vector<SeedMatch*>& overlaps, i64* badBack) {  // :163-193
    const int k = sg.k;
    static thread_local std::vector<SeedSeq*> seqs;
    static thread_local std::vector<SeedMatch*> overlap;
    consensusTrimTargets(ar, sg, overlaps, seqs);
    if (seqs.size() > 1) {
        multiAlignerConsensus(ar, seqs, k, overlap);
        if (overlap.size() > 1) return newSeedContig(ar, overlap, k, badBack);
    }
    return nullptr;
}

void consensusPrepare(ConsJob& job, const SeedIndex& sg, std::vector<SeedMatch*>& overlaps) {
    job.arena.clear();
    consensusTrimTargets(job.arena, sg, overlaps, job.seqs);
    job.aligned = job.seqs.size() > 1;
    if (job.aligned) reduceForConsensus(job.arena, job.seqs, sg.k, job.red, job.seedMap);
    else job.red.clear();
}

SeedContig* consensusFinish(ConsJob& job, const SeedIndex& sg, const dp_consensus_batch* batch, const uint64_t* seqOff, i64* badBack) {
    if (!job.aligned) return nullptr;
    const int k = sg.k;
    static thread_local std::vector<SeedMatch*> overlap;
    if (!batch || batch->flags[job.group]) {  // not on the device (or refused there): the host loop
        multiAlignerCore(job.arena, job.seqs, job.red, job.seedMap, k, overlap);
    } else {
        // consensus sequence (LoadSequence :35-42) and the per-sequence matches from the device batch
        const size_t ns = job.seqs.size();
        const uint32_t clen = batch->cons_len[job.group];
        const int32_t* c = batch->cons + batch->cons_off[job.group];
        SeedSeq* cons = job.arena.make();
        int32_t* d = job.arena.alloc(clen);
        memcpy(d, c, (size_t)clen * 4);
        cons->seg = d;
        cons->n = (int)clen;
        cons->length = -k;
        for (uint32_t i = 0; i < clen; i += 2) cons->length += d[i] + k;
        std::vector<SeedMatch>& pool = consensusMatchPool(ns);
        static thread_local std::vector<SeedMatch*> matches;
        matches.assign(ns, nullptr);
        for (size_t i = 0; i < ns; i++) {
            if (!job.red[i]) continue;
            SeedMatch* m = &pool[i];
            const uint32_t sq = job.firstSeq + (uint32_t)i;
            const uint32_t n = batch->match_len[sq];
            const int32_t* a = batch->match_a + seqOff[sq];
            const int32_t* b = batch->match_b + seqOff[sq];
            m->MatchA.assign(a, a + n);
            m->MatchB.resize(n);
            for (uint32_t t = 0; t < n; t++) m->MatchB[t] = job.seedMap[i][(size_t)b[t]];
            m->SeqA = nullptr;
            m->SeqB = job.seqs[i];
            m->QueryID = 0;
            m->ReverseComplementQuery = false;
            matches[i] = m;
        }
        for (i64 i = (i64)matches.size() - 1; i >= 0; i--) {  // :258-266 swap-with-last removal
            SeedMatch* m = matches[(size_t)i];
            if (!m || m->MatchA.size() < 3) {
                matches[(size_t)i] = matches.back();
                matches.pop_back();
            } else {
                m->SeqA = cons;
            }
        }
        overlap.assign(matches.begin(), matches.end());
    }
    if (overlap.size() > 1) return newSeedContig(job.arena, overlap, k, badBack);
    return nullptr;
}

// ---------------------------------------------------------------------------------------------------------------
// commands/command.go:18-56 MakeArgs (unique-prefix aliases) and downpore.go:34-51 parseArgs

void ArgTable::make(const std::vector<std::string>& n, const std::vector<std::string>& d, const std::vector<std::string>& desc) {
    names = n;
    defaults = d;
    descriptions = desc;
    for (size_t i = 0; i < n.size(); i++) args[n[i]] = d[i];
    std::vector<std::string> s = n;
    std::sort(s.begin(), s.end());
    for (size_t i = 0; i < s.size(); i++) {
        if (i == s.size() - 1 || s[i][0] != s[i + 1][0]) {
            alias[s[i]] = s[i].substr(0, 1);
        } else {
            size_t j = i + 1, minLen = 1;
            while (j < s.size() && s[j][0] == s[i][0]) {
                size_t same = 1;
                while (same < s[j].size() && same < s[j - 1].size() && s[j][same] == s[j - 1][same]) same++;
                if (same >= minLen) minLen = same + 1;
                j++;
            }
            if (minLen < 4)
                for (size_t t = i; t < j; t++) alias[s[t]] = s[t].substr(0, minLen);
            i = j - 1;
        }
    }
}

bool ArgTable::parse(int argc, char** argv, std::string& err) {
    std::unordered_map<std::string, std::string> inv;
    for (auto& kv : alias) inv[kv.second] = kv.first;
    for (int i = 2; i < argc; i += 2) {
        std::string name = argv[i];
        size_t p = 0;
        while (p < name.size() && name[p] == '-') p++;
        name = name.substr(p);
        auto it = inv.find(name);
        if (it != inv.end()) name = it->second;
        if (!args.count(name)) {
            err = "Unrecognised argument:" + name;
            return false;
        }
        if (i + 1 >= argc) {
            err = "Missing value for argument:" + name;  // the reference panics on os.Args[i+1]
            return false;
        }
        args[name] = argv[i + 1];
    }
    return true;
}


// ---- packBytes for a whole read (sequence/sequence.go:59-93): 2 bits per base, ((b >> 1) ^ ((b & 4) >> 2)) & 3, four bases per byte with
// the first in the top bits, the last byte's unused bits zero.  `map` hands its reads to the device this way (host_map.cpp): a
// quarter of the bytes over PCIe.  dst receives ceil(n / 4) bytes.
namespace {
size_t packScalar(const unsigned char* src, size_t n, uint8_t* dst) {  // whole groups of four; returns the bases consumed
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        dst[i / 4] = (uint8_t)((baseCode(src[i]) << 6) | (baseCode(src[i + 1]) << 4) | (baseCode(src[i + 2]) << 2) | baseCode(src[i + 3]));
    return i;
}
__attribute__((target("avx2"))) inline __m256i packCodes32(const unsigned char* p) {  // 32 bases -> one packed byte per 32-bit lane
    const __m256i x = _mm256_loadu_si256((const __m256i*)p);
    const __m256i a = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(0x7f));    // b >> 1
    const __m256i b = _mm256_srli_epi16(_mm256_and_si256(x, _mm256_set1_epi8(4)), 2);       // (b & 4) >> 2
    const __m256i c = _mm256_and_si256(_mm256_xor_si256(a, b), _mm256_set1_epi8(3));
    // per byte pair (first, second): x 4, x 1; then per 16-bit pair: x 16, x 1
    return _mm256_madd_epi16(_mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0104)), _mm256_set1_epi32(0x00010010));
}
__attribute__((target("avx2"))) size_t packAvx2(const unsigned char* src, size_t n, uint8_t* dst) {  // 128 bases -> 32 bytes per step
    const __m256i order = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i p01 = _mm256_packus_epi32(packCodes32(src + i), packCodes32(src + i + 32));
        const __m256i p23 = _mm256_packus_epi32(packCodes32(src + i + 64), packCodes32(src + i + 96));
        const __m256i q = _mm256_permutevar8x32_epi32(_mm256_packus_epi16(p01, p23), order);
        _mm256_storeu_si256((__m256i*)(dst + i / 4), q);
    }
    return i;
}
}  // namespace
void packBases(const char* src, size_t n, uint8_t* dst, bool scalarOnly) {
    const bool avx2 = !scalarOnly && __builtin_cpu_supports("avx2");
    const unsigned char* s = (const unsigned char*)src;
    size_t i = avx2 ? packAvx2(s, n, dst) : 0;
    i += packScalar(s + i, n - i, dst + i / 4);
    if (i < n) {  // one to three bases in the last byte
        uint32_t v = 0;
        for (size_t j = 0; j < 4; j++) v = (v << 2) | (i + j < n ? baseCode(s[i + j]) : 0u);
        dst[i / 4] = (uint8_t)v;
    }
}

}  // namespace dph
