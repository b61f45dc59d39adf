// ORACLE — test infrastructure only (see oracle.hpp).  L2 seed space:
// seeds/seeds.go, seeds/sequence.go (cited parts), seeds/alignment.go.
#include "oracle.hpp"

#include <algorithm>
#include <stdexcept>

namespace dpo {

// ---------------------------------------------------------------------------------------------
// seeds/seeds.go

SeedIndex::SeedIndex(int k) : seedSize(k) {  // :23-31
    size_t size = 1;
    for (int j = k; j > 0; j--) size *= 4;
    kmers.assign(size, 0);
    kmerMap.assign(size, 0);
}

// seeds/seeds.go:33-50
// The scan half of NewSeedSequence: reads only kmers/kmerMap, so several reads can be scanned by different threads
// (bench's all-cores CPU baseline; the reference does the same with num_workers goroutines).
std::shared_ptr<std::vector<i64>> SeedIndex::scanSegments(const PackedSeq& seq) const {
    int k = seedSize;
    i64 count = seq.countKmers(seq.length, k, kmers.data());
    auto store = std::make_shared<std::vector<i64>>((size_t)(count * 2 + 1), 0);
    // WriteSegments writes exactly what CountKmers counted unless the count early-exited (then the
    // reference would write out of bounds and panic).
    std::vector<i64> tmp((size_t)(seq.nbytes() * 8 + 64), 0);
    i64 wrote = packedWriteSegmentsAsm(seq.data(), (i64)seq.nbytes(), 4 - seq.firstLen, 4 - seq.finalLen, k,
                                       kmers.data(), tmp.data());
    if (wrote != count * 2 + 1) throw std::runtime_error("oracle: WriteSegments overflow (reference would panic)");
    std::copy(tmp.begin(), tmp.begin() + wrote, store->begin());
    for (size_t i = 1; i < store->size(); i += 2) (*store)[i] = (i64)kmerMap[(size_t)(*store)[i]];
    return store;
}

SeedSequence* SeedIndex::newSeedSequence(const PackedSeq& seq) { return newSeedSequence(seq, scanSegments(seq)); }

SeedSequence* SeedIndex::newSeedSequence(const PackedSeq& seq, std::shared_ptr<std::vector<i64>> store) {
    SeedSequence* s = arena.make();
    s->store = store;
    s->lo = 0;
    s->n = store->size();
    s->length = seq.length;
    s->id = seq.id;
    s->name = std::make_shared<std::string>(seq.getName());
    s->offset = seq.offset;
    s->inset = seq.inset;
    s->rc = false;
    return s;
}

void SeedIndex::addSeedKmer(i64 kmer) {  // :132-141
    if (!kmers[(size_t)kmer]) {
        kmers[(size_t)kmer] = 1;
        kmerMap[(size_t)kmer] = (int32_t)size;
        while ((i64)sequenceSets.size() <= size) {
            sequenceSets.emplace_back();
            seedMap.push_back(-1);
        }
        seedMap[(size_t)size] = kmer;
        size++;
    }
}

// seeds/seeds.go:62-156
void SeedIndex::addSeeds(const PackedSeq& seq, i64 minSeeds, const double* ranks) {
    int k = seedSize;
    i64 mask = ((i64)1 << (2 * k)) - 1;
    // :72 count := seq.CountKmers(minSeeds,...) is discarded (:74 count = 0)
    i64 count = 0;
    if (count < minSeeds) {
        std::vector<u64> topN((size_t)(minSeeds - count), 0);
        std::vector<double> topNValues((size_t)(minSeeds - count), 0.0);
        i64 kmer = seq.kmerAt(0, k);
        i64 nextIndex = k;
        i64 L = seq.length;
        while (nextIndex < L - k) {
            bool reset = false;
            double bestValue = 0.0;
            i64 bestSeed = 0;
            for (i64 i = 0; nextIndex < L && i < k; i++) {
                kmer = seq.nextKmer(kmer, mask, nextIndex);
                nextIndex++;
                if (kmers[(size_t)kmer]) {
                    reset = true;
                    break;
                }
                double value = ranks[kmer];
                if (seq.qual) value *= (double)(*seq.qual)[seq.qlo + (size_t)(nextIndex - k / 2)];  // :99-101
                if (value > bestValue) {
                    bestValue = value;
                    bestSeed = kmer;
                }
            }
            if (!reset) {
                size_t n = 0;
                for (; n < topNValues.size() && topNValues[n] < bestValue; n++) {
                    if (n > 0) {
                        topNValues[n - 1] = topNValues[n];
                        topN[n - 1] = topN[n];
                    }
                }
                if (n > 0) {
                    topNValues[n - 1] = bestValue;
                    topN[n - 1] = (u64)bestSeed;
                }
            }
            nextIndex += k;
            if (nextIndex < L - k) kmer = seq.kmerAt(nextIndex, k);
            nextIndex += k;
        }
        for (u64 km : topN) {
            addSeedKmer((i64)km);
            addSeedKmer((i64)reverseComplementKmer(km, k));
        }
    }
}

// seeds/seeds.go:160-200
void SeedIndex::addSingleSeeds(const PackedSeq& seq, i64 seedRate, const double* ranks) {
    int k = seedSize;
    i64 mask = ((i64)1 << (2 * k)) - 1;
    for (i64 i = 0; i < seq.length - seedRate; i += seedRate) {
        i64 count = seq.countKmersBetween(i, i + seedRate, 1, k, kmers.data());
        if (count == 0) {
            i64 end = i + seedRate;
            i64 kmer = seq.kmerAt(i, k);
            double bestValue = ranks[kmer];
            i64 bestKmer = kmer;
            for (i64 j = i + k; j < end; j++) {
                kmer = seq.nextKmer(kmer, mask, j);
                double value = ranks[kmer];
                if (value > bestValue) {
                    bestValue = value;
                    bestKmer = kmer;
                }
            }
            addSeedKmer(bestKmer);
        }
    }
}

// seeds/seeds.go:272-290
void SeedIndex::addSequence(SeedSequence* seq) {
    i64 maxSeed = 0;
    const i64* sg = seq->seg();
    for (size_t i = 1; i < seq->n; i += 2)
        if (sg[i] > maxSeed) maxSeed = sg[i];
    IntSet seedSet(maxSeed + 1);
    for (size_t i = 1; i < seq->n; i += 2) seedSet.add((u64)sg[i]);
    sequences.push_back(seq);
    seedSets.push_back(std::move(seedSet));
}

// seeds/seeds.go:292-305 + :372-384 (one worker owns every seed; result independent of numWorkers)
void SeedIndex::indexSequences() {
    for (i64 i = (i64)sequences.size() - 1; i >= 0; i--) {
        SeedSequence* s = sequences[(size_t)i];
        const i64* sg = s->seg();
        for (size_t j = 1; j < s->n; j += 2) {
            i64 seed = sg[j];
            sequenceSets[(size_t)seed].add((u64)i);
        }
    }
}

// seeds/seeds.go:335-353
std::vector<u64> SeedIndex::matches(const SeedSequence* query, double hitFraction) const {
    std::vector<const IntSet*> all;
    i64 prevSeed = -1;
    u64 maxSeqs = (u64)sequences.size();
    const i64* sg = query->seg();
    for (size_t i = 1; i < query->n; i += 2) {
        i64 seed = sg[i];
        const IntSet* adj = &sequenceSets[(size_t)seed];
        if (seed != prevSeed && adj->size() < maxSeqs) {
            all.push_back(adj);
            prevSeed = seed;
        }
    }
    if (all.size() < 5) return {};
    i64 minCount = (i64)(hitFraction * (double)all.size() + 0.5);
    return getSharedIDs(all, minCount, true);
}

// ---------------------------------------------------------------------------------------------
// seeds/sequence.go

u64 reverseComplementKmer(u64 seed, int k) {  // :125-132
    u64 rc = 0;
    for (int j = 0; j < k; j++) {
        rc = (rc << 2) | ((seed ^ 3) & 3);
        seed >>= 2;
    }
    return rc;
}

// :134-159
SeedSequence* ssReverseComplement(SeedSequence* s, int k, SeedIndex& index) {
    if (s->reverseComplement != nullptr) return s->reverseComplement;
    size_t n = s->n;
    auto store = std::make_shared<std::vector<i64>>(n, 0);
    const i64* sg = s->seg();
    for (size_t i = 0; i < n; i++) {
        if ((i & 1) == 0) {
            (*store)[n - 1 - i] = sg[i];
        } else {
            u64 seed = (u64)index.seedMap[(size_t)sg[i]];
            u64 rc = reverseComplementKmer(seed, k);
            (*store)[n - 1 - i] = (i64)index.kmerMap[(size_t)rc];
        }
    }
    SeedSequence* ns = index.arena.make();
    ns->store = store;
    ns->lo = 0;
    ns->n = n;
    ns->id = s->id;
    ns->name = nullptr;
    ns->length = s->length;
    ns->offset = s->offset;
    ns->inset = s->inset;
    ns->reverseComplement = s;
    ns->rc = !s->rc;
    ns->Parent = s->Parent;
    return ns;
}

// :46-50
SeedSequence* ssSubSequence(Arena& a, SeedSequence* s, i64 start, i64 end, i64 length, i64 offset, i64 inset) {
    SeedSequence* sub = a.make();
    sub->store = s->store;
    sub->lo = s->lo + (size_t)(start * 2);
    sub->n = (size_t)(end * 2 + 3 - start * 2);
    if (sub->lo + sub->n > s->lo + s->n || end < start - 1)
        throw std::runtime_error("oracle: SeedSequence.SubSequence slice bounds (reference would panic)");
    sub->length = length;
    sub->offset = offset;
    sub->inset = inset;
    sub->rc = s->rc;
    sub->id = s->id;
    sub->Parent = s;
    return sub;
}

i64 SeedSequence::getSeedOffset(i64 index, int k) const {  // :1239-1246
    index = index * 2 + 1;
    const i64* sg = seg();
    i64 off = sg[0];
    for (i64 i = 2; i < index; i += 2) off += sg[i] + k;
    return off;
}
i64 SeedSequence::getSeedOffsetFromEnd(i64 index, int k) const {  // :1269-1276
    index = index * 2 + 1;
    const i64* sg = seg();
    i64 off = sg[n - 1];
    for (i64 i = (i64)n - 3; i > index; i -= 2) off += sg[i] + k;
    return off;
}
i64 SeedSequence::getMaxSeed() const {  // :1286-1294
    i64 m = 0;
    const i64* sg = seg();
    for (size_t i = 1; i < n; i += 2)
        if (sg[i] > m) m = sg[i];
    return m;
}

// :54-82
SeedSequence* ssTrimmed(Arena& a, SeedSequence* s, i64 startOffset, i64 startSeed, i64 endOffset, i64 endSeed,
                        int k, i64* startSeedOut) {
    const i64* sg = s->seg();
    while (startSeed > 0 && startOffset >= sg[startSeed * 2] + k) {
        startOffset -= sg[startSeed * 2] + k;
        startSeed--;
    }
    i64 numSeeds = (i64)s->n / 2;
    while (endSeed < numSeeds - 1 && endOffset >= sg[endSeed * 2 + 2] + k) {
        endOffset -= sg[endSeed * 2 + 2] + k;
        endSeed++;
    }
    i64 offset = s->getSeedOffset(startSeed, k) - startOffset;
    i64 inset = s->getSeedOffsetFromEnd(endSeed, k) - endOffset;
    SeedSequence* trimmed;
    if (s->rc)
        trimmed = ssSubSequence(a, s, startSeed, endSeed, s->length - offset - inset, s->offset + inset, s->inset + offset);
    else
        trimmed = ssSubSequence(a, s, startSeed, endSeed, s->length - offset - inset, s->offset + offset, s->inset + inset);
    auto store = std::make_shared<std::vector<i64>>(trimmed->seg(), trimmed->seg() + trimmed->n);
    (*store)[0] = startOffset;
    (*store)[trimmed->n - 1] = endOffset;
    trimmed->store = store;
    trimmed->lo = 0;
    if (startSeedOut) *startSeedOut = startSeed;
    return trimmed;
}

// :85-123
SeedSequence* ssReduced(Arena& a, SeedSequence* s, const IntSet& whitelist, int k, i64 minSeeds,
                        std::vector<i64>* index) {
    i64 count = 0;
    i64 n = (i64)s->n;
    const i64* sg = s->seg();
    i64 prev = -1;
    for (i64 i = 1; i < n; i += 2) {
        i64 next = sg[i];
        if (next != prev && whitelist.contains((u64)next)) {
            count++;
            prev = next;
        }
    }
    if (count < minSeeds) return nullptr;
    auto segs = std::make_shared<std::vector<i64>>((size_t)(count * 2 + 1), 0);
    i64 offset = sg[0];
    if (index) index->assign((size_t)count, 0);
    prev = -1;
    i64 j = 0;
    for (i64 i = 1; i < n; i += 2) {
        i64 seed = sg[i];
        if (prev != seed && whitelist.contains((u64)seed)) {
            (*segs)[(size_t)j] = offset;
            (*segs)[(size_t)j + 1] = seed;
            if (index) (*index)[(size_t)(j / 2)] = i / 2;
            j += 2;
            offset = sg[i + 1];
            prev = seed;
        } else {
            offset += sg[i + 1] + k;
        }
    }
    (*segs)[(size_t)j] = offset;
    SeedSequence* r = a.make();
    r->store = segs;
    r->lo = 0;
    r->n = segs->size();
    r->length = s->length;
    r->offset = s->offset;
    r->inset = s->inset;
    r->rc = s->rc;
    r->id = s->id;
    r->Parent = s;
    return r;
}

// :800-816
void smReverseComplement(SeedMatch& m, int k, SeedIndex& index) {
    m.SeqA = ssReverseComplement(m.SeqA, k, index);
    m.SeqB = ssReverseComplement(m.SeqB, k, index);
    i64 end = (i64)m.MatchA.size() - 1;
    i64 lengthA = (i64)m.SeqA->n / 2 - 1;
    i64 lengthB = (i64)m.SeqB->n / 2 - 1;
    for (i64 i = 0; i < (i64)m.MatchA.size() / 2; i++) {
        std::swap(m.MatchA[(size_t)i], m.MatchA[(size_t)(end - i)]);
        std::swap(m.MatchB[(size_t)i], m.MatchB[(size_t)(end - i)]);
    }
    for (size_t i = 0; i < m.MatchA.size(); i++) {
        m.MatchA[i] = lengthA - m.MatchA[i];
        m.MatchB[i] = lengthB - m.MatchB[i];
    }
}

// :830-858
// Returns false (a = b = 0) where the reference would panic with an index out of range.
bool smGetBasesCovered(const SeedMatch& m, int k, i64* a, i64* b) {
    *a = *b = 0;
    if (m.MatchA.empty()) return false;
    i64 countA = (i64)m.MatchA.size() * k;
    i64 countB = countA;
    i64 prevA = m.MatchA[0];
    i64 prevB = m.MatchB[0];
    const i64* sa = m.SeqA->seg();
    const i64* sb = m.SeqB->seg();
    const i64 naSeg = (i64)m.SeqA->n, nbSeg = (i64)m.SeqB->n;
    for (size_t i = 1; i < m.MatchA.size(); i++) {
        i64 s = m.MatchA[i];
        if (s * 2 >= naSeg || m.MatchB[i] * 2 >= nbSeg || prevA * 2 + 2 >= naSeg || prevB * 2 + 2 >= nbSeg || prevA < 0 ||
            prevB < 0)
            return false;
        i64 d1 = sa[prevA * 2 + 2];
        i64 d2 = sb[prevB * 2 + 2];
        for (i64 j = prevA + 2; j <= s; j++) d1 += sa[j * 2] + k;
        i64 s2 = m.MatchB[i];
        for (i64 j = prevB + 2; j <= s2; j++) d2 += sb[j * 2] + k;
        if (d1 < 0) countA += d1;
        if (d2 < 0) countB += d2;
        prevB = s2;
        prevA = s;
    }
    *a = countA;
    *b = countB;
    return true;
}

// :1190-1237
void smGetBaseIndex(const SeedMatch& m, i64 aIndex, int k, i64* indexOut, i64* basesOut, i64* distOut) {
    const i64* sa = m.SeqA->seg();
    const i64* sb = m.SeqB->seg();
    i64 nb = (i64)m.SeqB->n;
    i64 before = 0;
    while (before < (i64)m.MatchA.size() && m.MatchA[(size_t)before] <= aIndex) before++;
    if (before == 0) {
        i64 offset = 0;
        for (i64 i = m.MatchA[0]; i > aIndex; i--) offset += sa[i * 2] + k;
        i64 bIndex = m.MatchB[0];
        i64 distance = 0;
        for (i64 i = bIndex * 2; i > 0 && offset > 0; i -= 2) {
            offset -= sb[i] + k;
            distance += sb[i] + k;
            bIndex--;
        }
        if (bIndex == 0) {
            *indexOut = 0;
            *basesOut = -offset;
            *distOut = distance + offset;
            return;
        }
        *indexOut = bIndex;
        *basesOut = -offset;
        *distOut = distance;
        return;
    }
    before--;
    i64 bIndex = m.MatchB[(size_t)before];
    if (aIndex == m.MatchA[(size_t)before]) {
        *indexOut = bIndex;
        *basesOut = 0;
        *distOut = 0;
        return;
    }
    i64 offset = 0;
    for (i64 i = m.MatchA[(size_t)before] + 1; i <= aIndex; i++) offset += sa[i * 2] + k;
    i64 distance = 0;
    for (i64 i = bIndex * 2 + 2; i < nb && offset >= sb[i]; i += 2) {
        offset -= sb[i] + k;
        distance += sb[i] + k;
        bIndex++;
    }
    if (bIndex >= nb / 2) {
        *indexOut = bIndex - 1;
        *basesOut = offset;
        *distOut = distance + offset;
        return;
    }
    *indexOut = bIndex;
    *basesOut = offset;
    *distOut = distance + offset;
}

// ---------------------------------------------------------------------------------------------
// seeds/alignment.go:270-616 — overlap chaining

void gapRange(i64 gap, int k, i64* mn, i64* mx) {  // :411-424
    i64 minGap = (gap * 2) / 3 - k;
    i64 maxGap = (gap * 3) / 2 + k + 1;
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

namespace {
struct PairState {  // :286-295
    i64 aPos = 0, bPos = 0, aGap = 0, bGap = 0, aGapIndex = 0, length = 0;
    PairState* prev = nullptr;
};
}  // namespace

std::vector<SeedMatch> SeedAligner::pairwiseAlignments(SeedSequence* a, SeedSequence* b, const IntSet& aSet,
                                                       const IntSet& bSet, i64 minMatches, int k) {
    const i64* aSegments = a->seg();
    const i64 aN = (i64)a->n;
    const i64* bSegments = b->seg();
    const i64 bN = (i64)b->n;
    if (minMatches == 0) minMatches = 1;
    // state pool :298-324 — a free list of 10 000 states; modelled as an allocator with the same capacity.
    const size_t POOL = 10000, OPEN = 500, RESULTS = 500;
    std::vector<std::unique_ptr<PairState>> pool;
    size_t live = 0;
    auto popState = [&]() -> PairState* {
        if (live >= POOL) throw std::runtime_error("oracle: seedAligner state pool exhausted (reference would panic)");
        pool.emplace_back(new PairState());
        live++;
        return pool.back().get();
    };
    auto pushState = [&](PairState*) { live--; };

    // prepareInitial :341-388
    const size_t redCap = (size_t)maxLength, mapCap = (size_t)maxLength / 2;
    std::vector<i64> aRedBuf(redCap, 0), aMapping(mapCap, 0);
    std::vector<PairState*> initials((size_t)maxLength, nullptr);
    i64 maxAIndex = aN - minMatches * 2 + 1;
    i64 aLen = 0, offset = -k, startSize = 0, prevSeedA = -1;
    for (i64 i = 1; i < aN; i += 2) {
        i64 aSeed = aSegments[i];
        if (!bSet.contains((u64)aSeed)) {
            offset += aSegments[i - 1] + k;
            maxAIndex--;
            continue;
        }
        if (aSeed == prevSeedA && (i >= aN - 2 || aSegments[i + 2] == prevSeedA)) {
            offset += aSegments[i - 1] + k;
            maxAIndex--;
            continue;
        }
        prevSeedA = aSeed;
        offset += aSegments[i - 1] + k;
        if ((size_t)(aLen * 2 + 1) >= redCap || (size_t)aLen >= mapCap)
            throw std::runtime_error("oracle: seedAligner reduced buffer overflow (reference would panic)");
        aRedBuf[(size_t)(aLen * 2)] = offset;
        aRedBuf[(size_t)(aLen * 2 + 1)] = aSeed;
        aMapping[(size_t)aLen] = i / 2;
        offset = -k;
        if (aLen <= maxAIndex) {
            PairState* st = popState();
            st->aPos = aLen * 2 + 1;
            st->length = 0;
            st->prev = nullptr;
            initials[(size_t)aLen] = st;
            startSize++;
        }
        aLen++;
    }
    if ((size_t)(aLen * 2) >= redCap) throw std::runtime_error("oracle: seedAligner reduced buffer overflow (reference would panic)");
    aRedBuf[(size_t)(aLen * 2)] = 0;
    while (startSize > 0 && initials[(size_t)(startSize - 1)]->aPos > maxAIndex) {
        startSize--;
        pushState(initials[(size_t)startSize]);
    }
    const i64 initialSize = startSize;
    const i64* aRed = aRedBuf.data();
    const i64 aRedLen = aLen * 2 + 1;

    std::vector<PairState*> open(OPEN, nullptr), results(RESULTS, nullptr);
    i64 openSize = 0, resultsSize = 0;

    auto removeOpenState = [&](i64 index) {  // :390-409
        PairState* s = open[(size_t)index];
        open[(size_t)index] = open[(size_t)(openSize - 1)];
        openSize--;
        if (s->length >= minMatches) {
            if ((s->length * 2) / 3 > minMatches) minMatches = (s->length * 2) / 3;
            if ((size_t)resultsSize >= RESULTS) throw std::runtime_error("oracle: seedAligner results overflow (reference would panic)");
            results[(size_t)resultsSize++] = s;
        } else {
            for (PairState* p = s; p != nullptr; p = p->prev) pushState(p);
        }
    };

    const i64 bLen = bN;
    i64 maxBIndex = bN - minMatches * 2 + 1;
    i64 bOffset = 0;
    i64 prevSeed = -1;
    for (i64 bIndex = 1; bIndex < bLen; bIndex += 2) {
        i64 bSeed = bSegments[bIndex];
        if (!aSet.contains((u64)bSeed)) {
            bOffset += bSegments[bIndex + 1] + k;
            continue;
        }
        if (bSeed == prevSeed && (bIndex >= bN - 2 || bSegments[bIndex + 2] == prevSeed)) {
            bOffset += bSegments[bIndex + 1] + k;
            continue;
        }
        prevSeed = bSeed;
        i64 found = -1;
        // searchMatch: :465-547 (the "dominated chain" block :499-514 is dead code: found is always -1 there)
        for (i64 i = openSize - 1; i >= 0; i--) {
            PairState* s = open[(size_t)i];
            s->bGap += bOffset;
            i64 minGap, maxGap;
            gapRange(s->bGap, k, &minGap, &maxGap);
            bool brokeOut = false;
            while (s->aGap < minGap) {
                if (s->aGapIndex >= aRedLen) {
                    removeOpenState(i);
                    brokeOut = true;
                    break;
                }
                s->aGap += aRed[s->aGapIndex + 1] + k;
                s->aGapIndex += 2;
            }
            if (brokeOut) break;  // break searchMatch
            if (s->aGap <= maxGap) {
                i64 g = s->aGap;
                bool extended = false;
                for (i64 j = s->aGapIndex; j < aRedLen && g <= maxGap; j += 2) {
                    if (aRed[j] == bSeed) {
                        found = j;
                        PairState* ns = popState();
                        ns->prev = s;
                        ns->aPos = j;
                        ns->bPos = bIndex;
                        ns->aGapIndex = j + 2;
                        ns->aGap = aRed[j + 1];
                        ns->bGap = bSegments[bIndex + 1];
                        ns->length = s->length + 1;
                        open[(size_t)i] = ns;
                        if ((ns->length * 2) / 3 > minMatches) {
                            minMatches = (ns->length * 2) / 3;
                            maxBIndex = bN - minMatches * 2 + 1;
                        }
                        extended = true;
                        break;
                    }
                    g += aRed[j + 1] + k;
                }
                if (extended) break;  // break searchMatch
            }
            if (s->length + (bN - bIndex) < minMatches) {
                removeOpenState(i);
            } else {
                s->bGap += bSegments[bIndex + 1] + k;
            }
        }
        bOffset = 0;
        if (bIndex <= maxBIndex) {  // :550-587
            for (i64 i = 0; i < initialSize; i++) {
                PairState* s = initials[(size_t)i];
                i64 aPos = s->aPos;
                if (aPos != found && aRed[aPos] == bSeed) {
                    if (found != -1) {
                        for (i64 j = 0; j < openSize; j++) {
                            if (open[(size_t)j]->bPos == bIndex && open[(size_t)j]->aPos == aPos) {
                                found = aPos;
                                break;
                            }
                        }
                    }
                    if (found == aPos || openSize >= (i64)OPEN) continue;
                    PairState* ns = popState();
                    ns->aPos = s->aPos;
                    ns->bPos = bIndex;
                    ns->aGapIndex = s->aPos + 2;
                    ns->aGap = aRed[s->aPos + 1];
                    ns->bGap = bSegments[bIndex + 1];
                    ns->length = 1;
                    ns->prev = nullptr;
                    open[(size_t)openSize++] = ns;
                }
            }
        }
    }
    for (i64 i = 0; i < openSize; i++) {  // :597-604
        PairState* s = open[(size_t)i];
        if (s->length >= minMatches) {
            if ((size_t)resultsSize >= RESULTS) throw std::runtime_error("oracle: seedAligner results overflow (reference would panic)");
            results[(size_t)resultsSize++] = s;
        }
    }
    std::vector<SeedMatch> out;
    for (i64 i = resultsSize - 1; i >= 0; i--) {  // :608-614 + extractMatch :326-335
        PairState* s = results[(size_t)i];
        SeedMatch m;
        m.MatchA.assign((size_t)s->length, 0);
        m.MatchB.assign((size_t)s->length, 0);
        for (PairState* p = s; p != nullptr; p = p->prev) {
            m.MatchA[(size_t)(p->length - 1)] = aMapping[(size_t)(p->aPos / 2)];
            m.MatchB[(size_t)(p->length - 1)] = p->bPos / 2;
        }
        m.SeqA = a;
        m.SeqB = b;
        out.push_back(std::move(m));
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// seeds/sequence.go:361-576 — map chaining (Match / dynamicMatch / extendChain)

namespace {
struct Chain {
    std::vector<i64> a, b;  // one backing array per started chain (Go: shared backing arrays)
};
struct ChainRef {  // a Go slice header into a chain's backing array: (array, len)
    Chain* c = nullptr;
    size_t len = 0;
};

// extendChain :476-576.  a = query, b = seq.
void extendChain(SeedSequence* a, SeedSequence* b, std::vector<ChainRef>& chains, i64 aIndex, i64 bIndex, int k,
                 Chain* cur) {
    const i64* as = a->seg();
    const i64* bs = b->seg();
    const i64 an = (i64)a->n, bn = (i64)b->n;
    i64 offsetA = as[aIndex + 1];
    i64 offsetB = bs[bIndex + 1];
    aIndex += 2;
    bIndex += 2;
    while (aIndex < an && bIndex < bn) {
        i64 aSeedIndex = aIndex / 2;
        i64 minBOffset, maxBOffset;
        if (offsetA < 0) {
            minBOffset = -k;
            maxBOffset = 0;
        } else {
            minBOffset = (offsetA * 2) / 3 - k;
            maxBOffset = (offsetA * 3) / 2 + k;
        }
        while (maxBOffset < offsetB) {
            offsetA += as[aIndex + 1] + k;
            aIndex += 2;
            if (aIndex >= an) return;
            aSeedIndex = aIndex / 2;
            minBOffset = (offsetA * 2) / 3 - k;
            maxBOffset = (offsetA * 3) / 2 + k;
        }
        while (offsetB < minBOffset) {
            offsetB += bs[bIndex + 1] + k;
            bIndex += 2;
            if (bIndex >= bn) return;
        }
        i64 oldBIndex = bIndex;
        i64 oldBOffset = offsetB;
        bool matched = false;
        i64 seedA = as[aIndex];
        while (offsetB <= maxBOffset) {
            if (seedA == bs[bIndex]) {
                ChainRef& ex = chains[(size_t)aSeedIndex];
                if (ex.c != nullptr) {
                    if (bIndex / 2 == ex.c->b[ex.len - 1] && ex.len > cur->a.size()) return;
                }
                cur->a.push_back(aSeedIndex);
                cur->b.push_back(bIndex / 2);
                chains[(size_t)aSeedIndex] = ChainRef{cur, cur->a.size()};
                offsetA = as[aIndex + 1];
                offsetB = bs[bIndex + 1];
                aIndex += 2;
                bIndex += 2;
                matched = true;
                break;
            } else {
                offsetB += bs[bIndex + 1] + k;
                bIndex += 2;
                if (bIndex >= bn) break;
            }
        }
        if (!matched) {
            offsetA += as[aIndex + 1] + k;
            aIndex += 2;
            offsetB = oldBOffset;
            bIndex = oldBIndex;
        }
    }
}

// dynamicMatch :401-471
std::vector<SeedMatch> dynamicMatch(SeedSequence* seq, SeedSequence* query, i64 minMatch, int k) {
    if (minMatch == 0) minMatch = 1;
    const i64* qs = query->seg();
    const i64* ss = seq->seg();
    const i64 qn = (i64)query->n, sn = (i64)seq->n;
    std::vector<ChainRef> chains((size_t)(qn / 2));
    std::vector<std::unique_ptr<Chain>> owned;
    std::vector<SeedMatch> good;
    for (i64 qIndex = 1; qIndex < qn - minMatch * 2 + 2; qIndex += 2) {
        if (qs[qIndex - 1] < 0 && qIndex > 1 && qs[qIndex + 1] < 0 && qs[qIndex] == qs[qIndex - 2] &&
            qs[qIndex] == qs[qIndex + 2])
            continue;
        i64 qsi = qIndex / 2;
        if (chains[(size_t)qsi].c != nullptr) continue;
        i64 prevSeed = -1;
        for (i64 i = 1; i < sn - minMatch * 2 + 2; i += 2) {
            i64 nextSeed = ss[i];
            ChainRef& cr = chains[(size_t)qsi];
            if (nextSeed == qs[qIndex] && nextSeed != prevSeed && (cr.c == nullptr || cr.c->b[cr.len - 1] != i / 2)) {
                owned.emplace_back(new Chain());
                Chain* c = owned.back().get();
                c->a.push_back(qsi);
                c->b.push_back(i / 2);
                chains[(size_t)qsi] = ChainRef{c, 1};
                extendChain(query, seq, chains, qIndex, i, k, c);
                if ((i64)c->a.size() >= minMatch) {
                    i64 nextLength = ((i64)c->a.size() * 2) / 3;
                    if (nextLength > minMatch) {
                        minMatch = nextLength;
                        for (i64 j = (i64)good.size() - 1; j >= 0; j--) {
                            if ((i64)good[(size_t)j].MatchA.size() < nextLength) {
                                good[(size_t)j] = std::move(good.back());
                                good.pop_back();
                            }
                        }
                    }
                    SeedMatch m;
                    m.MatchA = c->a;
                    m.MatchB = c->b;
                    m.SeqA = query;
                    m.SeqB = seq;
                    m.QueryID = -1;
                    good.push_back(std::move(m));
                    i64 remaining = 0;
                    for (auto& r : chains)
                        if (r.c == nullptr) remaining++;
                    if (remaining < (i64)c->a.size()) return good;
                }
            }
            prevSeed = nextSeed;
        }
    }
    return good;
}
}  // namespace

// Match :361-394
std::vector<SeedMatch> ssMatch(Arena& a, SeedSequence* seq, SeedSequence* query, const IntSet* querySet,
                               const IntSet* seqSet, i64 minMatch, int k) {
    SeedSequence* s = seq;
    SeedSequence* q = query;
    std::vector<i64> qIndex, sIndex;
    bool haveQ = false, haveS = false;
    if (querySet) {
        s = ssReduced(a, seq, *querySet, k, minMatch, &sIndex);
        haveS = (s != nullptr);
    }
    if (seqSet) {
        q = ssReduced(a, query, *seqSet, k, minMatch, &qIndex);
        haveQ = (q != nullptr);
    }
    if (s == nullptr || q == nullptr) return {};
    std::vector<SeedMatch> ms = dynamicMatch(s, q, minMatch, k);
    for (auto& m : ms) {
        if (haveQ)
            for (auto& p : m.MatchA) p = qIndex[(size_t)p];
        if (haveS)
            for (auto& p : m.MatchB) p = sIndex[(size_t)p];
        m.SeqA = query;
        m.SeqB = seq;
    }
    return ms;
}

// ---------------------------------------------------------------------------------------------
// seeds/alignment.go:23-268 — multiAligner.Consensus (a fresh aligner per call: overlap/combine.go:186)

SeedSequence* multiAlignerConsensus(Arena& arena, std::vector<SeedSequence*>& seqs, int k,
                                    std::vector<std::unique_ptr<SeedMatch>>& matchesOut) {
    const size_t ns = seqs.size();
    // :25-44
    i64 maxSeed = 100;
    for (auto* s : seqs) {
        i64 ms = s->getMaxSeed();
        if (ms > maxSeed) maxSeed = ms;
    }
    std::vector<IntSet> tempSets;
    for (size_t i = 0; i < ns; i++) tempSets.emplace_back(maxSeed + 1);
    for (size_t i = 0; i < ns; i++) {
        tempSets[i].clear();
        const i64* sg = seqs[i]->seg();
        for (size_t j = 1; j < seqs[i]->n; j += 2) tempSets[i].add((u64)sg[j]);
    }
    std::vector<const IntSet*> ptrs;
    for (auto& t : tempSets) ptrs.push_back(&t);
    IntSet useSeeds = IntSet::fromUInts(getSharedIDs(ptrs, 2, true));  // :45

    // :48-57
    std::vector<std::vector<i64>> seedMap(ns);
    std::vector<SeedSequence*> reds(ns, nullptr);
    for (size_t i = 0; i < ns; i++) reds[i] = ssReduced(arena, seqs[i], useSeeds, k, 1, &seedMap[i]);
    auto segp = [&](size_t i) -> const i64* { return reds[i] ? reds[i]->seg() : nullptr; };
    auto segn = [&](size_t i) -> i64 { return reds[i] ? (i64)reds[i]->n : 0; };

    // :58-68 — pos/offset/gaps start at length 30 and grow to len(seqs)
    size_t stateLen = std::max<size_t>(30, ns);
    std::vector<i64> pos(stateLen, 0), offs(stateLen, 0), gaps(stateLen, 0);
    for (size_t i = 0; i < ns; i++) {
        pos[i] = -1;
        offs[i] = 0;
        gaps[i] = 50;
    }
    std::vector<i64> consensus;
    std::vector<std::unique_ptr<SeedMatch>> matches(ns);
    for (size_t i = 0; i < ns; i++) {
        if (reds[i]) {
            matches[i].reset(new SeedMatch());
            matches[i]->SeqB = seqs[i];
        }
    }
    std::vector<i64> supported(ns, 0), dist(ns, 0);
    bool finished = false;
    while (!finished) {
        i64 fCount = 0;
        i64 near = 100000;
        for (size_t i = 0; i < ns; i++) {  // :85-136
            const i64* segment = segp(i);
            i64 sl = segn(i);
            i64 p = pos[i];
            supported[i] = 0;
            if (segment == nullptr || p >= (sl - 1) / 2 - 1) {
                fCount++;
                continue;
            }
            i64 d = segment[p * 2 + 2] - offs[i];
            dist[i] = d;
            if (d < near && d > -k) {
                i64 nextSeed = segment[p * 2 + 3];
                i64 minD, maxD;
                gapRange(d + gaps[i], k, &minD, &maxD);
                minD -= gaps[i];
                maxD -= gaps[i];
                if (near > maxD) near = maxD;
                supported[i] = 1;
                for (size_t j = 0; j < ns; j++) {
                    const i64* segment2 = segp(j);
                    i64 sl2 = segn(j);
                    if (segment2 == nullptr || j == i) continue;
                    i64 p2 = pos[j] + 1;
                    if (p2 < sl2 / 2) {
                        i64 min2, max2;
                        gapRange(d + gaps[j], k, &min2, &max2);
                        if (min2 > minD) min2 = minD;
                        if (max2 < maxD) max2 = maxD;
                        i64 otherD = segment2[p2 * 2] - offs[j];
                        while (otherD < min2 && p2 < sl2 / 2) {
                            p2++;
                            otherD += segment2[p2 * 2] + k;
                        }
                        while (otherD < max2 && p2 < sl2 / 2) {
                            if (segment2[p2 * 2 + 1] == nextSeed) {
                                supported[i]++;
                                dist[i] += otherD;
                                break;
                            }
                            p2++;
                            otherD += segment2[p2 * 2] + k;
                        }
                    }
                }
            }
        }
        if (fCount >= (i64)ns) break;
        // :141-159
        i64 minseed = -1, mindist = 0, minsup = 0, minD = 0, maxD = 0;
        for (size_t i = 0; i < ns; i++) {
            i64 d = dist[i];
            if (supported[i] > 1) {
                d = d / supported[i];
                i64 seed = segp(i)[pos[i] * 2 + 3];
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
        if (minseed == -1) {  // :162-189
            i64 minIndex = -1, minDist = 100000;
            for (size_t i = 0; i < ns; i++) {
                i64 d = dist[i];
                if (supported[i] > 1) d = d / supported[i];
                // NB reference compares against len(segments)/2 == len(seqs)/2, not the segment length
                if (segp(i) != nullptr && pos[i] < (i64)ns / 2 && d < minDist) {
                    minDist = d;
                    minIndex = (i64)i;
                }
            }
            if (minIndex == -1) break;
            for (size_t i = 0; i < ns; i++) {
                if (segp(i) != nullptr) {
                    gaps[i] += minDist;
                    offs[i] += minDist;
                }
            }
            gaps[(size_t)minIndex] = 0;
            offs[(size_t)minIndex] = 0;
            pos[(size_t)minIndex]++;
            continue;
        }
        consensus.push_back(mindist);
        consensus.push_back(minseed);
        fCount = 0;
        for (size_t i = 0; i < ns; i++) {  // :196-250
            const i64* segment = segp(i);
            i64 sl = segn(i);
            if (segment == nullptr) {
                fCount++;
                continue;
            }
            i64 matchDex = pos[i] + 1;
            if (matchDex < sl / 2) {
                i64 min2, max2;
                gapRange(mindist + gaps[i], k, &min2, &max2);
                if (min2 > minD) min2 = minD;
                if (max2 < maxD) max2 = maxD;
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
                        matches[i]->MatchA.push_back((i64)consensus.size() / 2 - 1);
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
                    // NB p may be -1 here: the reference then reads segment[0] (p*2+2 == 0)
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
    consensus.push_back(0);  // :256
    // LoadSequence :35-42
    SeedSequence* seedCons = arena.make();
    seedCons->store = std::make_shared<std::vector<i64>>(consensus);
    seedCons->lo = 0;
    seedCons->n = consensus.size();
    seedCons->length = -k;
    for (size_t i = 0; i < consensus.size(); i += 2) seedCons->length += consensus[i] + k;
    // :258-266
    for (i64 i = (i64)matches.size() - 1; i >= 0; i--) {
        SeedMatch* m = matches[(size_t)i].get();
        if (m == nullptr || m->MatchA.size() < 3) {
            matches[(size_t)i] = std::move(matches.back());
            matches.pop_back();
        } else {
            m->SeqA = seedCons;
        }
    }
    matchesOut = std::move(matches);
    return seedCons;
}

}  // namespace dpo
