// ORACLE — test infrastructure only (see oracle.hpp).  `map` command:
// mapping/mapping.go, commands/map.go.
#include "oracle.hpp"

#include <algorithm>
#include <cstdio>
#include <stdexcept>

namespace dpo {

Mapping* Mapper::mk() {
    pool.emplace_back(new Mapping());
    return pool.back().get();
}

// NewMapper mapping.go:67-109 (canonical: chunks indexed in generation order)
Mapper::Mapper(const PackedSeq& ref, bool circ, int k, const double* values, i64 seedRate, i64 edge, i64 chunkSize)
    : index(k), reference(ref), edgeSize(edge), circular(circ) {
    index.addSingleSeeds(reference, seedRate, values);
    i64 ind = 0;
    auto addChunk = [&](const PackedSeq& c) {
        SeedSequence* s = index.newSeedSequence(c);
        s->id = ind;
        index.addSequence(s);
        ind++;
    };
    for (i64 j = 0; j < 10; j++) {
        i64 start = j * chunkSize;
        i64 step = chunkSize * 10 - edgeSize;
        for (i64 i = start; i < reference.length - chunkSize / 2; i += step) {
            i64 end = i + chunkSize;
            if (i >= reference.length) end = reference.length;
            addChunk(reference.subSequence(i, end));
        }
    }
    if (circular)
        addChunk(reference.subSequence(reference.length - edgeSize, reference.length)
                     .append(0, reference.subSequence(0, edgeSize)));
    index.indexSequences();
}

// AsString :112-122
std::string Mapper::asString(const Mapping& m) const {
    const char* rc = m.RC ? "-" : "+";
    i64 mappedLength = m.End - m.Start;
    if (circular && mappedLength < 0) mappedLength = reference.length - m.Start + m.End;
    std::string s = m.Query->getName() + "\t" + std::to_string(m.Query->length) + "\t" + std::to_string(m.QueryOffset) + "\t" +
                    std::to_string(m.Query->length - m.QueryInset) + "\t" + rc + "\t" + reference.getName() + "\t" +
                    std::to_string(reference.length) + "\t" + std::to_string(m.Start) + "\t" + std::to_string(m.End) + "\t" +
                    std::to_string(m.ids) + "\t" + std::to_string(mappedLength) + "\t255";
    return s;
}

// isConsistent :131-160
bool Mapper::isConsistent(const Mapping* left, const Mapping* right) const {
    return mappingsConsistent(left, left->Query->length, right, circular, reference.length);
}
bool mappingsConsistent(const Mapping* left, i64 leftQueryLen, const Mapping* right, bool circular, i64 referenceLength) {
    if (left->RC != right->RC) return false;
    i64 expectedDistance = right->QueryOffset - leftQueryLen + left->QueryInset;
    i64 distance;
    if (!left->RC) distance = right->Start - left->End;
    else distance = left->Start - right->End;
    if (circular && distance < -50) distance += referenceLength;
    if (distance < 50 && expectedDistance < 50 && distance > -50) return true;
    if (distance < 500) return (expectedDistance < (distance * 3) / 2 && expectedDistance > (distance * 2) / 3);
    if (distance > 5000) return (expectedDistance < (distance * 10) / 9 && expectedDistance > (distance * 9) / 10);
    double ratio = (double)(distance - 500) / 4500.0;
    ratio = 3.0 / 2.0 + ratio * (10.0 / 9.0 - 3.0 / 2.0);
    return distance < (i64)((double)expectedDistance * ratio) && distance > (i64)((double)expectedDistance / ratio);
}

// removeDominated :387-428
std::vector<Mapping*> removeDominated(std::vector<Mapping*> open, const std::vector<Mapping*>* extendedIn,
                                             i64 queryLen) {
    // every call site passes the same slice for open and extended: sorting open reorders extended too.
    if (open.empty()) return open;
    goSort(open, [](Mapping* a, Mapping* b) { return a->QueryOffset < b->QueryOffset; });
    const std::vector<Mapping*>& extended = extendedIn ? *extendedIn : open;
    size_t j = 0;
    std::vector<uint8_t> toRemove(open.size(), 0);
    for (size_t i = 0; i < open.size(); i++) {
        Mapping* next = open[i];
        while (j < extended.size() && queryLen - extended[j]->QueryInset < next->QueryOffset) j++;
        if (j == extended.size()) return open;
        bool dominated = false;
        for (size_t kk = j; !dominated && kk < extended.size() && extended[kk]->QueryOffset < queryLen - next->QueryInset; kk++) {
            if (extended[kk]->ids * 4 > next->ids * 5) {
                i64 start = next->QueryOffset;
                if (extended[kk]->QueryOffset > start) start = extended[kk]->QueryOffset;
                i64 end = queryLen - next->QueryInset;
                if (extended[kk]->QueryInset > next->QueryInset) end = queryLen - extended[kk]->QueryInset;
                dominated = ((end - start) * 10 > (queryLen - next->QueryOffset - next->QueryInset) * 9);
            }
        }
        toRemove[i] = dominated;
    }
    i64 last = (i64)open.size() - 1;
    for (i64 i = last; i >= 0; i--) {
        if (toRemove[(size_t)i]) {
            open[(size_t)i] = open[(size_t)last];
            last--;
        }
    }
    open.resize((size_t)(last + 1));
    return open;
}

// matchPairs :174-203
void Mapper::matchPairs(std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& matched,
                        bool& matchedNil) {
    matched.clear();
    matchedNil = true;
    for (i64 i = (i64)openA.size() - 1; i >= 0; i--) {
        Mapping* ra = openA[(size_t)i];
        for (i64 j = (i64)openB.size() - 1; j >= 0; j--) {
            Mapping* rb = openB[(size_t)j];
            if (isConsistent(ra, rb)) {
                i64 qOffset = ra->QueryOffset;
                i64 qInset = rb->QueryInset;
                if (ra->RC) std::swap(ra, rb);
                Mapping* combined = mk();
                combined->Start = ra->Start;
                combined->End = rb->End;
                combined->Query = ra->Query;
                combined->QueryOffset = qOffset;
                combined->QueryInset = qInset;
                combined->RC = ra->RC;
                combined->ids = ra->ids + rb->ids;
                matchedNil = false;
                matched.push_back(combined);
                openA[(size_t)i] = openA.back();
                openA.pop_back();
                openB[(size_t)j] = openB.back();
                openB.pop_back();
                break;
            }
        }
    }
}

// findSplitPoint :207-288
void Mapper::findSplitPoint(const PackedSeq& query, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB,
                            i64 left, i64 right) {
    while (right - left >= edgeSize) {
        i64 start = (right + left - edgeSize) / 2;
        i64 end = start + edgeSize;
        std::vector<Mapping*> mid = performMapping(query.subSequence(start, end));
        i64 newLeft = left, newRight = right, afterA = 0, afterB = 0;
        for (Mapping* mm : mid) {
            mm->Query = &query;
            for (Mapping* ma : openA) {
                if (isConsistent(ma, mm)) {
                    ma->QueryInset = mm->QueryInset;
                    ma->ids += mm->ids;
                    if (ma->RC) ma->Start = mm->Start;
                    else ma->End = mm->End;
                    i64 midMatched = query.length - mm->QueryInset - mm->QueryOffset;
                    if (midMatched > afterA) afterA = midMatched;
                    if (query.length - mm->QueryInset > newLeft) newLeft = query.length - mm->QueryInset;
                    break;
                }
            }
            if (afterA < (edgeSize * 2) / 3) {
                for (Mapping* mb : openB) {
                    if (isConsistent(mm, mb)) {
                        mb->QueryOffset = mm->QueryOffset;
                        mb->ids += mm->ids;
                        if (mb->RC) mb->End = mm->End;
                        else mb->Start = mm->Start;
                        i64 midMatched = query.length - mm->QueryInset - mm->QueryOffset;
                        if (midMatched > afterB) afterB = midMatched;
                        if (mm->QueryOffset < newRight) newRight = mm->QueryOffset;
                        break;
                    }
                }
            }
        }
        if (afterA > 0 && afterB > 0) {
            std::vector<Mapping*> empty;
            if (newLeft - left > edgeSize * 2) findSplitPoint(query, openA, empty, newLeft - edgeSize * 2, newLeft - edgeSize);
            if (right - newRight > edgeSize * 2) findSplitPoint(query, empty, openB, newRight + edgeSize, newRight + edgeSize * 2);
            return;
        }
        if (afterA == 0 && afterB == 0) {
            std::vector<Mapping*> empty;
            if (!openA.empty()) findSplitPoint(query, openA, empty, left, start);
            if (!openB.empty()) findSplitPoint(query, empty, openB, end, right);
            return;
        }
        left = newLeft;
        right = newRight;
    }
}

static void updateQuery(std::vector<Mapping*>& ms, const PackedSeq* q) {
    for (Mapping* m : ms) m->Query = q;
}
static void appendAll(std::vector<Mapping*>& dst, const std::vector<Mapping*>& src) {
    dst.insert(dst.end(), src.begin(), src.end());
}

// mapNext :305-383.  Go slices returned by matchPairs alias their inputs; every use below either
// consumes the returned (shrunk) slice or appends to it, which value semantics reproduce.
void Mapper::mapNext(const PackedSeq& query, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB,
                     std::vector<Mapping*>& newA, std::vector<Mapping*>& newB, std::vector<Mapping*>& matched,
                     bool& matchedNil) {
    std::vector<Mapping*> extended;
    bool extNil;
    if (query.length < edgeSize * 4) {
        newA = performMapping(query.subSequence(edgeSize, query.length - edgeSize));
        newA = removeDominated(newA, nullptr, query.length);
        updateQuery(newA, &query);
        matchPairs(openA, newA, extended, extNil);
        if (!extNil) {
            std::vector<Mapping*> t = newA;
            appendAll(t, extended);
            openA = t;
        } else {
            appendAll(openA, newA);
        }
        matchPairs(openA, openB, matched, matchedNil);
        newA = openA;
        newB = openB;
        if (matchedNil) return;
        newA.clear();
        newB.clear();
        return;
    }
    // 1.
    newA = performMapping(query.subSequence(edgeSize, edgeSize * 2));
    newA = removeDominated(newA, nullptr, query.length);
    updateQuery(newA, &query);
    matchPairs(openA, newA, extended, extNil);
    appendAll(openA, newA);
    if (!extNil) appendAll(openA, extended);
    newB = performMapping(query.subSequence(query.length - edgeSize * 2, query.length - edgeSize));
    newB = removeDominated(newB, nullptr, query.length);
    updateQuery(newB, &query);
    {
        // openB, newB, extended = m.matchPairs(newB, openB)
        std::vector<Mapping*> a = newB, b = openB;
        matchPairs(a, b, extended, extNil);
        openB = a;
        newB = b;
    }
    appendAll(openB, newB);
    if (!extNil) appendAll(openB, extended);
    {
        std::vector<Mapping*> a = openA, b = openB;
        matchPairs(a, b, matched, matchedNil);
        newA = a;
        newB = b;
    }
    // 2.
    if (matchedNil) {
        if (query.length > edgeSize * 5) {
            openA = performMapping(query.subSequence(edgeSize * 2, edgeSize * 3));
            openA = removeDominated(openA, nullptr, query.length);
            updateQuery(openA, &query);
            {
                // openA, newA, extended = m.matchPairs(newA, openA)
                std::vector<Mapping*> a = newA, b = openA;
                matchPairs(a, b, extended, extNil);
                openA = a;
                newA = b;
            }
            if (!extNil) appendAll(openA, extended);
            appendAll(openA, newA);
        }
        if (query.length > edgeSize * 6) {
            openB = performMapping(query.subSequence(query.length - edgeSize * 3, query.length - edgeSize * 2));
            openB = removeDominated(openB, nullptr, query.length);
            updateQuery(openB, &query);
            {
                std::vector<Mapping*> a = openB, b = newB;
                matchPairs(a, b, extended, extNil);
                openB = a;
                newB = b;
            }
            if (!extNil) appendAll(openB, extended);
            appendAll(openB, newB);
        } else {
            openB = newB;
        }
        if (query.length > edgeSize * 5) {
            std::vector<Mapping*> a = openA, b = openB;
            matchPairs(a, b, matched, matchedNil);
            newA = a;
            newB = b;
        }
    }
}

// Map :430-487
std::vector<Mapping*> Mapper::map(const PackedSeq& query) {
    std::vector<Mapping*> results;
    if (query.length <= edgeSize * 2) {
        results = performMapping(query);
        results = removeDominated(results, nullptr, query.length);
        updateQuery(results, &query);
        return results;
    }
    // mapEnds :164-172
    std::vector<Mapping*> openA = performMapping(query.subSequence(0, edgeSize));
    std::vector<Mapping*> openB = performMapping(query.subSequence(query.length - edgeSize, query.length));
    openA = removeDominated(openA, nullptr, query.length);
    openB = removeDominated(openB, nullptr, query.length);
    updateQuery(openA, &query);
    updateQuery(openB, &query);
    std::vector<Mapping*> matched;
    bool matchedNil;
    matchPairs(openA, openB, matched, matchedNil);
    if (!matchedNil) {
        results = matched;
    } else if (query.length < edgeSize * 3) {
        results = openA;
        appendAll(results, openB);
    } else {
        std::vector<Mapping*> nA, nB;
        mapNext(query, openA, openB, nA, nB, matched, matchedNil);
        openA = nA;
        openB = nB;
        if (!matchedNil) {
            results = matched;
        } else {
            i64 left = edgeSize * 2;
            i64 right = query.length - edgeSize * 2;
            for (Mapping* a : openA)
                if (a->QueryInset > left) left = a->QueryInset;
            left = query.length - right;
            for (Mapping* b : openB)
                if (b->QueryOffset < right) right = b->QueryOffset;
            findSplitPoint(query, openA, openB, left, right);
            i64 size = query.length - edgeSize;
            for (i64 i = (i64)openA.size() - 1; i >= 0; i--) {
                if (openA[(size_t)i]->QueryInset >= size) {
                    openA[(size_t)i] = openA.back();
                    openA.pop_back();
                }
            }
            for (i64 i = (i64)openB.size() - 1; i >= 0; i--) {
                if (openB[(size_t)i]->QueryOffset >= size) {
                    openB[(size_t)i] = openB.back();
                    openB.pop_back();
                }
            }
            results = openA;
            appendAll(results, openB);
        }
    }
    return results;
}

// performMapping :489-611
std::vector<Mapping*> Mapper::performMapping(const PackedSeq& query) {
    int k = index.seedSize;
    Arena& ar = index.arena;
    SeedSequence* seedQuery = index.newSeedSequence(query);
    SeedSequence* rcQuery = index.newSeedSequence(query.reverseComplement());
    i64 minMatches = seedQuery->numSeeds() / 5;
    i64 minRCMatches = rcQuery->numSeeds() / 5;
    if (minMatches < 5) minMatches = 5;
    if (minRCMatches < 5) minRCMatches = 5;
    std::vector<u64> matchingIndices = index.matches(seedQuery, 0.25);
    std::vector<u64> matchingRCIndices = index.matches(rcQuery, 0.25);
    std::vector<Mapping*> results;
    i64 maxSeed = 0;
    for (i64 i = 0; i < seedQuery->numSeeds(); i++)
        if (seedQuery->getSeed(i) > maxSeed) maxSeed = seedQuery->getSeed(i);
    IntSet seedSet(maxSeed + 1);
    for (i64 i = 0; i < seedQuery->numSeeds(); i++) seedSet.add((u64)seedQuery->getSeed(i));
    for (u64 idx : matchingIndices) {
        const IntSet& matchSet = index.seedSets[(size_t)idx];
        if (matchSet.countIntersectionTo(seedSet, minMatches) < (u64)minMatches) continue;
        SeedSequence* match = index.sequences[(size_t)idx];
        std::vector<SeedMatch> seedMatches = ssMatch(ar, match, seedQuery, &seedSet, &matchSet, minMatches, k);
        for (auto& sm : seedMatches) {
            i64 start = match->offset + match->getSeedOffset(sm.MatchB[0], k);
            i64 end = reference.length - match->inset - match->getSeedOffsetFromEnd(sm.MatchB.back(), k);
            if (circular && start > reference.length) start -= reference.length;
            i64 qOffset = seedQuery->getSeedOffset(sm.MatchA[0], k);
            i64 qInset = seedQuery->getSeedOffsetFromEnd(sm.MatchA.back(), k);
            if (qOffset + qInset > (seedQuery->length * 2) / 3) continue;
            qOffset += seedQuery->offset;
            qInset += seedQuery->inset;
            i64 ca, ids;
            if (!smGetBasesCovered(sm, k, &ca, &ids)) throw std::runtime_error("oracle: performMapping GetBasesCovered (reference would panic)");
            Mapping* mp = mk();
            mp->Start = start;
            mp->End = end;
            mp->QueryOffset = qOffset;
            mp->QueryInset = qInset;
            mp->RC = false;
            mp->ids = ids;
            results.push_back(mp);
            i64 limit = ((i64)sm.MatchA.size() * 4) / 5;
            if (limit > minMatches) minMatches = limit;
            if (limit > minRCMatches) minRCMatches = limit;
        }
    }
    seedSet.clear();
    for (i64 i = 0; i < rcQuery->numSeeds(); i++) seedSet.add((u64)rcQuery->getSeed(i));
    for (u64 idx : matchingRCIndices) {
        const IntSet& matchSet = index.seedSets[(size_t)idx];
        if (matchSet.countIntersectionTo(seedSet, minRCMatches) < (u64)minRCMatches) continue;
        SeedSequence* match = index.sequences[(size_t)idx];
        std::vector<SeedMatch> seedMatches = ssMatch(ar, match, rcQuery, &seedSet, &matchSet, minRCMatches, k);
        for (auto& sm : seedMatches) {
            i64 start = match->offset + match->getSeedOffset(sm.MatchB[0], k);
            i64 end = reference.length - match->inset - match->getSeedOffsetFromEnd(sm.MatchB.back(), k);
            if (circular && start > reference.length) start -= reference.length;
            i64 qInset = rcQuery->getSeedOffset(sm.MatchA[0], k);
            i64 qOffset = rcQuery->getSeedOffsetFromEnd(sm.MatchA.back(), k);
            if (qOffset + qInset > (rcQuery->length * 2) / 3) continue;
            qInset += rcQuery->offset;
            qOffset += rcQuery->inset;
            i64 ca, ids;
            if (!smGetBasesCovered(sm, k, &ca, &ids)) throw std::runtime_error("oracle: performMapping GetBasesCovered (reference would panic)");
            Mapping* mp = mk();
            mp->Start = start;
            mp->End = end;
            mp->QueryOffset = qOffset;
            mp->QueryInset = qInset;
            mp->RC = true;
            mp->ids = ids;
            results.push_back(mp);
            i64 limit = ((i64)sm.MatchA.size() * 4) / 5;
            if (limit > minRCMatches) minRCMatches = limit;
        }
    }
    if (results.size() > 1) {
        goSort(results, [](Mapping* a, Mapping* b) { return a->Start < b->Start; });
        for (i64 i = (i64)results.size() - 1; i > 0; i--) {
            Mapping* ra = results[(size_t)(i - 1)];
            Mapping* rb = results[(size_t)i];
            if (ra->RC == rb->RC && rb->Start < ra->End) {
                if (ra->End - ra->Start > rb->End - rb->Start) {
                    results[(size_t)i] = results.back();
                    results.pop_back();
                } else {
                    results[(size_t)(i - 1)] = results[(size_t)i];
                    results[(size_t)i] = results.back();
                    results.pop_back();
                }
            }
        }
    }
    return results;
}

// commands/map.go:33-116
MapResult runMap(FastaSet& refSet, FastaSet& reads, const MapParams& p) {
    MapResult res;
    if (refSet.size() == 0) throw std::runtime_error("oracle: empty reference");
    PackedSeq reference = refSet.cached[0];  // cache=false: top-level sequence
    std::vector<u64> counts = kmerOccurrences(refSet.cached, p.k);
    std::vector<double> values = kmerValues(counts, p.k);
    res.err += "K-mer counting complete. Preparing to start indexing and querying...\n";
    Mapper mapper(reference, p.circular, p.k, values.data(), p.seedRate, p.querySize, p.chunkSize);
    i64 unmapped = 0, mapped = 0, multiple = 0, total = 0;
    const size_t arenaBase = mapper.index.arena.seqs.size();
    for (size_t id = 0; id < reads.size(); id++) {
        const PackedSeq& q = reads.cached[id];  // cache=false: top-level sequences
        std::vector<Mapping*> maps = mapper.map(q);
        if (!maps.empty()) {
            for (Mapping* m : maps) res.paf += mapper.asString(*m) + "\n";
            if (maps.size() == 1) mapped++;
            else multiple++;
            total += (i64)maps.size();
        } else {
            unmapped++;
        }
        // per-read scratch (seed sequences of the windows) is dead once the read's PAF is printed
        mapper.index.arena.seqs.resize(arenaBase);
    }
    char line[128];
    snprintf(line, sizeof line, "Uniquely mapped: %lld\nMultiple mappings: %lld\ntotal: %lld\nUnmapped: %lld\n",
             (long long)mapped, (long long)multiple, (long long)total, (long long)unmapped);
    res.err += line;
    return res;
}

}  // namespace dpo
