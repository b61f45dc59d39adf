// ORACLE — test infrastructure only (see oracle.hpp).  L3/L4 of the `overlap` command:
// sequence/seqio.go (FASTA rules), util/sequtil/kmers.go, overlap/overlap.go, overlap/combine.go,
// commands/overlap.go.
#include "oracle.hpp"
#include <mutex>
#include <atomic>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace dpo {

// ---------------------------------------------------------------------------------------------
// sequence/seqio.go — readFasta :106-276 (FASTA subset)

static std::string trimSpace(const std::string& s) {  // strings.TrimSpace
    size_t a = 0, b = s.size();
    auto sp = [](unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (a < b && sp((unsigned char)s[a])) a++;
    while (b > a && sp((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// One line as bufio.ReadBytes('\n') returns it (including the '\n' when present).
void FastaSet::addLine(const std::string& lastName, const std::string& line, i64 minLen, const std::string* qualLine) {
    // :209-245: a line is a sequence iff its first byte is in ['A','T']; kept iff len(line) >= minLen;
    // the stored sequence is line[:len-1] (the last byte is dropped whether or not it is '\n').
    if ((i64)line.size() >= minLen) {
        size_t id = cached.size();
        ignore.push_back(0);
        lengths.push_back((i64)line.size() - 1);
        names.push_back(trimSpace(lastName));
        auto nm = std::make_shared<std::string>(names.back());
        cached.push_back(newPackedSequence((i64)id, line.substr(0, line.size() - 1), nm));
        // :229-238 the quality line counts only when it is exactly one byte longer than the sequence (its '\n'); every
        // byte has 33 subtracted (as a Go byte: modulo 256)
        if (qualLine && qualLine->size() == line.size()) {
            auto q = std::make_shared<std::vector<uint8_t>>(line.size() - 1);
            for (size_t i = 0; i + 1 < line.size(); i++) (*q)[i] = (uint8_t)((unsigned char)(*qualLine)[i] - 33);
            cached.back().qual = q;
            cached.back().qlo = 0;
        }
        bases += (i64)line.size() - 1;
    }
}

FastaSet FastaSet::fromFile(const std::string& path, i64 minLen, bool himem) {
    FastaSet f;
    f.himem = himem;
    std::ifstream in(path, std::ios::binary);
    if (!in) return f;  // :285-290 closed channel
    std::stringstream ss;
    ss << in.rdbuf();
    std::string all = ss.str();
    size_t pos = 0;
    auto readLine = [&](std::string& out) -> bool {  // returns false at EOF with no data
        if (pos >= all.size()) {
            out.clear();
            return false;
        }
        size_t nl = all.find('\n', pos);
        if (nl == std::string::npos) {
            out = all.substr(pos);
            pos = all.size();
        } else {
            out = all.substr(pos, nl - pos + 1);
            pos = nl + 1;
        }
        return true;
    };
    std::string line, lastName;
    // :191-204 the first line is always consumed as a name/comment line; '@' makes the file a FASTQ
    if (!readLine(line)) return f;
    if (line.back() != '\n') return f;  // ReadBytes error (EOF before delimiter) => nothing read
    if (line[0] == '@') f.isFastq = true;
    lastName = line.substr(1);
    while (readLine(line)) {  // :208-267
        bool eof = line.back() != '\n';
        unsigned char c = (unsigned char)line[0];
        if (c >= 'A' && c <= 'T') {
            if (f.isFastq) {  // :222-238 (kept) / :246-255 (skipped): the '+' line and the quality line follow
                std::string plus, qual;
                const bool gotPlus = readLine(plus);
                if (!gotPlus || plus.back() != '\n' || plus[0] != '+') {
                    // err != nil || buf[0] != plus -> log.Fatal
                    f.error = "Invalid fastq format (on + line):" + plus;
                    return f;
                }
                const bool gotQual = readLine(qual);  // (its error is discarded)
                f.addLine(lastName, line, minLen, gotQual ? &qual : nullptr);
                eof = false;  // `err` is the '+' line's from here on (:224 reassigns it), and that one ended with '\n'
            } else {
                f.addLine(lastName, line, minLen);
            }
        } else if (c == '@') {
            f.isFastq = true;
            lastName = line.substr(1);
        } else {
            lastName = line.substr(1);
        }
        if (eof) break;
    }
    return f;
}

FastaSet FastaSet::fromReads(const std::vector<std::string>& names, const std::vector<std::string>& seqs, i64 minLen,
                             bool himem, const std::vector<std::string>* quals) {
    FastaSet f;
    f.himem = himem;
    for (size_t i = 0; i < seqs.size(); i++) {
        if (quals) {
            const std::string q = (*quals)[i] + "\n";
            f.addLine(names[i] + "\n", seqs[i] + "\n", minLen, &q);
        } else {
            f.addLine(names[i] + "\n", seqs[i] + "\n", minLen);
        }
    }
    return f;
}

PackedSeq FastaSet::served(size_t id) const {
    const PackedSeq& c = cached[id];
    if (himem) return c.subSequence(0, c.length);  // :115 (frontTrim = backTrim = 0)
    return c;                                      // :158 re-read as a fresh top-level sequence
}

// ---------------------------------------------------------------------------------------------
// util/sequtil/kmers.go

std::vector<u64> kmerOccurrences(const std::vector<PackedSeq>& seqs, int k) {  // :34-69
    i64 mask = ((i64)1 << (2 * k)) - 1;
    std::vector<u64> counts((size_t)1 << (2 * k), 0);
    for (const auto& seq : seqs) {
        i64 kmer = seq.kmerAt(0, k);
        counts[(size_t)kmer]++;
        for (i64 i = k; i < seq.length; i++) {
            kmer = seq.nextKmer(kmer, mask, i);
            counts[(size_t)kmer]++;
        }
    }
    return counts;
}

// commands/overlap.go:55-93 (== commands/map.go:46-71) with TopOccurrences (kmers.go:87-112).
std::vector<double> kmerValues(std::vector<u64>& counts, int k) {
    size_t n = counts.size();
    std::vector<double> values(n, 0.0);
    u64 tot = 0;
    for (u64 c : counts) tot += c;
    double tf = (double)tot;
    const double targetFreq = 0.000005;
    for (size_t i = 0; i < n; i++) {
        u64 count = counts[i];
        double freq = (double)count / tf;
        if (count < 3) values[i] = 0;
        else if (freq <= targetFreq) values[i] = 1.0 - (targetFreq - freq);
        else values[i] = 1.0 - (freq - targetFreq);
    }
    // TopOccurrences(counts, k, len/100, len/50): merge fwd+rc in place (:90-96; palindromes and
    // later-visited partners end up with re-added sums exactly as the sequential loop produces)
    for (size_t i = 0; i < n; i++) {
        size_t rc = (size_t)reverseComplementKmer((u64)i, k);
        u64 c = counts[i] + counts[rc];
        counts[i] = c;
        counts[rc] = c;
    }
    // sort ascending by value (canonical tie rule: by k-mer id, see oracle.hpp); top = last topN ids.
    size_t topN = n / 100;
    if (topN > 0) {
        // threshold select instead of a 4^k-element sort: T = value of the (n-topN)-th element
        std::vector<u64> tmp(counts);
        std::nth_element(tmp.begin(), tmp.begin() + (n - topN), tmp.end());
        u64 T = tmp[n - topN];
        size_t above = 0;
        for (u64 c : counts)
            if (c > T) above++;
        size_t needTies = topN - above;  // taken from the highest ids among count == T
        for (size_t i = n; i-- > 0;) {
            if (counts[i] > T) values[i] = 0;
            else if (counts[i] == T && needTies > 0) {
                values[i] = 0;
                needTies--;
            }
        }
    }
    values[0] = 0;
    return values;
}

// ---------------------------------------------------------------------------------------------
// overlap/overlap.go

// PrepareQueries :157-214 with getEdges :55-89 / getCentres :91-117 / getAll :119-155 and addWeighted :45-53, canonical
// synchronous order (every window's AddSeeds completes before the next budget test).
std::vector<SeedQuery> Overlapper::prepareQueries(i64 numSeeds, i64 seedLimit, const double* values,
                                                  const std::vector<PackedSeq>& seqs, int queryType) {
    const bool weightSides = (queryType & 8) != 0;
    if (weightSides) numSeeds /= 2;  // :161-163
    auto feed = [&](const PackedSeq& sub) {  // what reaches AddSeedsWorker for one query window
        const i64 sideSize = 200;
        if (weightSides && sub.length > 400) {
            index.addSeeds(sub.subSequence(0, sideSize), numSeeds, values);
            index.addSeeds(sub.subSequence(sub.length - sideSize, sub.length), numSeeds, values);
        } else {
            index.addSeeds(sub, numSeeds, values);
        }
    };
    std::vector<PackedSeq> cached;
    for (const auto& s : seqs) {
        if (index.size >= seedLimit) break;
        if (queryType & 1) {  // QueryEdges
            if (s.length < overlap * 2) {
                feed(s);
                cached.push_back(s);
            } else {
                PackedSeq s1 = s.subSequence(0, overlap);
                PackedSeq s2 = s.subSequence(s.length - overlap, s.length);
                feed(s1);
                feed(s2);
                cached.push_back(s1);
                cached.push_back(s2);
            }
        } else if (queryType & 2) {  // QueryCentre
            i64 start = (s.length - overlap) / 2;
            if (start < 0) start = 0;
            i64 end = start + overlap;
            if (end >= s.length) end = s.length - 1;
            PackedSeq centre = s.subSequence(start, end);
            feed(centre);
            cached.push_back(centre);
        } else {  // QueryAll
            if (s.length < overlap * 2) {
                feed(s);
                cached.push_back(s);
            } else {
                const i64 slices = s.length / overlap;
                for (i64 i = 0; i < slices; i++) {
                    const i64 start = (i * s.length) / slices;
                    i64 end = ((i + 1) * s.length) / slices;
                    if (i == slices - 1) end = s.length;
                    PackedSeq sub = s.subSequence(start, end);
                    feed(sub);
                    cached.push_back(sub);
                }
            }
        }
    }
    std::vector<SeedQuery> queries;
    i64 queryID = 0;
    int k = index.seedSize;
    for (const auto& s : cached) {
        SeedSequence* ss = index.newSeedSequence(s);
        SeedQuery q{queryID, ss->id, ss, true, false};
        queries.push_back(q);
        SeedQuery rc{queryID, q.SequenceID, ssReverseComplement(ss, k, index), true, true};
        queryID++;
        queries.push_back(rc);
    }
    return queries;
}

// chunkWorker :253-318 (body for one SeedSequence); every piece it would hand to AddSequence goes to `add` (the index's, or a test's)
void chunkPieces(Arena& ar, SeedSequence* s, i64 chunkSize, i64 overlap, i64 minSeeds, int k, const std::function<void(SeedSequence*)>& add);
void Overlapper::chunkAndAdd(SeedSequence* s) {
    chunkPieces(index.arena, s, chunkSize, overlap, minSeeds, index.seedSize, [&](SeedSequence* piece) { index.addSequence(piece); });
}
void chunkPieces(Arena& ar, SeedSequence* s, i64 chunkSize, i64 overlap, i64 minSeeds, int k, const std::function<void(SeedSequence*)>& add) {
    i64 numChunks = s->length / chunkSize + 1;
    if (numChunks == 1 || s->numSeeds() < minSeeds * 3) {
        if (s->numSeeds() >= minSeeds) add(s);
        return;
    }
    i64 prevSeedIndex = 0;
    i64 totalOffset = s->getSeedOffset(0, k);
    i64 lengthInBases = 0;
    for (;;) {
        i64 seedCount = 0;
        if (prevSeedIndex >= s->numSeeds() - 150) {
            if (prevSeedIndex == 0) {
                add(s);
            } else {
                i64 newFirstGap = s->getNextSeedOffset(prevSeedIndex - 1, k) - k;
                lengthInBases += s->getSeedOffsetFromEnd(prevSeedIndex, k) + k + newFirstGap;
                add(ssSubSequence(ar, s, prevSeedIndex, s->numSeeds() - 1, lengthInBases,
                                                totalOffset - newFirstGap, 0));
            }
            break;
        }
        for (; lengthInBases < chunkSize && seedCount < 100 && prevSeedIndex + seedCount < s->numSeeds(); seedCount++)
            lengthInBases += s->getNextSeedOffset(prevSeedIndex + seedCount, k);
        if (seedCount >= minSeeds) {
            i64 newFirstGap = s->getNextSeedOffset(prevSeedIndex - 1, k) - k;
            lengthInBases += newFirstGap;
            add(ssSubSequence(ar, s, prevSeedIndex, prevSeedIndex + seedCount - 1, lengthInBases,
                                            totalOffset - newFirstGap,
                                            s->length - totalOffset - lengthInBases + newFirstGap));
            totalOffset += lengthInBases - newFirstGap;
            lengthInBases = 0;
            prevSeedIndex += seedCount;
            if (prevSeedIndex >= s->numSeeds()) break;
            for (seedCount = 0; seedCount < 5 && lengthInBases < overlap / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                i64 step = s->getNextSeedOffset(prevSeedIndex, k);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        } else {
            prevSeedIndex += seedCount;
            for (seedCount = 0; lengthInBases < overlap / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                i64 step = s->getNextSeedOffset(prevSeedIndex, k);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        }
    }
}

// AddSequences :217-250
void Overlapper::addSequences(const std::vector<PackedSeq>& seqs) {
    // DPO_SCAN_THREADS=n (bench's all-cores CPU baseline only): the per-read scans run on n threads, everything else
    // stays sequential and in file order, so the output is unchanged
    const char* te = getenv("DPO_SCAN_THREADS");
    const int nt = te ? atoi(te) : 1;
    if (nt > 1 && seqs.size() > 1) {
        std::vector<std::shared_ptr<std::vector<i64>>> stores(seqs.size());
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        std::string failure;
        std::mutex fmu;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&] {
                try {
                    for (;;) {
                        const size_t i0 = next.fetch_add(64);
                        if (i0 >= seqs.size()) break;
                        for (size_t i = i0; i < std::min(seqs.size(), i0 + 64); i++) stores[i] = index.scanSegments(seqs[i]);
                    }
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk(fmu);
                    failure = e.what();
                }
            });
        for (auto& x : th) x.join();
        if (!failure.empty()) throw std::runtime_error(failure);
        for (size_t i = 0; i < seqs.size(); i++) chunkAndAdd(index.newSeedSequence(seqs[i], stores[i]));
    } else {
        for (const auto& s : seqs) chunkAndAdd(index.newSeedSequence(s));
    }
    index.indexSequences();
}

// FindOverlaps :320-340 + matchWorker :346-387
std::vector<std::unique_ptr<SeedMatch>> Overlapper::findOverlaps(const std::vector<SeedQuery>& queries) {
    std::vector<std::unique_ptr<SeedMatch>> output;
    int k = index.seedSize;
    IntSet seedSet;
    SeedAligner aligner(overlap / 2);
    for (const auto& q : queries) {
        seedSet.clear();
        for (i64 i = 0; i < q.Query->numSeeds(); i++) seedSet.add((u64)q.Query->getSeed(i));
        std::vector<u64> matches = index.matches(q.Query, hitFraction);
        i64 minMatches = (i64)(hitFraction * (double)q.Query->numSeeds() + 0.5);
        for (u64 match : matches) {
            const IntSet& matchSet = index.seedSets[(size_t)match];
            if (matchSet.countIntersectionTo(seedSet, minMatches) < (u64)minMatches) continue;
            SeedSequence* m = index.sequences[(size_t)match];
            std::vector<SeedMatch> sMatches = aligner.pairwiseAlignments(q.Query, m, seedSet, matchSet, minMatches, k);
            if (!sMatches.empty()) {
                SeedMatch* best = nullptr;
                i64 bestCount = 0;  // never updated in the reference (:369-375)
                for (auto& sm : sMatches) {
                    i64 ca, c;
                    if (!smGetBasesCovered(sm, k, &ca, &c)) throw std::runtime_error("oracle: matchWorker GetBasesCovered (reference would panic)");
                    if (c > bestCount) best = &sm;
                }
                if (best == nullptr) throw std::runtime_error("oracle: matchWorker best==nil (reference would panic)");
                best->QueryID = q.ID;
                best->ReverseComplementQuery = q.ReverseComplement;
                output.emplace_back(new SeedMatch(std::move(*best)));
                i64 blen = (i64)output.back()->MatchA.size();
                if (blen * 2 > minMatches * 3) minMatches = (blen * 2) / 3;
            }
        }
    }
    return output;
}

// ---------------------------------------------------------------------------------------------
// overlap/combine.go

static void trimRest(Arena& ar, std::vector<SeedMatch*>& ms, int k, i64 bestIndex, i64 backIndex, SeedSequence* consensus,
                     std::vector<SeedSequence*>& parts, std::vector<uint8_t>& cantTrim, i64* badBack);
// trimToBestSeed :21-111
static void trimToBestSeed(Arena& ar, i64 upto, std::vector<SeedMatch*>& ms, i64 minMatch, int k,
                           SeedSequence** consensusOut, std::vector<SeedSequence*>& parts,
                           std::vector<uint8_t>& cantTrim, i64* badBack) {
    parts.assign(ms.size(), nullptr);
    cantTrim.assign(ms.size(), 0);
    i64 bestIndex = 0, backIndex = 0;
    trimBestIndices(upto, ms, minMatch, ms[0]->SeqA->numSeeds(), &bestIndex, &backIndex);
    SeedSequence* consensus = ssTrimmed(ar, ms[0]->SeqA, 0, bestIndex, 0, backIndex, k, nullptr);
    trimRest(ar, ms, k, bestIndex, backIndex, consensus, parts, cantTrim, badBack);
    *consensusOut = consensus;
}
// step 1 of trimToBestSeed, :24-58: the best front and back seeds
void trimBestIndices(i64 upto, const std::vector<SeedMatch*>& ms, i64 minMatch, i64 length, i64* bestOut, i64* backOut) {
    i64 bestCount = 0, bestScore = 0, bestIndex = upto, backCount = 0, backScore = 0;
    i64 backIndex = length - upto - 1;
    for (i64 i = 0; i < upto; i++) {
        i64 count = 0, bCount = 0;
        for (SeedMatch* match : ms) {
            for (i64 index : match->MatchA) {
                if (index == i) count++;
                if (index >= i) break;
            }
            for (i64 j = (i64)match->MatchA.size() - 1; j > 0; j--) {
                i64 index = match->MatchA[(size_t)j];
                if (index == length - 1 - i) bCount++;
                if (index <= length - 1 - i) break;
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
// step 2, :59-110
static void trimRest(Arena& ar, std::vector<SeedMatch*>& ms, int k, i64 bestIndex, i64 backIndex, SeedSequence* consensus,
                     std::vector<SeedSequence*>& parts, std::vector<uint8_t>& cantTrim, i64* badBack) {
    for (size_t j = 0; j < ms.size(); j++) {
        SeedMatch* match = ms[j];
        i64 index, bases, frontDistance, bIndex, backBases, backDistance;
        smGetBaseIndex(*match, bestIndex, k, &index, &bases, &frontDistance);
        smGetBaseIndex(*match, backIndex, k, &bIndex, &backBases, &backDistance);
        cantTrim[j] = frontDistance > 50 || frontDistance < -50 || backDistance > 50 || backDistance < -50;
        if (bases > -k && index < match->SeqB->numSeeds() - 1) {
            bases = match->SeqB->getNextSeedOffset(index, k) - bases;
            index++;
        } else if (bases < 0) {
            bases = -bases + k;
        }
        parts[j] = ssTrimmed(ar, match->SeqB, bases, index, backBases, bIndex, k, nullptr);
        SeedSequence* oldCons = match->SeqA;
        match->SeqB = parts[j];
        match->SeqA = consensus;
        i64 front = 0;
        while (front < (i64)match->MatchB.size() && match->MatchB[(size_t)front] < index) front++;
        i64 back = (i64)match->MatchB.size() - 1;
        while (back >= 0 && match->MatchB[(size_t)back] > bIndex) back--;
        if (front < 0 || back + 1 > (i64)match->MatchA.size() || back < front) {
            // :93-102 — the reference prints a "Bad back:" diagnostic into STDOUT (it formats *SeedSequence
            // pointers, so the text is not reproducible) and, when back+1 < front, panics on the slice
            // expression.  Canonical: no text; the match becomes empty; the event is counted.
            if (badBack) (*badBack)++;
            (void)oldCons;
            if (back + 1 < front) back = front - 1;
        }
        std::vector<i64> na(match->MatchA.begin() + front, match->MatchA.begin() + back + 1);
        std::vector<i64> nb(match->MatchB.begin() + front, match->MatchB.begin() + back + 1);
        match->MatchA = std::move(na);
        match->MatchB = std::move(nb);
        for (size_t n = 0; n < match->MatchB.size(); n++) {
            i64 oldIndex = match->MatchB[n];
            match->MatchA[n] -= bestIndex;
            match->MatchB[n] = oldIndex - index;
        }
    }
}

// NewSeedContig :113-133
static std::unique_ptr<SeedContig> newSeedContig(Arena& ar, std::vector<SeedMatch*>& ms, int k, i64* badBack) {
    i64 minMatch = 5;
    if (ms.size() < 5) minMatch = (i64)ms.size();
    SeedSequence* consensus;
    std::vector<SeedSequence*> parts;
    std::vector<uint8_t> trimFailed;
    trimToBestSeed(ar, ms[0]->SeqA->numSeeds() / 4, ms, minMatch, k, &consensus, parts, trimFailed, badBack);
    std::unique_ptr<SeedContig> c(new SeedContig());
    c->Combined = consensus;
    size_t n = ms.size();
    c->Parts.assign(n, 0);
    c->ReverseComplement.assign(n, 0);
    c->Offsets.assign(n, 0);
    c->Lengths.assign(n, 0);
    c->Approximate = trimFailed;
    c->SeqLengths.assign(n, 0);
    c->Matches = ms;
    for (size_t i = 0; i < n; i++) {
        SeedSequence* part = parts[i];
        c->Parts[i] = part->id;
        c->ReverseComplement[i] = part->rc;
        SeedSequence* parent = part;
        while (parent->Parent != nullptr) parent = parent->Parent;
        c->SeqLengths[i] = parent->length;
        c->Offsets[i] = part->offset;
        c->Lengths[i] = parent->length - part->offset - part->inset;
    }
    return c;
}

// BuildConsensus :163-193
std::unique_ptr<SeedContig> buildConsensus(SeedIndex& sg, std::vector<SeedMatch*>& overlaps, i64* badBack) {
    int k = sg.seedSize;
    Arena& ar = sg.arena;
    std::vector<SeedSequence*> seqs;
    for (SeedMatch* lap : overlaps)
        if (lap->ReverseComplementQuery) smReverseComplement(*lap, k, sg);
    for (SeedMatch* lap : overlaps) {
        SeedSequence* s = lap->SeqB;
        i64 ca, cb;
        if (!smGetBasesCovered(*lap, k, &ca, &cb)) throw std::runtime_error("oracle: BuildConsensus GetBasesCovered (reference would panic)");
        if (ca < 25 || cb < 25) continue;
        s = ssTrimmed(ar, s, overlaps[0]->SeqA->getSeedOffset(lap->MatchA[0], k), lap->MatchB[0],
                      overlaps[0]->SeqA->getSeedOffsetFromEnd(lap->MatchA.back(), k), lap->MatchB.back(), k, nullptr);
        seqs.push_back(s);
    }
    if (seqs.size() > 1) {
        std::vector<std::unique_ptr<SeedMatch>> overlap;
        multiAlignerConsensus(ar, seqs, k, overlap);
        if (overlap.size() > 1) {
            std::vector<SeedMatch*> ms;
            for (auto& m : overlap) ms.push_back(m.get());
            std::unique_ptr<SeedContig> c = newSeedContig(ar, ms, k, badBack);
            c->owned = std::move(overlap);
            return c;
        }
    }
    return nullptr;
}

// ---------------------------------------------------------------------------------------------
// commands/overlap.go:96-233 (Run + finalCheckWorker), canonical single-worker order.

OverlapResult runOverlap(FastaSet& set, const OverlapParams& p, const double* valuesOrNull, i64 maxRounds,
                         bool keepTraces) {
    OverlapResult res;
    const int k = p.k;
    std::vector<double> ownValues;
    const double* values = valuesOrNull;
    char line[256];
    snprintf(line, sizeof line, "Counting all %d-mers in the input...\n", k);
    res.err += line;
    if (!values) {
        std::vector<u64> counts = kmerOccurrences(set.cached, k);  // first pass: top-level sequences (:41)
        ownValues = kmerValues(counts, k);
        values = ownValues.data();
    }
    res.err += "Counting complete. Starting indexing and querying...";
    i64 firstSequence = 0;
    for (i64 round = 0;; round++) {
        if (maxRounds >= 0 && round >= maxRounds) break;
        SeedIndex seedIndex(k);
        Overlapper lap(seedIndex, p.chunkSize, p.overlapSize, p.numSeeds, p.minHits);
        // GetNSequencesFrom(firstSequence, queryBatchSize) (seqio.go:278): non-ignored ids >= firstSequence
        std::vector<PackedSeq> qseqs;
        if (!(firstSequence != 0 && firstSequence >= (i64)set.size())) {
            for (size_t id = (size_t)firstSequence; id < set.size() && (i64)qseqs.size() < p.queryBatchSize; id++)
                if (!set.ignore[id]) qseqs.push_back(set.served(id));
        }
        std::vector<SeedQuery> queries = lap.prepareQueries(p.numSeeds, p.seedBatchSize, values, qseqs, p.queryType);
        if (queries.empty()) break;
        i64 numQuerySeqs = 0;
        firstSequence = queries.back().SequenceID + 1;
        for (auto& q : queries) {
            if (q.ID >= numQuerySeqs) numQuerySeqs = q.ID + 1;
            if (q.SequenceID >= firstSequence) firstSequence = q.SequenceID + 1;
        }
        std::vector<PackedSeq> all;
        for (size_t id = 0; id < set.size(); id++)
            if (!set.ignore[id]) all.push_back(set.served(id));
        lap.addSequences(all);
        if (round == 0)
            snprintf(line, sizeof line, "Using query sets of around %lld sequences against %lld sequences.\n",
                     (long long)firstSequence, (long long)set.size());
        else
            snprintf(line, sizeof line, "Using query set with %lld  sequences starting from %lld sequences against %lld sequences.\n",
                     (long long)numQuerySeqs, (long long)firstSequence, (long long)set.size());
        res.err += line;

        RoundTrace tr;
        if (keepTraces) {
            tr.seedKmers = seedIndex.seedMap;
            tr.firstSequence = firstSequence;
            tr.numQuerySeqs = numQuerySeqs;
            for (auto& q : queries) {
                tr.querySegments.emplace_back(q.Query->seg(), q.Query->seg() + q.Query->n);
                tr.queryIDs.push_back(q.ID);
                tr.querySeqIDs.push_back(q.SequenceID);
                tr.queryLength.push_back(q.Query->length);
                tr.queryOffset.push_back(q.Query->offset);
                tr.queryInset.push_back(q.Query->inset);
                tr.candidates.push_back(seedIndex.matches(q.Query, p.minHits));
            }
            for (auto* s : seedIndex.sequences) {
                tr.indexedSegments.emplace_back(s->seg(), s->seg() + s->n);
                tr.indexedIds.push_back(s->id);
                tr.indexedLength.push_back(s->length);
                tr.indexedOffset.push_back(s->offset);
                tr.indexedInset.push_back(s->inset);
            }
        }

        std::vector<std::unique_ptr<SeedMatch>> matches = lap.findOverlaps(queries);
        if (keepTraces) {
            for (auto& m : matches) {
                // recover the query index: fwd = 2*ID, rc = 2*ID+1
                tr.matchQueryIndex.push_back(m->QueryID * 2 + (m->ReverseComplementQuery ? 1 : 0));
                i64 tgt = -1;
                for (size_t t = 0; t < seedIndex.sequences.size(); t++)
                    if (seedIndex.sequences[t] == m->SeqB) {
                        tgt = (i64)t;
                        break;
                    }
                tr.matchTarget.push_back(tgt);
                tr.matchA.push_back(m->MatchA);
                tr.matchB.push_back(m->MatchB);
            }
        }
        std::vector<std::vector<SeedMatch*>> queryResults((size_t)numQuerySeqs);
        i64 hits = 0, qHits = 0;
        for (auto& m : matches) {
            hits++;
            auto& qr = queryResults[(size_t)m->QueryID];
            if (qr.size() == 1) qHits++;
            qr.push_back(m.get());
        }
        snprintf(line, sizeof line, "Total %lld hits across %lld overlaps.\n", (long long)hits, (long long)qHits);
        res.err += line;
        std::string roundPaf;
        std::vector<i64> newlyIgnored;
        auto setIgnore = [&](i64 id) {
            if (!set.ignore[(size_t)id]) newlyIgnored.push_back(id);
            set.ignore[(size_t)id] = 1;
        };
        for (auto& results : queryResults) {  // finalCheckWorker :197-233
            if (results.size() <= 1) continue;
            std::unique_ptr<SeedContig> contig = buildConsensus(seedIndex, results, &res.badBack);
            if (contig && contig->Parts.size() > 1) {
                if (contig->SeqLengths[0] <= p.overlapSize * 2) setIgnore(contig->Parts[0]);
                i64 queryStart = contig->Offsets[0];
                i64 queryEnd = queryStart + contig->Lengths[0];
                for (size_t i = 0; i + 1 < contig->Parts.size(); i++) {
                    size_t id = i + 1;
                    i64 part = contig->Parts[id];
                    const char* rc = "+";
                    i64 start = contig->Offsets[id];
                    i64 end = start + contig->Lengths[id];
                    if (contig->ReverseComplement[0] != contig->ReverseComplement[id]) rc = "-";
                    i64 covered = p.overlapSize;
                    if (end - start > p.overlapSize) covered = end - start;
                    if (contig->SeqLengths[id] * 9 <= covered * 10) setIgnore(part);
                    i64 ident, identB;
                    if (!smGetBasesCovered(*contig->Matches[i], k, &ident, &identB)) {
                        // the reference panics here (index out of range); canonical: ident 0
                        ident = 0;
                        res.emptyMatchPanics++;
                    }
                    std::string s = set.names[(size_t)contig->Parts[0]] + "\t" + std::to_string(contig->SeqLengths[0]) + "\t" +
                                    std::to_string(queryStart) + "\t" + std::to_string(queryEnd) + "\t" + rc + "\t" +
                                    set.names[(size_t)part] + "\t" + std::to_string(contig->SeqLengths[id]) + "\t" +
                                    std::to_string(start) + "\t" + std::to_string(end) + "\t" + std::to_string(ident) +
                                    "\t0\t255\n";
                    roundPaf += s;
                }
            }
        }
        res.paf += roundPaf;
        res.rounds = round + 1;
        if (keepTraces) {
            tr.hits = hits;
            tr.qHits = qHits;
            tr.paf = roundPaf;
            tr.newlyIgnored = newlyIgnored;
            res.traces.push_back(std::move(tr));
        }
    }
    snprintf(line, sizeof line, "[oracle] rounds=%lld bad_back_suppressed=%lld empty_match_panics_avoided=%lld\n",
             (long long)res.rounds, (long long)res.badBack, (long long)res.emptyMatchPanics);
    res.err += line;
    return res;
}

}  // namespace dpo
