// Product host side of `downpore map` (mapping/mapping.go, commands/map.go:33-116) above the C ABI.
//
// mapping.Mapper.Map is adaptive per read (ends -> one/two steps in -> binary split search), every step being a
// performMapping of one window.  To batch windows of MANY reads into each GPU call while keeping Map()'s control flow
// written exactly as the reference's sequential code, each read's Map() runs as a stackful coroutine (host_coro.hpp):
// performMapping() files a window request and yields; when every live coroutine is waiting, the scheduler scans all
// requested windows (dp_scan), runs the candidate/chaining stage (dp_map_windows) and resumes the coroutines with
// their chains.  Results are emitted in read order (the canonical single-worker order).
#include <algorithm>
#include <cstdio>
#include <cstring>

#include <chrono>

#include "dph.hpp"
#include "host_util.hpp"

#define DPH_CORO_IMPLEMENTATION  // (the switch routine itself lives in this translation unit)
#include "host_coro.hpp"

namespace dph {

namespace {

struct Mapping {  // mapping/mapping.go:11-20
    i64 queryLen = 0;  // Query.Len() of the sequence updateQuery() attached (the whole read)
    bool hasQuery = false;
    i64 Start = 0, End = 0, QueryOffset = 0, QueryInset = 0;
    bool RC = false;
    i64 ids = 0;
};

// Go sort.Sort stand-in (DESIGN.md canonical semantics): insertion sort <= 12 elements, stable sort above.
template <class T, class Less>
void goSort(std::vector<T>& v, Less less) {
    if (v.size() <= 12) {
        for (size_t i = 1; i < v.size(); i++)
            for (size_t j = i; j > 0 && less(v[j], v[j - 1]); j--) std::swap(v[j], v[j - 1]);
    } else {
        std::stable_sort(v.begin(), v.end(), less);
    }
}

struct Chunk {  // an indexed reference chunk (seeds.SeedSequence of a reference.SubSequence)
    const int32_t* seg = nullptr;
    int n = 0;
    i64 offset = 0, inset = 0;
};

struct WindowReq {
    uint32_t read = 0;  // index into the read set
    i64 a = 0, b = 0;   // view [a, b) of the top-level read
    bool whole = false; // the top-level read itself (len <= 2*edge)
};

struct WindowResult {
    // forward / reverse-complement seed sequences of the window and the chains the GPU kept
    std::vector<int32_t> seg[2];
    struct ChainRef {
        uint32_t target;
        std::vector<int32_t> a, b;
    };
    std::vector<ChainRef> chains[2];
};

struct Task;
struct Sched;

struct MapperImpl {
    dp_ctx* ctx = nullptr;
    int k = 11;
    i64 edgeSize = 1000;
    bool circular = true;
    i64 refLen = 0;
    std::string refName;
    std::vector<Chunk> chunks;
    std::vector<int32_t> chunkSegs;  // host copy of the chunk scan
    Sched* sched = nullptr;
    std::string err;

    bool isConsistent(const Mapping* l, const Mapping* r) const;
    void matchPairs(std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& matched, bool& matchedNil,
                    Task& t);
    void findSplitPoint(Task& t, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, i64 left, i64 right);
    void mapNext(Task& t, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& newA,
                 std::vector<Mapping*>& newB, std::vector<Mapping*>& matched, bool& matchedNil);
    std::vector<Mapping*> map(Task& t);
    std::vector<Mapping*> performMapping(Task& t, i64 a, i64 b, bool whole);
    std::string asString(const Mapping& m, const std::string& qname, i64 qlen) const;
};

struct Task {
    CoroPoint at;           // where the coroutine stands while it is switched out
    char* stack = nullptr;  // from the scheduler's stack pool (uninitialised memory, reused by later reads)
    uint32_t read = 0;
    i64 L = 0;
    bool done = false, waiting = false;
    WindowReq req;
    WindowResult res;
    std::vector<std::unique_ptr<Mapping>> pool;
    std::vector<Mapping*> results;
    MapperImpl* m = nullptr;
    Mapping* mk() {
        pool.emplace_back(new Mapping());
        return pool.back().get();
    }
};

struct Sched {
    CoroPoint main;  // the scheduler's own stack while a coroutine runs
};

// first frame of every coroutine: never returns (the last switch leaves it for good)
void coroTrampoline(void* arg) {
    Task* t = (Task*)arg;
    t->results = t->m->map(*t);
    t->done = true;
    coroSwitch(t->at, t->m->sched->main, true);
}

// ---- mapping.go:131-160
bool MapperImpl::isConsistent(const Mapping* left, const Mapping* right) const {
    if (left->RC != right->RC) return false;
    const i64 expectedDistance = right->QueryOffset - left->queryLen + left->QueryInset;
    i64 distance = !left->RC ? right->Start - left->End : left->Start - right->End;
    if (circular && distance < -50) distance += refLen;
    if (distance < 50 && expectedDistance < 50 && distance > -50) return true;
    if (distance < 500) return (expectedDistance < (distance * 3) / 2 && expectedDistance > (distance * 2) / 3);
    if (distance > 5000) return (expectedDistance < (distance * 10) / 9 && expectedDistance > (distance * 9) / 10);
    double ratio = (double)(distance - 500) / 4500.0;
    ratio = 3.0 / 2.0 + ratio * (10.0 / 9.0 - 3.0 / 2.0);
    return distance < (i64)((double)expectedDistance * ratio) && distance > (i64)((double)expectedDistance / ratio);
}

// ---- removeDominated mapping.go:387-428 (open and extended are the same slice at every call site)
std::vector<Mapping*> removeDominated(std::vector<Mapping*> open, i64 queryLen) {
    if (open.empty()) return open;
    goSort(open, [](Mapping* a, Mapping* b) { return a->QueryOffset < b->QueryOffset; });
    const std::vector<Mapping*>& extended = open;
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
    for (i64 i = last; i >= 0; i--)
        if (toRemove[(size_t)i]) {
            open[(size_t)i] = open[(size_t)last];
            last--;
        }
    open.resize((size_t)(last + 1));
    return open;
}

void updateQuery(std::vector<Mapping*>& ms, i64 len) {
    for (Mapping* m : ms) {
        m->queryLen = len;
        m->hasQuery = true;
    }
}
void appendAll(std::vector<Mapping*>& d, const std::vector<Mapping*>& s) { d.insert(d.end(), s.begin(), s.end()); }

// ---- matchPairs mapping.go:174-203
void MapperImpl::matchPairs(std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& matched,
                            bool& matchedNil, Task& t) {
    matched.clear();
    matchedNil = true;
    for (i64 i = (i64)openA.size() - 1; i >= 0; i--) {
        Mapping* ra = openA[(size_t)i];
        for (i64 j = (i64)openB.size() - 1; j >= 0; j--) {
            Mapping* rb = openB[(size_t)j];
            if (isConsistent(ra, rb)) {
                const i64 qOffset = ra->QueryOffset, qInset = rb->QueryInset;
                if (ra->RC) std::swap(ra, rb);
                Mapping* c = t.mk();
                c->Start = ra->Start;
                c->End = rb->End;
                c->queryLen = ra->queryLen;
                c->hasQuery = ra->hasQuery;
                c->QueryOffset = qOffset;
                c->QueryInset = qInset;
                c->RC = ra->RC;
                c->ids = ra->ids + rb->ids;
                matchedNil = false;
                matched.push_back(c);
                openA[(size_t)i] = openA.back();
                openA.pop_back();
                openB[(size_t)j] = openB.back();
                openB.pop_back();
                break;
            }
        }
    }
}

// ---- findSplitPoint mapping.go:207-288
void MapperImpl::findSplitPoint(Task& t, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, i64 left, i64 right) {
    const i64 qLen = t.L;
    while (right - left >= edgeSize) {
        const i64 start = (right + left - edgeSize) / 2, end = start + edgeSize;
        std::vector<Mapping*> mid = performMapping(t, start, end, false);
        i64 newLeft = left, newRight = right, afterA = 0, afterB = 0;
        for (Mapping* mm : mid) {
            mm->queryLen = qLen;
            mm->hasQuery = true;
            for (Mapping* ma : openA) {
                if (isConsistent(ma, mm)) {
                    ma->QueryInset = mm->QueryInset;
                    ma->ids += mm->ids;
                    if (ma->RC) ma->Start = mm->Start;
                    else ma->End = mm->End;
                    const i64 midMatched = qLen - mm->QueryInset - mm->QueryOffset;
                    if (midMatched > afterA) afterA = midMatched;
                    if (qLen - mm->QueryInset > newLeft) newLeft = qLen - mm->QueryInset;
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
                        const i64 midMatched = qLen - mm->QueryInset - mm->QueryOffset;
                        if (midMatched > afterB) afterB = midMatched;
                        if (mm->QueryOffset < newRight) newRight = mm->QueryOffset;
                        break;
                    }
                }
            }
        }
        if (afterA > 0 && afterB > 0) {
            std::vector<Mapping*> empty;
            if (newLeft - left > edgeSize * 2) findSplitPoint(t, openA, empty, newLeft - edgeSize * 2, newLeft - edgeSize);
            if (right - newRight > edgeSize * 2) findSplitPoint(t, empty, openB, newRight + edgeSize, newRight + edgeSize * 2);
            return;
        }
        if (afterA == 0 && afterB == 0) {
            std::vector<Mapping*> empty;
            if (!openA.empty()) findSplitPoint(t, openA, empty, left, start);
            if (!openB.empty()) findSplitPoint(t, empty, openB, end, right);
            return;
        }
        left = newLeft;
        right = newRight;
    }
}

// ---- mapNext mapping.go:305-383 (Go slice aliasing reproduced with value copies, cf. oracle/mapping.cpp notes)
void MapperImpl::mapNext(Task& t, std::vector<Mapping*>& openA, std::vector<Mapping*>& openB, std::vector<Mapping*>& newA,
                         std::vector<Mapping*>& newB, std::vector<Mapping*>& matched, bool& matchedNil) {
    const i64 qLen = t.L;
    std::vector<Mapping*> extended;
    bool extNil;
    if (qLen < edgeSize * 4) {
        newA = removeDominated(performMapping(t, edgeSize, qLen - edgeSize, false), qLen);
        updateQuery(newA, qLen);
        matchPairs(openA, newA, extended, extNil, t);
        if (!extNil) {
            std::vector<Mapping*> tmp = newA;
            appendAll(tmp, extended);
            openA = tmp;
        } else {
            appendAll(openA, newA);
        }
        matchPairs(openA, openB, matched, matchedNil, t);
        newA = openA;
        newB = openB;
        if (matchedNil) return;
        newA.clear();
        newB.clear();
        return;
    }
    newA = removeDominated(performMapping(t, edgeSize, edgeSize * 2, false), qLen);
    updateQuery(newA, qLen);
    matchPairs(openA, newA, extended, extNil, t);
    appendAll(openA, newA);
    if (!extNil) appendAll(openA, extended);
    newB = removeDominated(performMapping(t, qLen - edgeSize * 2, qLen - edgeSize, false), qLen);
    updateQuery(newB, qLen);
    {
        std::vector<Mapping*> a = newB, b = openB;
        matchPairs(a, b, extended, extNil, t);
        openB = a;
        newB = b;
    }
    appendAll(openB, newB);
    if (!extNil) appendAll(openB, extended);
    {
        std::vector<Mapping*> a = openA, b = openB;
        matchPairs(a, b, matched, matchedNil, t);
        newA = a;
        newB = b;
    }
    if (matchedNil) {
        if (qLen > edgeSize * 5) {
            openA = removeDominated(performMapping(t, edgeSize * 2, edgeSize * 3, false), qLen);
            updateQuery(openA, qLen);
            {
                std::vector<Mapping*> a = newA, b = openA;
                matchPairs(a, b, extended, extNil, t);
                openA = a;
                newA = b;
            }
            if (!extNil) appendAll(openA, extended);
            appendAll(openA, newA);
        }
        if (qLen > edgeSize * 6) {
            openB = removeDominated(performMapping(t, qLen - edgeSize * 3, qLen - edgeSize * 2, false), qLen);
            updateQuery(openB, qLen);
            {
                std::vector<Mapping*> a = openB, b = newB;
                matchPairs(a, b, extended, extNil, t);
                openB = a;
                newB = b;
            }
            if (!extNil) appendAll(openB, extended);
            appendAll(openB, newB);
        } else {
            openB = newB;
        }
        if (qLen > edgeSize * 5) {
            std::vector<Mapping*> a = openA, b = openB;
            matchPairs(a, b, matched, matchedNil, t);
            newA = a;
            newB = b;
        }
    }
}

// ---- Map mapping.go:430-487
std::vector<Mapping*> MapperImpl::map(Task& t) {
    const i64 qLen = t.L;
    std::vector<Mapping*> results;
    if (qLen <= edgeSize * 2) {
        results = removeDominated(performMapping(t, 0, qLen, true), qLen);
        updateQuery(results, qLen);
        return results;
    }
    std::vector<Mapping*> openA = performMapping(t, 0, edgeSize, false);           // mapEnds :164-172
    std::vector<Mapping*> openB = performMapping(t, qLen - edgeSize, qLen, false);
    openA = removeDominated(openA, qLen);
    openB = removeDominated(openB, qLen);
    updateQuery(openA, qLen);
    updateQuery(openB, qLen);
    std::vector<Mapping*> matched;
    bool matchedNil;
    matchPairs(openA, openB, matched, matchedNil, t);
    if (!matchedNil) return matched;
    if (qLen < edgeSize * 3) {
        results = openA;
        appendAll(results, openB);
        return results;
    }
    std::vector<Mapping*> nA, nB;
    mapNext(t, openA, openB, nA, nB, matched, matchedNil);
    openA = nA;
    openB = nB;
    if (!matchedNil) return matched;
    i64 left = edgeSize * 2, right = qLen - edgeSize * 2;
    for (Mapping* a : openA)
        if (a->QueryInset > left) left = a->QueryInset;
    left = qLen - right;  // sic (:461)
    for (Mapping* b : openB)
        if (b->QueryOffset < right) right = b->QueryOffset;
    findSplitPoint(t, openA, openB, left, right);
    const i64 size = qLen - edgeSize;
    for (i64 i = (i64)openA.size() - 1; i >= 0; i--)
        if (openA[(size_t)i]->QueryInset >= size) {
            openA[(size_t)i] = openA.back();
            openA.pop_back();
        }
    for (i64 i = (i64)openB.size() - 1; i >= 0; i--)
        if (openB[(size_t)i]->QueryOffset >= size) {
            openB[(size_t)i] = openB.back();
            openB.pop_back();
        }
    results = openA;
    appendAll(results, openB);
    return results;
}

i64 segSeedOffset(const int32_t* seg, int index, int k) {
    index = index * 2 + 1;
    i64 o = seg[0];
    for (int i = 2; i < index; i += 2) o += seg[i] + k;
    return o;
}
i64 segSeedOffsetFromEnd(const int32_t* seg, int n, int index, int k) {
    index = index * 2 + 1;
    i64 o = seg[n - 1];
    for (int i = n - 3; i > index; i -= 2) o += seg[i] + k;
    return o;
}
// GetBasesCovered second return (target side), seeds/sequence.go:830-858
i64 basesCoveredB(const int32_t* sb, const std::vector<int32_t>& mb, int k) {
    i64 countB = (i64)mb.size() * k;
    int prevB = mb[0];
    for (size_t i = 1; i < mb.size(); i++) {
        const int s2 = mb[i];
        i64 d2 = sb[prevB * 2 + 2];
        for (int j = prevB + 2; j <= s2; j++) d2 += sb[j * 2] + k;
        if (d2 < 0) countB += d2;
        prevB = s2;
    }
    return countB;
}

// ---- performMapping mapping.go:489-611: the window is sent to the GPU batch; this coroutine resumes with the chains.
std::vector<Mapping*> MapperImpl::performMapping(Task& t, i64 a, i64 b, bool whole) {
    if (b > t.L) b = t.L;  // SubSequence clamps end (sequence.go:354)
    t.req.read = t.read;
    t.req.a = a;
    t.req.b = b;
    t.req.whole = whole;
    t.waiting = true;
    coroSwitch(t.at, sched->main);  // yield until the batch has been processed
    t.waiting = false;
    const WindowResult& R = t.res;
    const i64 qlen = b - a;
    // view metadata: SubSequence of a top-level read (offset a, inset L-(b-1)); the RC view swaps them (sequence.go:196)
    i64 vOffset = whole ? 0 : a, vInset = whole ? 0 : t.L - (b - 1);
    std::vector<Mapping*> results;
    for (int s = 0; s < 2; s++) {
        const int32_t* qseg = R.seg[s].data();
        const int qn = (int)R.seg[s].size();
        const i64 sqOffset = s == 0 ? vOffset : vInset, sqInset = s == 0 ? vInset : vOffset;
        for (const auto& ch : R.chains[s]) {
            const Chunk& c = chunks[ch.target];
            i64 start = c.offset + segSeedOffset(c.seg, ch.b[0], k);
            const i64 end = refLen - c.inset - segSeedOffsetFromEnd(c.seg, c.n, ch.b.back(), k);
            if (circular && start > refLen) start -= refLen;
            Mapping* mp = t.mk();
            if (s == 0) {
                i64 qOffset = segSeedOffset(qseg, ch.a[0], k) + sqOffset;
                i64 qInset = segSeedOffsetFromEnd(qseg, qn, ch.a.back(), k) + sqInset;
                mp->QueryOffset = qOffset;
                mp->QueryInset = qInset;
            } else {  // offsets/insets are swapped: they are based on the reverse-complement query (:569-580)
                i64 qInset = segSeedOffset(qseg, ch.a[0], k) + sqOffset;
                i64 qOffset = segSeedOffsetFromEnd(qseg, qn, ch.a.back(), k) + sqInset;
                mp->QueryOffset = qOffset;
                mp->QueryInset = qInset;
            }
            mp->Start = start;
            mp->End = end;
            mp->RC = s == 1;
            mp->ids = basesCoveredB(c.seg, ch.b, k);
            results.push_back(mp);
        }
    }
    (void)qlen;
    if (results.size() > 1) {  // :590-608
        goSort(results, [](Mapping* x, Mapping* y) { return x->Start < y->Start; });
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

// ---- AsString mapping.go:112-122
std::string MapperImpl::asString(const Mapping& m, const std::string& qname, i64 qlen) const {
    i64 mappedLength = m.End - m.Start;
    if (circular && mappedLength < 0) mappedLength = refLen - m.Start + m.End;
    char buf[512];
    snprintf(buf, sizeof buf, "\t%lld\t%lld\t%lld\t%s\t", (long long)qlen, (long long)m.QueryOffset, (long long)(qlen - m.QueryInset),
             m.RC ? "-" : "+");
    std::string s = qname + buf + refName;
    snprintf(buf, sizeof buf, "\t%lld\t%lld\t%lld\t%lld\t%lld\t255", (long long)refLen, (long long)m.Start, (long long)m.End,
             (long long)m.ids, (long long)mappedLength);
    return s + buf;
}


}  // namespace

// ---- test hooks: the mapper's two decision rules on bare numbers (tests/test_hand_known_answers.py: answers worked from the Go text)
// l = {RC, Query.Len(), QueryInset, Start, End}, r = {RC, QueryOffset, Start, End}
bool handIsConsistent(const i64* l, const i64* r, bool circular, i64 refLen) {
    MapperImpl m;
    m.circular = circular;
    m.refLen = refLen;
    Mapping L, R;
    L.RC = l[0] != 0;
    L.queryLen = l[1];
    L.hasQuery = true;
    L.QueryInset = l[2];
    L.Start = l[3];
    L.End = l[4];
    R.RC = r[0] != 0;
    R.QueryOffset = r[1];
    R.Start = r[2];
    R.End = r[3];
    return m.isConsistent(&L, &R);
}
// maps = n x {QueryOffset, QueryInset, ids}; kept[] = indices of the survivors in the order removeDominated returns them
int handRemoveDominated(const i64* maps, int n, i64 queryLen, int* kept) {
    std::vector<Mapping> store((size_t)n);
    std::vector<Mapping*> open;
    for (int i = 0; i < n; i++) {
        store[(size_t)i].QueryOffset = maps[3 * i];
        store[(size_t)i].QueryInset = maps[3 * i + 1];
        store[(size_t)i].ids = maps[3 * i + 2];
        open.push_back(&store[(size_t)i]);
    }
    std::vector<Mapping*> out = removeDominated(open, queryLen);
    for (size_t i = 0; i < out.size(); i++) kept[i] = (int)(out[i] - store.data());
    return (int)out.size();
}
namespace {
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// commands/map.go:33-116 + NewMapper mapping.go:67-109

// the staging block [reference | join chunk | all reads] of the last map command of this process, kept for the next one
namespace {
std::mutex g_stagingMu;
std::unique_ptr<char[]> g_staging;
size_t g_stagingCap = 0;
std::unique_ptr<char[]> stagingTake(size_t need, size_t* cap) {
    {
        std::lock_guard<std::mutex> lk(g_stagingMu);
        if (g_staging && g_stagingCap >= need) {
            *cap = g_stagingCap;
            g_stagingCap = 0;
            return std::move(g_staging);
        }
    }
    *cap = need;
    return std::unique_ptr<char[]>(new char[need]);
}
void stagingPut(std::unique_ptr<char[]> p, size_t cap) {
    std::lock_guard<std::mutex> lk(g_stagingMu);
    if (cap > g_stagingCap) {  // (the smaller of two blocks goes: its munmap is the caller's 40 ms, once)
        g_staging = std::move(p);
        g_stagingCap = cap;
    }
}
}  // namespace
size_t releaseMapStaging() {
    std::lock_guard<std::mutex> lk(g_stagingMu);
    const size_t freed = g_staging ? g_stagingCap : 0;
    g_staging.reset();
    g_stagingCap = 0;
    return freed;
}

int runMap(const ReadSet& refSet, const ReadSet& reads, const MapParams& p, int device, std::string& paf, std::string& errText,
           MapStats* stats, std::string& error) {
    const double tRun0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool prof = getenv("DPH_PROFILE") != nullptr;
    double tMark = tRun0;
    auto mark = [&](const char* what) {
        if (!prof) return;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[map setup] %-28s %.1f ms\n", what, 1e3 * (t - tMark));
        tMark = t;
    };
    if (refSet.size() == 0) {
        error = "empty reference";
        return -1;
    }
    const int k = p.k;
    const i64 refLen = refSet.length(0);
    const char* ref = refSet.seq(0);
    // ---- device read set: [0] reference, [1] circular join chunk, then (forward, reverse complement) of every read.  Its host
    // staging copy is made by a thread of its own while the device computes the reference's k-mer table
    std::vector<i64> off(1, 0);
    std::string join;
    if (p.circular) join = std::string(ref + (refLen - p.querySize), (size_t)p.querySize) + std::string(ref, (size_t)p.querySize);
    // one staging buffer [reference | join chunk | all reads]; the reads are one contiguous block of the read set, copied
    // (and its pages first touched) by the worker pool in 4 MiB pieces
    const size_t head = (size_t)refLen + join.size();
    const size_t readBytes = reads.size() ? (size_t)(reads.off[reads.size()] - reads.off[0]) : 0;
    // Round 6: the reads cross PCIe the way the reference holds them - 2 bits per base (sequence.packedSequence) - packed here by
    // the worker pool straight into a pinned block in the device's own layout (every read on a 16-byte boundary) and put in place by
    // dp_reads_upload_packed_rc, which also makes the reverse strands: 100 MB instead of 400 at config 3, one copy instead of three
    // (staging, pinned ring, link).  DP_TUNE=map_ascii_upload=1: the ASCII path below, as before (it is what map_async_upload uses).
    const bool asyncUpload = dph_tune("map_async_upload", 0) != 0;
    const bool packedUpload = !asyncUpload && !dph_tune("map_ascii_upload", 0);
    const bool packScalar = dph_tune("pack_scalar", 0) != 0;  // (tests: the packer without its AVX2 path)
    std::vector<uint32_t> plens;
    std::vector<uint64_t> poff;
    uint8_t* pinned = nullptr;
    struct PinnedBack {
        uint8_t*& p;
        ~PinnedBack() {
            if (p) dp_host_free(p);
        }
    } pinnedBack{pinned};
    if (packedUpload) {
        const size_t n = 2 + reads.size();
        plens.resize(n);
        poff.assign(n + 1, 0);
        plens[0] = (uint32_t)refLen;
        plens[1] = (uint32_t)join.size();
        for (size_t r = 0; r < reads.size(); r++) plens[2 + r] = (uint32_t)reads.length(r);
        for (size_t r = 0; r < n; r++) poff[r + 1] = poff[r] + ((((uint64_t)plens[r] + 3) / 4 + 15) & ~(uint64_t)15);
        pinned = (uint8_t*)dp_host_alloc((size_t)poff[n] + 64);
        if (!pinned) {
            error = "dp_host_alloc: no pinned memory for the packed reads";
            return DP_ERR_HIP;
        }
    }
    size_t stagingCap = 0;
    std::unique_ptr<char[]> staging = packedUpload ? std::unique_ptr<char[]>() : stagingTake(head + readBytes + 1, &stagingCap);
    std::thread concatThread([&] {
        if (packedUpload) {
            packBases(ref, (size_t)refLen, pinned + poff[0], packScalar);
            packBases(join.data(), join.size(), pinned + poff[1], packScalar);
            // pieces of about 2 M bases: whole reads
            std::vector<size_t> cut(1, 0);
            i64 acc = 0;
            for (size_t r = 0; r < reads.size(); r++) {
                acc += reads.length(r);
                if (acc >= ((i64)2 << 20)) {
                    cut.push_back(r + 1);
                    acc = 0;
                }
            }
            if (cut.back() != reads.size()) cut.push_back(reads.size());
            parallelFor(cut.size() - 1, [&](size_t i) {
                for (size_t r = cut[i]; r < cut[i + 1]; r++) packBases(reads.seq(r), (size_t)reads.length(r), pinned + poff[2 + r], packScalar);
            });
            return;
        }
        memcpy(staging.get(), ref, (size_t)refLen);
        memcpy(staging.get() + refLen, join.data(), join.size());
        off.push_back((i64)refLen);
        off.push_back((i64)head);
        const char* src = reads.size() ? reads.seq(0) : nullptr;
        const size_t piece = (size_t)4 << 20, nPieces = (readBytes + piece - 1) / piece;
        char* dst = staging.get() + head;
        parallelFor(nPieces, [&](size_t i) {
            const size_t b = i * piece, e = std::min(readBytes, b + piece);
            memcpy(dst + b, src + b, e - b);
        });
        const i64 shift = (i64)head - (reads.size() ? reads.off[0] : 0);
        for (size_t r = 0; r < reads.size(); r++) off.push_back(reads.off[r + 1] + shift);
    });
    struct ConcatJoin {
        std::thread& t;
        ~ConcatJoin() {
            if (t.joinable()) t.join();
        }
    } concatJoin{concatThread};
    dp_ctx* ctx = nullptr;
    if (dp_ctx_create(device, &ctx) != 0) {
        error = dp_last_error(nullptr);
        return DP_ERR_NODEVICE;
    }
    auto fail = [&](int rc) {
        error = dp_last_error(ctx);
        dp_ctx_destroy(ctx), ctx = nullptr;
        return rc;
    };
    mark("context");
    // ---- value table from every sequence of the reference file (map.go:45-71); KmerOccurrences on the GPU
    int rc = dp_reads_upload(ctx, (const uint8_t*)refSet.bases.data(), refSet.off.data(), (uint32_t)refSet.size());
    if (rc) return fail(rc);
    std::vector<double> values((size_t)1 << (2 * k));
    rc = dp_kmer_values(ctx, k, values.data());  // KmerOccurrences + value table + 1 % cut on the GPU
    if (rc) return fail(rc);
    mark("reference upload + value table");
    errText += "K-mer counting complete. Preparing to start indexing and querying...\n";

    // ---- AddSingleSeeds seeds/seeds.go:160-200 on the (top-level) reference, host, sequential - on a thread of its own while
    // this one concatenates the reads and the device uploads and packs them (neither needs the seeds)
    SeedIndex index(k);
    // Round 4: everything about a window that does not depend on the seeds added before it - its best k-mer, and which k-mers of
    // its count region are the best of ANY window (only those can ever be seeds) - comes from the device for all windows at once
    // (dp_single_seed_candidates: the reference and the value table are resident right now); the thread below then walks the
    // windows in order and probes five or six candidates per window instead of seed_rate k-mers (3.9 s -> 0.1 s per 375 Mb of
    // reference).  DP_MAP_SEEDS_HOST=1: the whole walk on the host as before.
    dp_single_seed_batch ssb;
    memset(&ssb, 0, sizeof ssb);
    bool deviceSeeds = false;
    {
        if (!dph_tune("map_seeds_host", 0) && refLen < ((i64)1 << 32)) {  // (tests: AddSingleSeeds walked on the host)
            rc = dp_single_seed_candidates(ctx, 0, k, p.seedRate, &ssb);
            if (rc) return fail(rc);
            deviceSeeds = true;
            mark("single-seed candidates (device)");
        }
    }
    std::thread seedThread([&] {
        if (deviceSeeds) {
            const double ts0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            for (uint32_t w = 0; w < ssb.n_windows; w++) {
                bool any = false;
                for (uint32_t c = ssb.cand_off[w]; c < ssb.cand_off[w + 1] && !any; c++) any = index.isSeed(ssb.cand[c]);
                if (!any) index.addSeedKmer(ssb.best[w]);
            }
            if (prof)
                fprintf(stderr, "[map setup] single-seed walk: %u windows, %u candidates, %zu seeds, %.1f ms on its thread\n", ssb.n_windows,
                        ssb.cand_off[ssb.n_windows], index.seedMap.size(),
                        1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - ts0));
            return;
        }
        const uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * k)) - 1);
        const int finalLen = (int)(refLen % 4);  // top-level sequence: 0 when len%4 == 0 (sequence.go:70,88)
        const i64 skipBack = 4 - finalLen;
        auto code = [&](i64 pos) -> uint32_t { return pos < refLen ? baseCode((unsigned char)ref[pos]) : 0u; };  // zero padding
        auto kmerAt = [&](i64 pos) {
            uint32_t v = 0;
            for (int j = 0; j < k; j++) v = (v << 2) | code(pos + j);
            return v;
        };
        for (i64 i = 0; i < refLen - p.seedRate; i += p.seedRate) {
            // CountKmersBetween(i, i+seedRate, 1, ...) == 0 ?  (sequence.go:332-337 + asm:81-203: whole bytes only,
            // parent's skipBack, do-while group loop)
            const i64 startB = (i + 3) / 4, endB = (i + p.seedRate) / 4;
            const i64 nb = endB - startB;
            const i64 nk = 4 * (nb - 1) - skipBack - k + 1;
            i64 groups = (nk & ~(i64)3) / 4;
            if (groups < 1) groups = 1;
            const i64 P = 4 + 4 * groups + (nk & 3);
            bool any = false;
            {
                const i64 p0 = startB * 4;
                uint32_t km = kmerAt(p0);
                for (i64 j = 0; j < P; j++) {
                    if (j) km = ((km << 2) | code(p0 + j + k - 1)) & mask;
                    if (index.isSeed(km)) {
                        any = true;
                        break;
                    }
                }
            }
            if (!any) {
                const i64 end = i + p.seedRate;
                uint32_t km = kmerAt(i);
                double bestValue = values[km];
                uint32_t best = km;
                for (i64 j = i + k; j < end; j++) {
                    km = ((km << 2) | code(j)) & mask;
                    const double v = values[km];
                    if (v > bestValue) {
                        bestValue = v;
                        best = km;
                    }
                }
                index.addSeedKmer(best);
            }
        }
    });
    struct SeedJoin {  // (every way out of this function waits for the thread)
        std::thread& t;
        ~SeedJoin() {
            if (t.joinable()) t.join();
        }
    } seedJoin{seedThread};
    // device ids: 0 reference, 1 join chunk, then (forward, reverse complement) per read; the reverse strands are made
    // on the device
    concatThread.join();
    mark("concatenate reads (waited for)");
    // Round 5, DP_MAP_ASYNC_UPLOAD=1: the reads travel while they are mapped (dp_reads_upload_rc_begin: this call returns when the
    // reference and the join chunk are packed; the mapper threads wait for "reads below n are packed" block by block).  Built, parity
    // tested and left OFF: set-up 14.5 -> 8 ms and the best run 56.0 -> 54.4 ms, but the runs of a process spread 55 - 107 ms where
    // they were 56 - 65 (means of 12 runs 69 / 77 against 62 / 69 ms, profiles/r05/map_threads_and_reads_in_flight.txt): the upload's
    // copy threads and the link compete with six mapper threads for the same host cores and queues.
    // (giving 400 MB of staging back to the system is 40 ms of munmap - round 3's profile had booked it as "AddSingleSeeds (waited
    // for)" - and on a thread of its own it holds the address-space lock against this one's allocations just as long: the block is
    // kept for the process's next map command instead, which then also finds its pages touched)
    struct StagingBack {  // every way out: the upload's thread has read the last byte before the block changes hands
        dp_ctx*& c;
        std::unique_ptr<char[]>& st;
        size_t& cap;
        ~StagingBack() {
            if (c) (void)dp_reads_upload_wait(c, 0xffffffffu);
            if (st) stagingPut(std::move(st), cap);
        }
    } stagingBack{ctx, staging, stagingCap};
    rc = packedUpload  ? dp_reads_upload_packed_rc(ctx, pinned, plens.data(), (uint32_t)plens.size(), 2)
         : asyncUpload ? dp_reads_upload_rc_begin(ctx, (const uint8_t*)staging.get(), off.data(), (uint32_t)(off.size() - 1), 2, 2)
                       : dp_reads_upload_rc(ctx, (const uint8_t*)staging.get(), off.data(), (uint32_t)(off.size() - 1), 2);
    if (rc) return fail(rc);
    mark(packedUpload ? "packed upload + reverse strands" : asyncUpload ? "upload begun (reference packed)" : "upload + pack (both strands)");
    seedThread.join();
    mark("AddSingleSeeds (waited for)");
    rc = dp_round_begin(ctx, k, index.seedMap.data(), (uint32_t)index.seedMap.size());
    if (rc) return fail(rc);

    // ---- chunk schedule mapping.go:79-96, canonical generation order
    MapperImpl M;
    M.ctx = ctx;
    M.k = k;
    M.edgeSize = p.querySize;
    M.circular = p.circular;
    M.refLen = refLen;
    M.refName = refSet.names[0];
    std::vector<dp_scan_item> items;
    std::vector<std::pair<i64, i64>> meta;  // offset, inset per chunk
    for (i64 j = 0; j < 10; j++) {
        const i64 start = j * p.chunkSize, step = p.chunkSize * 10 - p.querySize;
        for (i64 i = start; i < refLen - p.chunkSize / 2; i += step) {
            i64 end = i + p.chunkSize;
            if (i >= refLen) end = refLen;
            if (end > refLen) end = refLen;  // SubSequence clamps
            dp_scan_item it;
            it.read = 0;
            it.start = (uint32_t)i;
            it.n_kmers = (uint32_t)std::max<i64>(0, (end - i) - k + 1);
            it.min_seeds = 0;
            items.push_back(it);
            meta.push_back({i, refLen - (end - 1)});  // SubSequence: offset+start, inset + length - (end-1)
        }
    }
    if (p.circular) {
        // Append(...) is a fresh top-level sequence of 2*edge bases: len%4==0 loses its last 4 k-mers (asm:88-96)
        const i64 jl = (i64)join.size();
        i64 nk = jl - k + 1;
        if (jl % 4 == 0) nk -= 4;
        dp_scan_item it;
        it.read = 1;
        it.start = 0;
        it.n_kmers = (uint32_t)std::max<i64>(0, nk);
        it.min_seeds = 0;
        items.push_back(it);
        meta.push_back({refLen - p.querySize, refLen - (p.querySize - 1)});  // offset of the first part, inset of the second
    }
    dp_seedseq_batch sb;
    rc = dp_scan(ctx, items.data(), (uint32_t)items.size(), &sb);
    if (rc) return fail(rc);
    M.chunkSegs.assign(sb.segs, sb.segs + sb.n_segs);
    std::vector<dp_seq_ref> refs(items.size());
    M.chunks.resize(items.size());
    for (size_t i = 0; i < items.size(); i++) {
        refs[i].seg_off = sb.seg_off[i];
        refs[i].n_seeds = sb.n_seeds[i];
        refs[i].reserved = 0;
        M.chunks[i].seg = M.chunkSegs.data() + sb.seg_off[i];
        M.chunks[i].n = (int)(sb.seg_off[i + 1] - sb.seg_off[i]);
        M.chunks[i].offset = meta[i].first;
        M.chunks[i].inset = meta[i].second;
    }
    // ---- DP_MAP_SHARDS=N: the reference index spread over N contexts (DP_MAP_DEVICES=0,1,..: their GPUs, round robin;
    // default all on this one) - what BASELINE config 5 does with a 3 Gb reference on 8 GPUs.  Shard s holds the chunks
    // [c0, c1) (c0 a multiple of 64): their segments (imported from the chunk scan above), posting and seed-set words.  The
    // sets' global windows are combined here once (include/downpore_hip.h, dp_index_set_global).
    struct Shard {
        dp_ctx* ctx = nullptr;
        uint32_t c0 = 0, c1 = 0;
    };
    std::vector<Shard> shards;
    struct ShardGuard {
        std::vector<Shard>& v;
        ~ShardGuard() {
            for (Shard& sh : v)
                if (sh.ctx) dp_ctx_destroy(sh.ctx);
        }
    } shardGuard{shards};
    int nShards = 1;
    if (const char* e = getenv("DP_MAP_SHARDS")) nShards = std::max(1, atoi(e));
    if (nShards > 1) {
        std::vector<int> devs;
        if (const char* e = getenv("DP_MAP_DEVICES")) {
            for (const char* q = e; *q;) {
                devs.push_back(atoi(q));
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
        }
        if (devs.empty()) devs.push_back(device);
        const uint32_t nChunks = (uint32_t)items.size(), S = (uint32_t)index.seedMap.size();
        const uint32_t per = ((nChunks + (uint32_t)nShards - 1) / (uint32_t)nShards + 63) / 64 * 64;  // whole 64-chunk words
        std::vector<uint32_t> global((size_t)S * 4), local((size_t)S * 4);
        for (uint32_t i = 0; i < S; i++) {  // NewIntSet(): count 0, start 1, end 0 (last + 1 = 1)
            global[4 * (size_t)i + 0] = 0;
            global[4 * (size_t)i + 1] = 1;
            global[4 * (size_t)i + 2] = 0;
            global[4 * (size_t)i + 3] = 1;
        }
        for (uint32_t c0 = 0; c0 < nChunks; c0 += per) {
            Shard sh;
            sh.c0 = c0;
            sh.c1 = std::min(nChunks, c0 + per);
            if (dp_ctx_create(devs[shards.size() % devs.size()], &sh.ctx) != 0) {
                error = dp_last_error(nullptr);
                dp_ctx_destroy(ctx), ctx = nullptr;
                return DP_ERR_NODEVICE;
            }
            shards.push_back(sh);
            dp_ctx* sc = sh.ctx;
            auto sfail = [&](int rc2) {
                error = dp_last_error(sc);
                dp_ctx_destroy(ctx), ctx = nullptr;
                return rc2;
            };
            rc = dp_round_begin(sc, k, index.seedMap.data(), S);
            if (rc) return sfail(rc);
            const uint64_t s0 = sb.seg_off[sh.c0], s1 = sb.seg_off[sh.c1];
            rc = dp_scan_import_segments(sc, M.chunkSegs.data() + s0, s1 - s0);
            if (rc) return sfail(rc);
            std::vector<dp_seq_ref> lrefs(refs.begin() + sh.c0, refs.begin() + sh.c1);
            for (dp_seq_ref& r : lrefs) r.seg_off -= s0;
            rc = dp_index_build(sc, lrefs.data(), (uint32_t)lrefs.size());
            if (rc) return sfail(rc);
            rc = dp_index_meta(sc, local.data(), S);
            if (rc) return sfail(rc);
            const uint32_t wb = sh.c0 / 64;
            for (uint32_t i = 0; i < S; i++) {
                const uint32_t cnt = local[4 * (size_t)i];
                if (!cnt) continue;
                uint32_t* g = &global[4 * (size_t)i];
                const uint32_t st = local[4 * (size_t)i + 1] + wb, en = local[4 * (size_t)i + 2] + wb;
                if (g[0] == 0) {
                    g[1] = st;
                    g[2] = en;
                } else {
                    g[1] = std::min(g[1], st);
                    g[2] = std::max(g[2], en);
                }
                g[0] += cnt;
                g[3] = g[2] + 1;
            }
        }
        for (Shard& sh : shards) {
            rc = dp_index_set_global(sh.ctx, global.data(), S, sh.c0 / 64, nChunks);
            if (rc) {
                error = dp_last_error(sh.ctx);
                dp_ctx_destroy(ctx), ctx = nullptr;
                return rc;
            }
        }
    } else {
        rc = dp_index_build(ctx, refs.data(), (uint32_t)refs.size());
        if (rc) return fail(rc);
    }
    // dp_scan below reuses the device scan buffer: keep the chunk segments in a dedicated import
    // (dp_index_build references the device-resident scan output, so windows must not overwrite it)
    if (stats) stats->n_chunks = items.size(), stats->n_seeds = index.seedMap.size();

    mark("round begin + chunk scan + index");
    // ---- Map every read: coroutines + batched windows.  The reads are dealt to a few host threads in contiguous ranges: every
    // thread runs this loop on a context of its own (the packed reads are shared, the reference index - half a megabyte of
    // chunk segments at E. coli scale - is built once per context), so one thread's mapper control flow (mapEnds / mapNext /
    // findSplitPoint of 4 096 reads in flight) runs while another thread's windows are on the GPU.  Output is per read, in read
    // order, whatever the thread.  (mapping.go:613-619 MapWorker: the reference does the same with goroutines.)
    auto wallNow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tLoop0 = wallNow();
    if (stats) stats->t_setup_s = tLoop0 - tRun0;
    std::vector<std::string> out(reads.size());
    std::vector<int> nmaps(reads.size(), 0);
    i64 unmapped = 0, mapped = 0, multiple = 0, total = 0;
    size_t nThreadsPlanned = 1;
    struct LoopStats {
        MapStats st;
        double tScan = 0, tChain = 0, wall = 0;
        int rc = 0;
        std::string error;
    };
    dp_ctx* const ownerCtx = ctx;  // (owns the reads: the threads ask it how far their upload has come)
    auto mapLoop = [&](dp_ctx* tctx, MapperImpl& Mt, size_t tIdx, size_t nT, LoopStats& ls) {
    const double tl0 = wallNow();
    double tScan = 0, tChain = 0;
    int rc = 0;
    MapStats* stats = &ls.st;
    auto fail = [&](int rc2) {
        ls.error = dp_last_error(tctx);
        ls.rc = rc2;
        return rc2;
    };
    dp_ctx* ctx = tctx;
    MapperImpl& M = Mt;
    Sched sched;
    M.sched = &sched;
    // reads in flight per thread = windows per dp_map_windows call (DP_MAP_INFLIGHT: another number; a coroutine's stack is 256 KiB
    // of address space, touched pages only)
    // Measured at config 3 (profiles/r05/map_threads_and_reads_in_flight.txt): 3 threads x 1 365 reads = 120 launches, 20.7 ms of
    // map_kernel, loops of 60 - 70 ms; 4 x 2 730 = 64 launches, 14.5 ms, loops of 41 - 46 ms; 8 192 and more per thread: fewer launches
    // still (10.6 ms of kernels) but the threads' control flow no longer fits their caches and runs less beside the GPU's work
    // and once a context's teardown no longer cost 5 ms, more threads paid: 6 x 2 730 = 78 launches, 18 ms of kernels, a run of 58 - 60 ms
    // (856 k reads/s) against 64 - 70 ms with 4 x 2 730 and 61 - 70 with 8 x 2 048
    // round 6, once a context no longer costs a thread 4 ms: 8 x 2 048 runs 46 - 47 ms where 6 x 2 730 runs 50 - 51 (three alternations on
    // one box, profiles/r06/map_threads_sweep.txt)
    size_t inflight = nThreadsPlanned >= 8 ? 2048 : std::max<size_t>(2730, 10922 / nThreadsPlanned);
    if (const char* e = getenv("DP_MAP_INFLIGHT")) inflight = (size_t)std::max(64, atoi(e));
    const size_t stackBytes = 256 * 1024;
    std::vector<std::unique_ptr<Task>> live;
    // coroutine stacks: allocated once (never zero-filled) and recycled — 50 k reads x 256 KiB of fresh zeroed vectors
    // used to be most of the run time
    std::vector<std::unique_ptr<char[]>> stackStore;
    std::vector<char*> freeStacks;
    // The reads are dealt in blocks of 1 024: thread t takes blocks t, t + nT, ... - so that every thread finds its first block on the
    // device a fraction of a millisecond into the upload (contiguous shares would leave the last thread waiting for five sixths of it).
    // Output is per read, in read order, whoever mapped it.
    const size_t nReads = reads.size(), BL = 1024;
    size_t myCount = 0;
    for (size_t blk = tIdx; blk * BL < nReads; blk += nT) myCount += std::min(BL, nReads - blk * BL);
    auto readOf = [&](size_t seq) { return ((seq / BL) * nT + tIdx) * BL + seq % BL; };
    size_t nextSeq = 0, waitedBelow = 0;
    std::vector<dp_scan_item> witems;
    std::vector<int32_t> wsegs;
    std::vector<uint64_t> woff;
    std::vector<uint32_t> wlen;
    double tp[6] = {0, 0, 0, 0, 0, 0};
    while (nextSeq < myCount || !live.empty()) {
        double tq = wallNow();
        auto lap = [&](int i) {
            const double t = wallNow();
            tp[i] += t - tq;
            tq = t;
        };
        while (live.size() < inflight && nextSeq < myCount) {
            const size_t nextRead = readOf(nextSeq);
            if (nextRead >= waitedBelow) {  // (the block may still be on its way to the device: dp_reads_upload_rc_begin)
                const size_t upto = std::min(nReads, (nextRead / BL + 1) * BL);
                rc = dp_reads_upload_wait(ownerCtx, (uint32_t)(2 + upto));  // (host reads 0 and 1: the reference and its join chunk)
                if (rc) return fail(rc);
                waitedBelow = upto;
            }
            std::unique_ptr<Task> t(new Task());
            t->m = &M;
            t->read = (uint32_t)nextRead;
            t->L = reads.length(nextRead);
            if (freeStacks.empty()) {
                stackStore.emplace_back(new char[stackBytes]);
                freeStacks.push_back(stackStore.back().get());
            }
            t->stack = freeStacks.back();
            freeStacks.pop_back();
            coroStart(t->at, t->stack, stackBytes, &coroTrampoline, t.get());
            nextSeq++;
            coroSwitch(sched.main, t->at);  // run until the first window request (or completion)
            live.push_back(std::move(t));
        }
        // retire finished tasks
        for (size_t i = 0; i < live.size();) {
            if (live[i]->done) {
                Task& t = *live[i];
                std::string& o = out[t.read];
                for (Mapping* mm : t.results) o += M.asString(*mm, reads.names[t.read], t.L) + "\n";
                nmaps[t.read] = (int)t.results.size();
                coroRelease(t.at);
                freeStacks.push_back(t.stack);
                live[i] = std::move(live.back());
                live.pop_back();
            } else {
                i++;
            }
        }
        lap(0);  // new tasks up to their first window + finished ones retired
        if (live.empty()) continue;
        // ---- batch: scan the requested windows (forward, reverse complement) ...
        witems.clear();
        for (auto& tp : live) {
            Task& t = *tp;
            const WindowReq& q = t.req;
            const uint32_t fr = 2 + 2 * q.read, rr = fr + 1;
            const i64 L = t.L, wl = q.b - q.a;
            dp_scan_item f, r;
            f.read = fr;
            r.read = rr;
            f.min_seeds = r.min_seeds = 0;
            i64 nk = wl - k + 1;
            if (q.whole && (L % 4) == 0) {
                // top-level query with finalLen == 0: the forward scan loses 4 k-mers; its ReverseComplement() has
                // firstLen == 0 and the scan starts 4 bases in (sequence.go:196, asm:109-132) — fixed up below
                f.start = 0;
                f.n_kmers = (uint32_t)std::max<i64>(0, nk - 4);
                r.start = 4;
                r.n_kmers = (uint32_t)std::max<i64>(0, nk - 4);
            } else {
                f.start = (uint32_t)q.a;
                f.n_kmers = (uint32_t)std::max<i64>(0, nk);
                r.start = (uint32_t)(L - q.b);
                r.n_kmers = (uint32_t)std::max<i64>(0, nk);
            }
            witems.push_back(f);
            witems.push_back(r);
        }
        // the chunk segments live in the device scan buffer; window scans must not clobber them -> the window scan uses a
        // second context-independent path: scan, then re-import the chunk segments before the map stage
        lap(1);  // window items
        dp_seedseq_batch wb;
        const double ts0 = wallNow();
        rc = dp_scan(ctx, witems.data(), (uint32_t)witems.size(), &wb);
        if (rc) return fail(rc);
        tScan += wallNow() - ts0;
        if (stats) {
            stats->k_scan_ms += wb.kernel_ms, stats->n_windows += witems.size() / 2;
            for (const dp_scan_item& wi : witems) stats->scan_bytes += (double)((wi.n_kmers + 3) / 4);  // packed bases of the window, both strands
        }
        wsegs.clear();
        woff.assign(1, 0);
        wlen.clear();
        for (size_t w = 0; w < witems.size(); w++) {
            Task& t = *live[w / 2];
            const bool rcside = (w & 1) != 0;
            const size_t base = wsegs.size();
            wsegs.insert(wsegs.end(), wb.segs + wb.seg_off[w], wb.segs + wb.seg_off[w + 1]);
            if (rcside && t.req.whole && (t.L % 4) == 0) {
                // firstLen==0 quirk: k-mer index 0 duplicates the k-mer at base 4, every later index is shifted by one
                int32_t* sg = wsegs.data() + base;
                const size_t n = wsegs.size() - base;
                if (n >= 3 && sg[0] == 0) {  // the duplicated k-mer is a seed: it appears twice, 1-k apart
                    std::vector<int32_t> fix;
                    fix.push_back(0);
                    fix.push_back(sg[1]);
                    fix.push_back(1 - k);
                    fix.insert(fix.end(), sg + 1, sg + n);
                    wsegs.resize(base);
                    wsegs.insert(wsegs.end(), fix.begin(), fix.end());
                } else {
                    sg[0] += 1;  // all indices shift by one (no hits: the single gap counts the extra k-mer too)
                }
            }
            woff.push_back(wsegs.size());
            wlen.push_back((uint32_t)(t.req.b - t.req.a));
        }
        for (auto& tp : live) {
            tp->res.seg[0].clear();
            tp->res.seg[1].clear();
            tp->res.chains[0].clear();
            tp->res.chains[1].clear();
        }
        for (size_t w = 0; w < witems.size(); w++)
            live[w / 2]->res.seg[w & 1].assign(wsegs.begin() + (i64)woff[w], wsegs.begin() + (i64)woff[w + 1]);
        tq = ts0 + (wallNow() - ts0);  // (the scan call itself is tScan)
        tp[3] += wallNow() - ts0;      // scan call + copying its segments out
        auto takeChains = [&](const dp_chain_batch& cb, uint32_t chunkBase) {
            for (uint32_t c = 0; c < cb.n_chains; c++) {
                const uint32_t w = cb.window[c];
                WindowResult::ChainRef cr;
                cr.target = cb.target[c] + chunkBase;
                cr.a.assign(cb.match_a + cb.off[c], cb.match_a + cb.off[c + 1]);
                cr.b.assign(cb.match_b + cb.off[c], cb.match_b + cb.off[c + 1]);
                live[w / 2]->res.chains[w & 1].push_back(std::move(cr));
            }
            if (stats) stats->k_map_ms += cb.kernel_ms, stats->n_chains += cb.n_chains, stats->map_bytes += cb.alg_bytes;
        };
        tq = wallNow();
        lap(2);
        const double tc0 = wallNow();
        if (shards.empty()) {
            rc = dp_scan_import_segments(ctx, M.chunkSegs.data(), M.chunkSegs.size());
            if (rc) return fail(rc);
            dp_chain_batch cb;
            rc = dp_map_windows(ctx, wsegs.data(), woff.data(), wlen.data(), (uint32_t)witems.size(), k, &cb);
            if (rc) return fail(rc);
            takeChains(cb, 0);
        } else {
            // candidates of a window in ascending chunk id = shard after shard; forward strand over all shards first, its
            // ratchet raises the reverse threshold too (mapping.go:543-549): the thresholds travel with the batch
            std::vector<int32_t> thr(witems.size(), -1);
            for (int phase = 0; phase < 2; phase++)
                for (Shard& sh : shards) {
                    dp_chain_batch cb;
                    rc = dp_map_windows_shard(sh.ctx, wsegs.data(), woff.data(), wlen.data(), (uint32_t)witems.size(), k, phase, thr.data(), &cb);
                    if (rc) {
                        ls.error = dp_last_error(sh.ctx);
                        ls.rc = rc;
                        return rc;
                    }
                    takeChains(cb, sh.c0);
                }
        }
        tChain += wallNow() - tc0;
        if (stats) stats->n_batches++;
        tq = wallNow();
        // ---- ... distribute and resume
        for (auto& tk : live) coroSwitch(sched.main, tk->at);
        lap(4);  // coroutines resumed: performMapping's tail + the mapper's control flow up to the next window
    }
    if (prof)
        fprintf(stderr, "[map loop] thread %zu of %zu: start/retire + upload waits %.1f ms, items %.1f, scan+copy %.1f (scan call %.1f), map call %.1f, resume %.1f\n", tIdx, nT,
                1e3 * tp[0], 1e3 * tp[1], 1e3 * tp[3], 1e3 * tScan, 1e3 * tChain, 1e3 * tp[4]);
    ls.tScan = tScan;
    ls.tChain = tChain;
    ls.wall = wallNow() - tl0;
    return 0;
    };  // mapLoop

    size_t nThreads = 1;
    if (shards.empty()) {
        const char* e = getenv("DP_MAP_THREADS");
        nThreads = (size_t)std::max(1, e ? atoi(e) : (hostThreads() >= 16 ? 8 : hostThreads() >= 12 ? 6 : hostThreads() >= 8 ? 4 : 3));
        size_t perThread = 2048;  // (fewer reads than that per thread are not worth a context; DP_MAP_MIN_READS_PER_THREAD: test hook)
        perThread = (size_t)std::max(1L, dph_tune("map_min_reads_per_thread", (long)perThread));
        nThreads = std::min(nThreads, std::max<size_t>(1, reads.size() / perThread));
    }
    nThreadsPlanned = nThreads;
    std::vector<LoopStats> lstats(nThreads);
    std::vector<dp_ctx*> tctx(nThreads, nullptr);
    std::vector<MapperImpl> Ms(nThreads, M);  // (chunks[i].seg keep pointing into M.chunkSegs, which outlives the threads)
    tctx[0] = ctx;
    {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nThreads; t++) {
            th.emplace_back([&, t] {
                LoopStats& ls = lstats[t];
                const double tt0 = wallNow();
                if (t > 0) {  // a context of its own: shared packed reads, the same seeds, the reference index from the chunk scan above
                    int rc2 = dp_ctx_create_shared(ctx, &tctx[t]);
                    const double ta = wallNow();
                    if (rc2 == 0) rc2 = dp_round_begin(tctx[t], k, index.seedMap.data(), (uint32_t)index.seedMap.size());
                    const double tb = wallNow();
                    if (rc2 == 0) rc2 = dp_scan_import_segments(tctx[t], M.chunkSegs.data(), M.chunkSegs.size());
                    const double tc = wallNow();
                    if (rc2 == 0) rc2 = dp_index_build(tctx[t], refs.data(), (uint32_t)refs.size());
                    if (prof) fprintf(stderr, "[map thread %zu] context %.2f ms, round begin %.2f, import %.2f, index %.2f\n", t, 1e3 * (ta - tt0), 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (wallNow() - tc));
                    if (rc2 != 0) {
                        ls.rc = rc2;
                        ls.error = tctx[t] ? dp_last_error(tctx[t]) : dp_last_error(nullptr);
                        return;
                    }
                }
                Ms[t].ctx = tctx[t];
                const double tt1 = wallNow();
                mapLoop(tctx[t], Ms[t], t, nThreads, ls);
                if (prof)
                    fprintf(stderr, "[map thread %zu] context + index %.1f ms, loop %.1f (its own clock %.1f), from the loops' start to this thread's end %.1f\n", t, 1e3 * (tt1 - tt0),
                            1e3 * (wallNow() - tt1), 1e3 * ls.wall, 1e3 * (wallNow() - tLoop0));
            });
        }
        for (auto& t : th) t.join();
    }
    const double tJoin = wallNow();
    for (size_t t = 1; t < nThreads; t++)
        if (tctx[t]) dp_ctx_destroy(tctx[t]);
    if (prof) fprintf(stderr, "[map end] threads joined at %.1f ms after the loops' start, their contexts destroyed in %.1f\n", 1e3 * (tJoin - tLoop0), 1e3 * (wallNow() - tJoin));
    for (size_t t = 0; t < nThreads; t++)
        if (lstats[t].rc != 0) {
            error = lstats[t].error;
            dp_ctx_destroy(ctx), ctx = nullptr;
            return lstats[t].rc;
        }
    if (stats) {
        double tScan = 0, tChain = 0;
        for (const LoopStats& ls : lstats) {
            stats->k_scan_ms += ls.st.k_scan_ms;
            stats->k_map_ms += ls.st.k_map_ms;
            stats->map_bytes += ls.st.map_bytes;
            stats->scan_bytes += ls.st.scan_bytes;
            stats->n_windows += ls.st.n_windows;
            stats->n_chains += ls.st.n_chains;
            stats->n_batches += ls.st.n_batches;
            tScan += ls.tScan / (double)nThreads;   // (means over the threads, which run side by side)
            tChain += ls.tChain / (double)nThreads;
        }
        stats->t_scan_s = tScan;
        stats->t_chain_s = tChain;
        stats->t_host_s = (wallNow() - tLoop0) - tScan - tChain;
    }
    const double tText0 = wallNow();
    for (size_t r = 0; r < reads.size(); r++) {
        paf += out[r];
        if (nmaps[r] > 0) {
            if (nmaps[r] == 1) mapped++;
            else multiple++;
            total += nmaps[r];
        } else {
            unmapped++;
        }
    }
    char line[160];
    snprintf(line, sizeof line, "Uniquely mapped: %lld\nMultiple mappings: %lld\ntotal: %lld\nUnmapped: %lld\n", (long long)mapped,
             (long long)multiple, (long long)total, (long long)unmapped);
    errText += line;
    const double tEnd0 = wallNow();
    dp_ctx_destroy(ctx), ctx = nullptr;
    if (prof) fprintf(stderr, "[map end] text joined in %.1f ms, context destroyed in %.1f, whole run %.1f\n", 1e3 * (tEnd0 - tText0), 1e3 * (wallNow() - tEnd0), 1e3 * (wallNow() - tRun0));
    return 0;
}

// ---- self-test of the coroutine switch (include/downpore_host.h, test hooks): n_tasks coroutines on recycled stacks, every one adds
// its number `yields` times with a switch back to the caller in between and touches its stack deeply; returns the sum (-1: a task
// was resumed with a damaged frame).  What the mapper does with its read tasks, without a GPU - and under the sanitizer builds.
namespace {
struct CoroTestTask {
    CoroPoint at, *main = nullptr;
    int id = 0, yields = 0;
    long sum = 0;
    bool done = false, bad = false;
};
void coroTestBody(void* arg) {
    CoroTestTask* t = (CoroTestTask*)arg;
    volatile char pad[8192];
    for (int y = 0; y < t->yields; y++) {
        for (size_t i = 0; i < sizeof(pad); i += 64) pad[i] = (char)(t->id + y);
        coroSwitch(t->at, *t->main);
        for (size_t i = 0; i < sizeof(pad); i += 64)
            if (pad[i] != (char)(t->id + y)) t->bad = true;
        t->sum += t->id;
    }
    t->done = true;
    coroSwitch(t->at, *t->main, true);
}
}  // namespace
long coroSelfTest(int nTasks, int yields) {
    const size_t stackBytes = (size_t)64 << 10;
    CoroPoint main;
    std::vector<std::unique_ptr<char[]>> store;
    std::vector<char*> freeStacks;
    long sum = 0;
    const int wave = 7;  // tasks alive at a time: stacks are reused by later tasks
    for (int base = 0; base < nTasks; base += wave) {
        std::vector<std::unique_ptr<CoroTestTask>> live;
        std::vector<char*> mine;
        for (int i = base; i < std::min(nTasks, base + wave); i++) {
            if (freeStacks.empty()) {
                store.emplace_back(new char[stackBytes]);
                freeStacks.push_back(store.back().get());
            }
            live.emplace_back(new CoroTestTask());
            CoroTestTask& t = *live.back();
            t.main = &main;
            t.id = i + 1;
            t.yields = yields;
            mine.push_back(freeStacks.back());
            freeStacks.pop_back();
            coroStart(t.at, mine.back(), stackBytes, &coroTestBody, &t);
            coroSwitch(main, t.at);
        }
        for (bool any = true; any;) {
            any = false;
            for (auto& t : live)
                if (!t->done) {
                    coroSwitch(main, t->at);
                    any = true;
                }
        }
        for (size_t i = 0; i < live.size(); i++) {
            if (live[i]->bad) return -1;
            sum += live[i]->sum;
            coroRelease(live[i]->at);
            freeStacks.push_back(mine[i]);
        }
    }
    return sum;
}

}  // namespace dph
