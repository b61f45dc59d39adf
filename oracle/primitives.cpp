// ORACLE — test infrastructure only (see oracle.hpp).  L1/L0 primitives:
// sequence/sequence.go + sequence/asm_amd64.s, util/bitset.go + util/asm_amd64.s.
#include "oracle.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace dpo {

uint8_t baseCode(uint8_t b) { return (uint8_t)(((b >> 1) ^ ((b & 4) >> 2)) & 3); }

// 8-byte load + BSWAPQ (asm_amd64.s:14-15 and every "MOVQ (AX),R; BSWAPQ R")
static inline u64 load8be(const uint8_t* p) {
    u64 v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}

// sequence/asm_amd64.s:33-78.  Do-while over groups of four input bytes (caller guarantees n>=4).
void packBytesAsm(const uint8_t* s, size_t n, uint8_t* out) {
    i64 r8 = (i64)n;
    do {
        uint8_t d = (uint8_t)((baseCode(s[0]) << 6) | (baseCode(s[1]) << 4) | (baseCode(s[2]) << 2) | baseCode(s[3]));
        *out++ = d;
        s += 4;
        r8 -= 4;
    } while (r8 >= 4);
}

// sequence/sequence.go:67-93
PackedSeq newPackedSequence(i64 id, const std::string& seq, std::shared_ptr<std::string> name) {
    size_t len = seq.size();
    size_t length = len / 4;
    size_t internalLength = length * 4;
    size_t finalLength = len - internalLength;
    size_t nb = (len + 3) / 4;
    auto buf = std::make_shared<std::vector<uint8_t>>(nb + 16, 0);  // +16: zero pad for over-reads
    if (internalLength >= 4) packBytesAsm((const uint8_t*)seq.data(), internalLength, buf->data());
    if (finalLength > 0) {
        uint8_t b = 0;
        for (size_t i = finalLength; i > 0; i--) {
            uint8_t nbq = baseCode((uint8_t)seq[len - i]);
            b = (uint8_t)((b << 2) | nbq);
        }
        if (finalLength < 4) b = (uint8_t)(b << (8 - finalLength * 2));
        (*buf)[nb - 1] = b;
    }
    PackedSeq s;
    s.buf = buf;
    s.lo = 0;
    s.hi = nb;
    s.id = id;
    s.name = name;
    s.length = (i64)len;
    s.firstLen = 4;
    s.finalLen = (int)finalLength;
    if (s.finalLen > s.length) s.finalLen = (int)s.length;
    return s;
}

std::string PackedSeq::getName() const {
    if (!name) return std::to_string(id);
    return *name;
}

// sequence/sequence.go:242-276
std::string PackedSeq::str() const {
    size_t nb = nbytes();
    std::string out(nb * 4, 'A');
    static const char L[4] = {'A', 'C', 'G', 'T'};
    int j = firstLen * 2 - 2;
    size_t count = 0;
    const uint8_t* d = data();
    for (size_t bi = 0; bi + 1 < nb; bi++) {
        uint8_t b = d[bi];
        while (j >= 0) {
            out[count++] = L[(b >> j) & 3];
            j -= 2;
        }
        j = 6;
    }
    if (nb > 0) {
        uint8_t b = d[nb - 1];
        int last = 8 - finalLen * 2;
        if (last == 8) last = 0;
        while (j >= last) {
            out[count++] = L[(b >> j) & 3];
            j -= 2;
        }
    }
    return out.substr(0, (size_t)length);
}

// sequence/sequence.go:353-370
PackedSeq PackedSeq::subSequence(i64 start, i64 end) const {
    if (end > length) end = length;
    end--;
    i64 off = start + 4 - firstLen;
    i64 offByte = off / 4;
    off -= offByte * 4;
    i64 in = end + 4 - firstLen;
    i64 inByte = in / 4;
    in -= inByte * 4;
    PackedSeq ss;
    ss.buf = buf;
    ss.lo = lo + (size_t)offByte;
    ss.hi = lo + (size_t)inByte + 1;
    ss.id = id;
    ss.offset = offset + start;
    ss.inset = inset + length - end;
    ss.name = name;
    ss.firstLen = (int)(4 - off);
    ss.finalLen = (int)(in + 1);
    ss.length = end - start + 1;
    if (qual) {  // :366-368 quality[start : end+1]
        ss.qual = qual;
        ss.qlo = qlo + (size_t)start;
    }
    return ss;
}

// sequence/sequence.go:179-198
PackedSeq PackedSeq::reverseComplement() const {
    size_t nb = nbytes();
    auto nbuf = std::make_shared<std::vector<uint8_t>>(nb + 16, 0);
    const uint8_t* d = data();
    for (size_t i = 0; i < nb; i++) {
        uint8_t b = (uint8_t)~d[i];
        (*nbuf)[nb - 1 - i] = (uint8_t)(((b & 3) << 6) | ((b & 12) << 2) | ((b & 48) >> 2) | ((b & 192) >> 6));
    }
    PackedSeq rc;
    rc.buf = nbuf;
    rc.lo = 0;
    rc.hi = nb;
    rc.id = id;
    rc.offset = inset;
    rc.inset = offset;
    rc.firstLen = finalLen;
    rc.finalLen = firstLen;
    rc.name = name;
    rc.length = length;
    if (qual) {  // :188-195 reversed copy
        auto q = std::make_shared<std::vector<uint8_t>>((size_t)length);
        for (i64 i = 0; i < length; i++) (*q)[(size_t)(length - 1 - i)] = (*qual)[qlo + (size_t)i];
        rc.qual = q;
        rc.qlo = 0;
    }
    return rc;
}

// sequence/sequence.go:164-177
PackedSeq PackedSeq::append(i64 nid, const PackedSeq& other) const {
    std::string s = str() + other.str();
    PackedSeq seq = newPackedSequence(nid, s, nullptr);
    seq.offset = offset;
    seq.inset = other.inset;
    return seq;
}

// sequence/sequence.go:440-442 + asm_amd64.s:3-30 (returns int32)
i64 PackedSeq::kmerAt(i64 index, int k) const {
    i64 off = index + 4 - firstLen;
    u64 cx = (u64)off & 3;
    u64 bx = (u64)off >> 2;
    u64 ax = load8be(data() + bx);
    ax <<= (cx << 1);
    ax >>= (64 - 2 * k);
    return (i64)(int32_t)(uint32_t)ax;
}

// sequence/sequence.go:447-453
i64 PackedSeq::nextKmer(i64 cur, i64 mask, i64 nextBaseIndex) const {
    nextBaseIndex += 4 - firstLen;
    uint8_t b = data()[nextBaseIndex / 4];
    unsigned sub = (unsigned)(3 - (nextBaseIndex & 3)) << 1;
    b = (uint8_t)((b >> sub) & 3);
    return ((cur << 2) | (i64)b) & mask;
}

// sequence/asm_amd64.s:81-203
i64 packedCountKmersAsm(const uint8_t* ax, i64 nbytes, i64 upTo, i64 skipFront, i64 skipBack, int k,
                        const uint8_t* seeds) {
    i64 r8 = nbytes;
    r8 -= 1;
    r8 <<= 2;
    r8 -= skipBack;
    r8 -= k;
    r8 += 1;
    i64 r15 = r8 & 3;
    r8 &= ~(i64)3;
    const unsigned cx = (unsigned)(64 - 2 * k);
    u64 r9 = 0;
    u64 r10 = load8be(ax);
    i64 bx = skipFront << 1;
    r10 <<= (bx & 63);
    do {  // initial:
        u64 r12 = r10 >> cx;
        r9 = (r9 & ~(u64)0xFF) | (u64)(uint8_t)((uint8_t)r9 + seeds[r12]);  // ADDB
        r10 <<= 2;
        bx += 2;
    } while (bx <= 6);
    bool ended = false;
    do {  // internal:
        ax += 1;
        u64 r14 = load8be(ax);
        uint8_t s0 = seeds[r14 >> cx];
        uint8_t s1 = seeds[(r14 << 2) >> cx];
        uint8_t s2 = seeds[(r14 << 4) >> cx];
        uint8_t s3 = seeds[(r14 << 6) >> cx];
        uint8_t sum = (uint8_t)(s0 + s1 + s2 + s3);
        r9 += sum;
        if ((i64)r9 >= upTo) {
            ended = true;
            break;
        }
        r8 -= 4;
    } while (r8 >= 4);
    if (!ended) {
        ax += 1;
        r10 = load8be(ax);
        while (r15 != 0) {  // tail:
            r9 += seeds[r10 >> cx];
            r10 <<= 2;
            r15 -= 1;
        }
    }
    return (i64)r9;
}

// sequence/asm_amd64.s:206-394
i64 packedWriteSegmentsAsm(const uint8_t* ax, i64 nbytes, i64 skipFront, i64 skipBack, int k,
                           const uint8_t* seeds, i64* out) {
    i64* r14 = out;
    const i64 x2 = -(i64)k;
    i64 r8 = nbytes;
    r8 -= 1;
    r8 <<= 2;
    r8 -= skipBack;
    r8 -= k;
    r8 += 1;
    i64 r15 = r8 & 3;
    r8 &= ~(i64)3;
    const unsigned cx = (unsigned)(64 - 2 * k);
    i64 r9 = 0;  // the running gap
    u64 r10 = load8be(ax);
    i64 bx = skipFront << 1;
    r10 <<= (bx & 63);
    auto probe = [&](u64 kmer) {
        if (seeds[kmer]) {
            r14[0] = r9;
            r14[1] = (i64)kmer;
            r14 += 2;
            r9 = x2;
        }
        r9 += 1;
    };
    do {  // initial:
        probe(r10 >> cx);
        r10 <<= 2;
        bx += 2;
    } while (bx <= 6);
    do {  // internal:
        ax += 1;
        u64 x1 = load8be(ax);
        probe(x1 >> cx);
        probe((x1 << 2) >> cx);
        probe((x1 << 4) >> cx);
        probe((x1 << 6) >> cx);
        r8 -= 4;
    } while (r8 >= 4);
    ax += 1;
    r10 = load8be(ax);
    while ((int32_t)r15 != 0) {  // tail:
        probe(r10 >> cx);
        r10 <<= 2;
        r15 -= 1;
    }
    r9 = r9 - x2 - 1;  // endtail: "turn back into bases"
    r14[0] = r9;
    r14 += 1;
    return (i64)(r14 - out);
}

i64 PackedSeq::countKmers(i64 upTo, int k, const uint8_t* seeds) const {
    return packedCountKmersAsm(data(), (i64)nbytes(), upTo, 4 - firstLen, 4 - finalLen, k, seeds);
}
// sequence/sequence.go:332-337
i64 PackedSeq::countKmersBetween(i64 from, i64 to, i64 upTo, int k, const uint8_t* seeds) const {
    i64 start = (from + 4 - firstLen + 3) / 4;
    i64 end = (to + 4 - firstLen) / 4;
    return packedCountKmersAsm(data() + start, end - start, upTo, 4 - firstLen, 4 - finalLen, k, seeds);
}
void PackedSeq::writeSegments(i64* segments, int k, const uint8_t* seeds) const {
    packedWriteSegmentsAsm(data(), (i64)nbytes(), 4 - firstLen, 4 - finalLen, k, seeds, segments);
}

// byteSequence.CountKmers / WriteSegments (sequence/sequence.go:278-324) — the reference's own
// differential oracle for the packed path in sequence_test.go.
static i64 byteKmerAt(const std::string& s, i64 index, int k) {
    i64 v = 0;
    for (i64 i = index; i < index + k; i++) v = (v << 2) | baseCode((uint8_t)s[(size_t)i]);
    return v;
}
i64 byteCountKmers(const std::string& s, i64 upTo, int k, const uint8_t* seeds) {
    i64 mask = ((i64)1 << (2 * k)) - 1;
    i64 seed = byteKmerAt(s, 0, k) >> 2;
    i64 count = 0;
    for (i64 i = k - 1; i < (i64)s.size(); i++) {
        seed = ((seed << 2) | baseCode((uint8_t)s[(size_t)i])) & mask;
        if (seeds[seed]) {
            count++;
            if (count >= upTo) break;
        }
    }
    return count;
}
i64 byteWriteSegments(const std::string& s, int k, const uint8_t* seeds, i64* segments) {
    i64 mask = ((i64)1 << (2 * k)) - 1;
    i64 seed = byteKmerAt(s, 0, k) >> 2;
    i64 kmerIndex = 0, prev = 0, count = 0;
    for (i64 i = k - 1; i < (i64)s.size(); i++) {
        seed = ((seed << 2) | baseCode((uint8_t)s[(size_t)i])) & mask;
        if (seeds[seed]) {
            segments[count] = kmerIndex - prev;
            segments[count + 1] = seed;
            prev = kmerIndex + k;
            count += 2;
        }
        kmerIndex++;
    }
    segments[count] = (i64)s.size() - prev;
    return count + 1;
}

// ---------------------------------------------------------------------------------------------
// util/bitset.go

IntSet IntSet::fromUInts(const std::vector<u64>& values) {  // :43-57
    u64 mx = 0;
    for (u64 v : values)
        if (v > mx) mx = v;
    IntSet s;
    s.vs.assign((size_t)(mx / 64 + 1), 0);
    s.start = mx / 64;
    s.end = 0;
    s.count = 0;
    for (u64 v : values) s.add(v);
    return s;
}

bool IntSet::contains(u64 x) const {  // :65-72
    u64 index = x >> 6;
    if (index < start || index > end) return false;
    return (vs[(size_t)index] & ((u64)1 << (x & 0x3F))) != 0;
}

void IntSet::add(u64 x) {  // :74-108
    u64 index = x >> 6;
    u64 bit = (u64)1 << (x & 0x3F);
    if ((i64)index >= (i64)vs.size()) vs.resize((size_t)index + 2, 0);
    if (end < start) {
        start = index;
        end = index;
        vs[(size_t)index] = bit;
        count = 1;
        return;
    }
    if (index < start) {
        start = index;
        vs[(size_t)index] = bit;
        count++;
        return;
    }
    if (index > end) {
        end = index;
        vs[(size_t)index] = bit;
        count++;
        return;
    }
    u64 old = vs[(size_t)index];
    if (old & bit) return;
    vs[(size_t)index] = old | bit;
    count++;
}

void IntSet::clear() {  // :145-153
    while (start <= end) {
        vs[(size_t)start] = 0;
        start++;
    }
    end = 0;
    start = (u64)vs.size() + 1;
    count = 0;
}

u64 IntSet::countIntersection(const IntSet& o) const {  // :163-177
    u64 s = start, e = end;
    if (o.start > s) s = o.start;
    if (e > o.end) e = o.end;
    u64 c = 0;
    for (; s <= e; s++) c += (u64)__builtin_popcountll(vs[(size_t)s] & o.vs[(size_t)s]);
    return c;
}

// util/bitset.go:179-195 + util/asm_amd64.s:14-117
u64 IntSet::countIntersectionTo(const IntSet& o, i64 maxCount) const {
    u64 s = start, e = end;
    if (o.start > s) s = o.start;
    if (e > o.end) e = o.end;
    if (s > e + 1) throw std::runtime_error("oracle: CountIntersectionTo slice bounds (reference would panic)");
    const u64* a = vs.data() + s;
    const u64* b = o.vs.data() + s;
    i64 cx = (i64)(e + 1 - s);
    i64 dx = 0;
    for (;;) {  // docount:
        if (cx <= 7) break;
        if (dx >= maxCount) return (u64)dx;
        for (int i = 0; i < 8; i++) dx += __builtin_popcountll(a[i] & b[i]);
        a += 8;
        b += 8;
        cx -= 8;
    }
    while (cx > 0) {  // tail:
        dx += __builtin_popcountll(a[0] & b[0]);
        a++;
        b++;
        cx--;
    }
    return (u64)dx;
}

// util/asm_amd64.s:121-193.  v1..v4.
void softUnion4(const u64* vs, i64 n, u64 out[4]) {
    u64 v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    for (i64 j = 0; j < n; j++) {  // the n>=4 unroll is arithmetically the same ladder from zeros
        u64 m = vs[j];
        v4 |= v3 & m;
        v3 |= v2 & m;
        v2 |= v1 & m;
        v1 |= m;
    }
    out[0] = v1;
    out[1] = v2;
    out[2] = v3;
    out[3] = v4;
}

// util/asm_amd64.s:196-314.  v5..v8.  (n<=5: reference leaves v1..v4 uninitialised; zero here.)
void softUnion8(const u64* vs, i64 n, u64 out[4]) {
    u64 v[9] = {0};
    for (i64 j = 0; j < n; j++) {
        u64 m = vs[j];
        for (int t = 8; t >= 2; t--) v[t] |= v[t - 1] & m;
        v[1] |= m;
    }
    out[0] = v[5];
    out[1] = v[6];
    out[2] = v[7];
    out[3] = v[8];
}

// util/asm_amd64.s:317-509.  v13..v16.  The 8th gathered word (asm:407-428) updates v2..v8 but
// NOT v1 — reproduced.  Caller guarantees n>=8 (n>=minCount>=13, bitset.go:338-342).
void softUnion16(const u64* vs, i64 n, u64 out[4]) {
    u64 v[17] = {0};
    for (i64 j = 0; j < n; j++) {
        u64 m = vs[j];
        for (int t = 16; t >= 2; t--) v[t] |= v[t - 1] & m;
        if (j != 7) v[1] |= m;
    }
    out[0] = v[13];
    out[1] = v[14];
    out[2] = v[15];
    out[3] = v[16];
}

// util/bitset.go:509-538
static void addSoftUnionIDs(u64 v, const u64* vs, i64 n, i64 minCount, std::vector<u64>& ids, u64 offset) {
    u64 bit = 1;
    u64 zs = (u64)__builtin_ctzll(v);
    bit <<= zs;
    v >>= zs;
    for (u64 j = zs; j < 64 && v != 0; j++) {
        if (v & 1) {
            i64 count = 0;
            for (i64 k = 0; k < n; k++) {
                if (vs[k] & bit) {
                    count++;
                    if (count >= minCount) {
                        ids.push_back(offset + j);
                        break;
                    }
                } else if (n - k + count <= minCount) {
                    break;
                }
            }
        }
        v >>= 1;
        bit <<= 1;
    }
}

// util/bitset.go:308-411
std::vector<u64> getSharedIDs(const std::vector<const IntSet*>& sets, i64 minCount, bool fast) {
    std::vector<u64> ids;
    if (minCount > 24) fast = false;
    i64 n = (i64)sets.size();
    u64 start = (u64)sets[0]->vs.size();
    u64 end = 0;
    std::vector<u64> lens((size_t)n);
    std::vector<const u64*> vs((size_t)n);
    u64 shortest = start;
    for (i64 i = 0; i < n; i++) {
        vs[(size_t)i] = sets[(size_t)i]->vs.data();
        lens[(size_t)i] = sets[(size_t)i]->end + 1;
        if (sets[(size_t)i]->start < start) start = sets[(size_t)i]->start;
        if (sets[(size_t)i]->end > end) end = sets[(size_t)i]->end;
        if (lens[(size_t)i] < shortest) shortest = lens[(size_t)i];
    }
    std::vector<u64> nextVs((size_t)n);
    for (u64 i = start; i <= end; i++) {
        if (shortest <= i) {
            u64 nextShortest = end;
            for (i64 j = 0; j < n; j++) {
                if (lens[(size_t)j] <= i) {
                    i64 last = n - 1;
                    if (last < minCount) return ids;
                    vs[(size_t)j] = vs[(size_t)last];
                    lens[(size_t)j] = lens[(size_t)last];
                    n = last;
                    j--;
                } else if (lens[(size_t)j] < nextShortest) {
                    nextShortest = lens[(size_t)j];
                }
            }
            shortest = nextShortest;
        }
        for (i64 j = 0; j < n; j++) nextVs[(size_t)j] = vs[(size_t)j][i];
        u64 v = 0;
        u64 r[4];
        if (minCount >= 13) {
            softUnion16(nextVs.data(), n, r);
            if (minCount >= 16) v = r[3];
            else if (minCount == 15) v = r[2];
            else if (minCount == 14) v = r[1];
            else v = r[0];
        } else if (minCount >= 5) {
            softUnion8(nextVs.data(), n, r);
            if (minCount >= 8) v = r[3];
            else if (minCount == 7) v = r[2];
            else if (minCount == 6) v = r[1];
            else v = r[0];
        } else {
            softUnion4(nextVs.data(), n, r);
            if (minCount == 4) v = r[3];
            else if (minCount == 3) v = r[2];
            else if (minCount == 2) v = r[1];
            else v = r[0];
        }
        if (v != 0) {
            if (fast) {
                u64 shifted = 0;
                while (v != 0) {
                    u64 zs = (u64)__builtin_ctzll(v);
                    ids.push_back((i << 6) + shifted + zs);
                    // Go: v >> 64 == 0 for unsigned shifts
                    v = (zs + 1 >= 64) ? 0 : (v >> (zs + 1));
                    shifted += zs + 1;
                }
            } else {
                addSoftUnionIDs(v, nextVs.data(), n, minCount, ids, i << 6);
            }
        }
    }
    return ids;
}

}  // namespace dpo
