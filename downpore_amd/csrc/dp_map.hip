// libdownpore_hip.so — map-flavour query (A19 + A20): the candidate loop of mapping.performMapping
// (mapping/mapping.go:494-589) for batches of (forward, reverse-complement) window pairs.  CDNA4 / gfx950 only.
//
// Per pair, one persistent wave:
//   Matches(0.25) for both windows comes from the shared index-query stage (query_kernel, dp_overlap.hip).
//   For each candidate chunk (ascending): exact-intersection prefilter (wave-parallel popcount), then
//   SeedSequence.Match (seeds/sequence.go:361-394) on lane 0: Reduced(target | query seeds), Reduced(query | target
//   seeds), dynamicMatch (:401-471) with extendChain (:476-576), indices mapped back through the reduction maps.
//   Every returned chain whose unmatched query flanks are <= 2/3 of the window is emitted and ratchets
//   minMatches / minRCMatches by 4/5 of its length (:536-549, 576-586); the forward ratchet also raises the reverse
//   threshold.
// LDS per wave: reduced query/target segments + index maps + per-query-seed chain heads.  Chain storage (Go slices
// sharing backing arrays) lives in an HBM scratch pool: one fixed-stride slot per started chain.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "dp_common.h"

typedef uint64_t u64;

#define M_WAVES 4
#define M_QMAX 256        // reduced query seeds held in LDS
#define M_TMAX 2048       // reduced target seeds held in LDS
#define M_CHAINS 1024     // chains that may be started per candidate (pool slots per wave)
#define M_GOOD 512

struct MWave {
    int32_t q[2 * M_QMAX + 1];
    int32_t t[2 * M_TMAX + 1];
    uint16_t qIdx[M_QMAX];
    uint16_t tIdx[M_TMAX];
    int32_t headChain[M_QMAX];  // chain slot recorded for each reduced query seed (-1 none)
    uint16_t headLen[M_QMAX];   // its length at that seed (the Go slice header's len)
    int32_t good[M_GOOD];
    __device__ __forceinline__ int qmax() const { return M_QMAX; }
    __device__ __forceinline__ int tmax() const { return M_TMAX; }
};
// Round 6: the same working set in global memory, for a batch in which a reduced window or a reduced chunk is larger than MWave's
// arrays (`map -query_size 8000 -seed_rate 10`: ~800 seeds a window; SeedSequence.Match has no such limit, seeds/sequence.go:361-394):
// the kernel's BIG variant, run again over the batch when the ordinary one reports the capacity (a wave's slice is sized from the
// batch's longest window and the index's longest chunk).
struct MBig {
    int32_t* q;
    int32_t* t;
    uint16_t* qIdx;
    uint16_t* tIdx;
    int32_t* headChain;
    uint16_t* headLen;
    int32_t good[M_GOOD];
    int qcap, tcap;
    __device__ __forceinline__ int qmax() const { return qcap; }
    __device__ __forceinline__ int tmax() const { return tcap; }
};
// 4-byte words of a wave's slice of the BIG workspace
static __host__ __device__ inline size_t m_big_words(size_t qcap, size_t tcap) {
    return (2 * qcap + 2) + (2 * tcap + 2) + (qcap + 2) / 2 + (tcap + 2) / 2 + (qcap + 2) + (qcap + 2) / 2 + 8;
}

__device__ __forceinline__ bool m_contains(const u64* __restrict__ set, int32_t x) { return (set[x >> 6] >> (x & 63)) & 1ull; }

// SeedSequence.Reduced (seeds/sequence.go:85-123).  Returns the number of reduced seeds or -1 if < minSeeds.
// err bit 1: LDS capacity.
template <typename IDX>
__device__ int m_reduce(const int32_t* __restrict__ seg, int n, const u64* __restrict__ whitelist, int k, int minSeeds,
                        int32_t* out, IDX* index, int cap, uint32_t* err) {
    int count = 0, prev = -1;
    for (int i = 1; i < n; i += 2) {
        const int next = seg[i];
        if (next != prev && m_contains(whitelist, next)) {
            count++;
            prev = next;
        }
    }
    if (count < minSeeds) return -1;
    if (count > cap) {
        *err |= 1;
        return -1;
    }
    int offset = seg[0];
    prev = -1;
    int j = 0;
    for (int i = 1; i < n; i += 2) {
        const int seed = seg[i];
        if (prev != seed && m_contains(whitelist, seed)) {
            out[j] = offset;
            out[j + 1] = seed;
            index[j / 2] = (IDX)(i / 2);
            j += 2;
            offset = seg[i + 1];
            prev = seed;
        } else {
            offset += seg[i + 1] + k;
        }
    }
    out[j] = offset;
    return count;
}

// The same on the whole wave: lane i takes seed base + i of every 64-seed piece; its whitelist probe is one of 64 in flight (on one
// lane the probes of a 250-seed chunk were 2 x 250 dependent global loads - half a millisecond per candidate).  A seed is kept when
// it is whitelisted and differs from the last KEPT seed; a whitelisted seed is dropped only when it equals the last kept one, which
// then stays what it was - so "differs from the previous whitelisted seed" decides, and all lanes decide at once.  The gap written
// before kept seed a (previous kept seed p, or -1) is sum_{t=p+1..a} gap_t + k (a - p - 1): prefix sums of the gaps.
template <typename IDX>
__device__ int m_reduce_wave(const int32_t* __restrict__ seg, int n, const u64* __restrict__ whitelist, int k, int minSeeds, int32_t* out,
                             IDX* index, int cap, uint32_t* err) {
    const int lane = dp_lane();
    const u64 below = (1ull << lane) - 1ull;
    const int nS = n >> 1;
    int kept = 0, carrySeed = -1, prevKept = -1;  // seeds kept so far, last whitelisted seed, index of the last kept seed
    long long gRun = 0, gPrevKept = 0;            // sum of gap_0 .. gap_{base-1}, prefix sum at the last kept seed
    bool over = false;
    for (int base = 0; base < nS; base += 64) {
        const int sI = base + lane;
        const bool valid = sI < nS;
        const int seed = valid ? seg[2 * sI + 1] : -1;
        const int gap = valid ? seg[2 * sI] : 0;
        const bool c = valid && m_contains(whitelist, seed);
        const u64 cmask = __ballot(c);
        const u64 cb = cmask & below;
        const int srcC = cb ? 63 - __builtin_clzll(cb) : 0;
        const int fromC = __shfl(seed, srcC, 64);
        const int prevC = cb ? fromC : carrySeed;
        const bool keep = c && seed != prevC;
        const u64 kmask = __ballot(keep);
        const long long G = gRun + (long long)wave_incl_sum(gap);  // inclusive prefix sum of the gaps up to this seed
        // the previous kept seed: in this piece (a lane below), or carried over
        const u64 kb = kmask & below;
        const int srcK = kb ? 63 - __builtin_clzll(kb) : 0;
        const long long gFrom = __shfl((long long)G, srcK, 64);
        const int pIdx = kb ? base + srcK : prevKept;
        const long long gP = kb ? gFrom : gPrevKept;
        if (keep) {
            const int j = kept + __popcll(kb);
            if (j < cap) {
                out[2 * j] = (int32_t)(G - gP + (long long)k * (sI - pIdx - 1));
                out[2 * j + 1] = seed;
                index[j] = (IDX)sI;
            } else {
                over = true;
            }
        }
        kept += __popcll(kmask);
        if (cmask) carrySeed = __shfl(seed, 63 - __builtin_clzll(cmask), 64);
        if (kmask) {
            const int lk = 63 - __builtin_clzll(kmask);
            prevKept = base + lk;
            gPrevKept = __shfl((long long)G, lk, 64);
        }
        gRun = __shfl((long long)G, 63, 64);
    }
    if (kept < minSeeds) return -1;
    if (__ballot(over) || kept > cap) {
        *err |= 1;
        return -1;
    }
    if (lane == 0) {  // final gap: everything after the last kept seed
        const long long gEnd = gRun + (long long)seg[2 * nS];
        out[2 * kept] = (int32_t)(gEnd - gPrevKept + (long long)k * (nS - 1 - prevKept));
    }
    __builtin_amdgcn_wave_barrier();
    return kept;
}

struct MChainPool {
    uint16_t* a;  // [M_CHAINS][stride]: stride = the reduced query's capacity (M_QMAX, or the BIG variant's)
    uint16_t* b;
    uint32_t stride;
    __device__ __forceinline__ uint16_t* A(int c) const { return a + (size_t)c * stride; }
    __device__ __forceinline__ uint16_t* B(int c) const { return b + (size_t)c * stride; }
};

// extendChain (seeds/sequence.go:476-576); a = reduced query, b = reduced target.  Returns the chain's final length.
template <class LT>
__device__ int m_extend(LT& L, int an, int bn, int aIndex, int bIndex, int k, int cur, int curLen, const MChainPool& P) {
    const int32_t* as = L.q;
    const int32_t* bs = L.t;
    uint16_t* ca = P.A(cur);
    uint16_t* cb = P.B(cur);
    int offsetA = as[aIndex + 1], offsetB = bs[bIndex + 1];
    aIndex += 2;
    bIndex += 2;
    while (aIndex < an && bIndex < bn) {
        int aSeedIndex = aIndex / 2;
        int minBOffset, maxBOffset;
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
            if (aIndex >= an) return curLen;
            aSeedIndex = aIndex / 2;
            minBOffset = (offsetA * 2) / 3 - k;
            maxBOffset = (offsetA * 3) / 2 + k;
        }
        while (offsetB < minBOffset) {
            offsetB += bs[bIndex + 1] + k;
            bIndex += 2;
            if (bIndex >= bn) return curLen;
        }
        const int oldBIndex = bIndex, oldBOffset = offsetB;
        bool matched = false;
        const int seedA = as[aIndex];
        while (offsetB <= maxBOffset) {
            if (seedA == bs[bIndex]) {
                const int hc = L.headChain[aSeedIndex];
                if (hc >= 0) {
                    const int hl = L.headLen[aSeedIndex];
                    if (bIndex / 2 == (int)P.B(hc)[hl - 1] && hl > curLen) return curLen;  // they have a better chain already
                }
                ca[curLen] = (uint16_t)aSeedIndex;
                cb[curLen] = (uint16_t)(bIndex / 2);
                curLen++;
                L.headChain[aSeedIndex] = cur;
                L.headLen[aSeedIndex] = (uint16_t)curLen;
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
    return curLen;
}

// dynamicMatch (seeds/sequence.go:401-471).  seq = reduced target (L.t, sn ints), query = reduced query (L.q, qn ints).
// Fills L.good with chain slots (in the reference's allGoodChains order) and their lengths in goodLen; returns count.
// err bit 2: chain pool exhausted, bit 4: good list overflow.
template <class LT>
__device__ int m_dynamic_match(LT& L, int qn, int sn, int minMatch, int k, const MChainPool& P, uint16_t* chainLen,
                               uint32_t* err) {
    if (minMatch == 0) minMatch = 1;
    const int nq = qn / 2;
    for (int i = 0; i < nq; i++) L.headChain[i] = -1;
    int nChains = 0, nGood = 0;
    const int32_t* qs = L.q;
    const int32_t* ss = L.t;
    for (int qIndex = 1; qIndex < qn - minMatch * 2 + 2; qIndex += 2) {
        if (qs[qIndex - 1] < 0 && qIndex > 1 && qs[qIndex + 1] < 0 && qs[qIndex] == qs[qIndex - 2] && qs[qIndex] == qs[qIndex + 2])
            continue;
        const int qsi = qIndex / 2;
        if (L.headChain[qsi] >= 0) continue;
        int prevSeed = -1;
        for (int i = 1; i < sn - minMatch * 2 + 2; i += 2) {
            const int nextSeed = ss[i];
            const int hc = L.headChain[qsi];
            if (nextSeed == qs[qIndex] && nextSeed != prevSeed && (hc < 0 || (int)P.B(hc)[L.headLen[qsi] - 1] != i / 2)) {
                if (nChains >= M_CHAINS) {
                    *err |= 2;
                    return nGood;
                }
                const int c = nChains++;
                P.A(c)[0] = (uint16_t)qsi;
                P.B(c)[0] = (uint16_t)(i / 2);
                L.headChain[qsi] = c;
                L.headLen[qsi] = 1;
                const int len = m_extend(L, qn, sn, qIndex, i, k, c, 1, P);
                chainLen[c] = (uint16_t)len;
                if (len >= minMatch) {
                    const int nextLength = (len * 2) / 3;
                    if (nextLength > minMatch) {
                        minMatch = nextLength;
                        for (int j = nGood - 1; j >= 0; j--) {
                            if ((int)chainLen[L.good[j]] < nextLength) {
                                L.good[j] = L.good[nGood - 1];
                                nGood--;
                            }
                        }
                    }
                    if (nGood >= M_GOOD) {
                        *err |= 4;
                        return nGood;
                    }
                    L.good[nGood++] = c;
                    int remaining = 0;
                    for (int x = 0; x < nq; x++) remaining += L.headChain[x] < 0;
                    if (remaining < len) return nGood;
                }
            }
            prevSeed = nextSeed;
        }
    }
    return nGood;
}

// dynamicMatch with the whole wave (round 4).  The reference's loop nest is "for every query seed that heads no chain yet: for
// every target seed: equal?" - |q| x |t| probes (125 x 250 for a 250-seed reference chunk) of which a handful are hits; on one lane
// that scan WAS map_kernel's time (1.24 ms per launch).  Here the 64 lanes probe 64 target seeds at once (equal to the query seed
// and not a repeat of the target seed in front of it: both are facts of positions, not of the walk's state), and lane 0 handles
// the hits in ascending order exactly as the one-lane loop does - the head-chain test at the moment of the hit, extendChain, the
// 2 len / 3 ratchet (which also shortens both loops' bounds), the "fewer open query seeds than this chain is long" exit.  What
// lane 0 decides (chain count, good count, minMatch, stop) is broadcast after every hit, so every lane runs the same loops.
template <class LT>
__device__ int m_dynamic_match_wave(LT& L, int qn, int sn, int minMatch, int k, const MChainPool& P, uint16_t* chainLen,
                                    uint32_t* err) {
    const int lane = dp_lane();
    if (minMatch == 0) minMatch = 1;
    const int nq = qn / 2, nsT = sn / 2;
    for (int i = lane; i < nq; i += 64) L.headChain[i] = -1;
    __builtin_amdgcn_wave_barrier();
    int nChains = 0, nGood = 0;
    const int32_t* qs = L.q;
    const int32_t* ss = L.t;
    for (int qIndex = 1; qIndex < qn - minMatch * 2 + 2; qIndex += 2) {
        if (qs[qIndex - 1] < 0 && qIndex > 1 && qs[qIndex + 1] < 0 && qs[qIndex] == qs[qIndex - 2] && qs[qIndex] == qs[qIndex + 2])
            continue;
        const int qsi = qIndex / 2;
        if (L.headChain[qsi] >= 0) continue;
        const int qseed = qs[qIndex];
        for (int tb = 0; 2 * tb + 1 < sn - minMatch * 2 + 2; tb += 64) {
            const int t = tb + lane, i = 2 * t + 1;
            bool hit = false;
            if (t < nsT && i < sn - minMatch * 2 + 2) {
                const int sd = ss[i];
                hit = sd == qseed && (t == 0 || ss[i - 2] != sd);  // (prevSeed of the reference's scan = the target seed in front)
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int ti = 2 * (tb + j) + 1;
                if (ti >= sn - minMatch * 2 + 2) break;  // (a ratchet inside this stretch moved the bound in front of the hit)
                int stop = 0;
                if (lane == 0) {
                    const int hc = L.headChain[qsi];
                    if (hc < 0 || (int)P.B(hc)[L.headLen[qsi] - 1] != ti / 2) {
                        if (nChains >= M_CHAINS) {
                            *err |= 2;
                            stop = 1;
                        } else {
                            const int c = nChains++;
                            P.A(c)[0] = (uint16_t)qsi;
                            P.B(c)[0] = (uint16_t)(ti / 2);
                            L.headChain[qsi] = c;
                            L.headLen[qsi] = 1;
                            const int len = m_extend(L, qn, sn, qIndex, ti, k, c, 1, P);
                            chainLen[c] = (uint16_t)len;
                            if (len >= minMatch) {
                                const int nextLength = (len * 2) / 3;
                                if (nextLength > minMatch) {
                                    minMatch = nextLength;
                                    for (int g = nGood - 1; g >= 0; g--) {
                                        if ((int)chainLen[L.good[g]] < nextLength) {
                                            L.good[g] = L.good[nGood - 1];
                                            nGood--;
                                        }
                                    }
                                }
                                if (nGood >= M_GOOD) {
                                    *err |= 4;
                                    stop = 1;
                                } else {
                                    L.good[nGood++] = c;
                                    int remaining = 0;
                                    for (int x = 0; x < nq; x++) remaining += L.headChain[x] < 0;
                                    if (remaining < len) stop = 1;
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                nChains = __builtin_amdgcn_readfirstlane(nChains);
                nGood = __builtin_amdgcn_readfirstlane(nGood);
                minMatch = __builtin_amdgcn_readfirstlane(minMatch);
                stop = __builtin_amdgcn_readfirstlane(stop);
                if (stop) return nGood;
            }
        }
    }
    return nGood;
}

struct MapRec {
    uint32_t window, target, off, len, seq;
};

// GetSeedOffset / GetSeedOffsetFromEnd on the ORIGINAL window segments (seeds/sequence.go:1239-1276)
__device__ __forceinline__ int m_seed_offset(const int32_t* seg, int index, int k) {
    index = index * 2 + 1;
    int o = seg[0];
    for (int i = 2; i < index; i += 2) o += seg[i] + k;
    return o;
}
__device__ __forceinline__ int m_seed_offset_from_end(const int32_t* seg, int n, int index, int k) {
    index = index * 2 + 1;
    int o = seg[n - 1];
    for (int i = n - 3; i > index; i -= 2) o += seg[i] + k;
    return o;
}

// A 16-bit word that ANOTHER lane of this wave has just stored (lane 0 keeps dynamicMatch's chains in the wave's pool): through the L2 -
// a plain load may be served from a stale line of the CU's vector L1
__device__ __forceinline__ uint32_t m_ld16_agent(const uint16_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t w = __hip_atomic_load((const uint32_t*)(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (a & 2) ? (w >> 16) : (w & 0xffffu);
}

// cursor: [0] records, [1] ints, [2] error bits, [3] overflow flag, [8..9] algorithmic bytes
template <bool BIG>
__global__ __launch_bounds__(64 * M_WAVES) void map_kernel(const int32_t* __restrict__ wsegs, const u64* __restrict__ woff,
                                                           const uint32_t* __restrict__ wlen, uint32_t n_pairs,
                                                           const u64* __restrict__ wsets, const uint32_t* __restrict__ qmeta,
                                                           const u64* __restrict__ cand, const dp_seq_ref* __restrict__ refs,
                                                           const int32_t* __restrict__ segs, const u64* __restrict__ seedsets,
                                                           uint32_t W, uint32_t SW, int k, uint16_t* __restrict__ poolA,
                                                           uint16_t* __restrict__ poolB, uint16_t* __restrict__ poolLen,
                                                           MapRec* __restrict__ recs, uint32_t rec_cap, int32_t* __restrict__ ma,
                                                           int32_t* __restrict__ mb, uint32_t int_cap, uint32_t* __restrict__ cursor,
                                                           int phase, int32_t* __restrict__ thr_io, int one_lane,
                                                           const u64* __restrict__ words_read, unsigned long long* __restrict__ prof,
                                                           uint32_t* __restrict__ big_ws, uint32_t big_q, uint32_t big_t) {
    // phase 2: both strands of every window pair, thresholds from the windows themselves (the whole index is here).
    // phase 0 / 1 (the index is one shard of the reference, dp_map_windows_shard): only the forward / only the reverse-complement
    // windows, starting from the thresholds the previous shard left in thr_io[pair][2] (< 0: none yet) and leaving its own there.
    typedef typename std::conditional<BIG, MBig, MWave>::type LT;
    __shared__ LT sh[M_WAVES];
    LT& L = sh[threadIdx.x >> 6];
    const int lane = dp_lane();
    const uint32_t gw = blockIdx.x * M_WAVES + (threadIdx.x >> 6);
    const uint32_t pstride = BIG ? big_q : (uint32_t)M_QMAX;
    if constexpr (BIG) {
        if (lane == 0) {
            const size_t qc = big_q, tc = big_t;
            uint32_t* base = big_ws + (size_t)gw * m_big_words(qc, tc);
            L.q = (int32_t*)base;
            L.t = (int32_t*)(base + (2 * qc + 2));
            L.qIdx = (uint16_t*)(base + (2 * qc + 2) + (2 * tc + 2));
            L.tIdx = (uint16_t*)(base + (2 * qc + 2) + (2 * tc + 2) + (qc + 2) / 2);
            L.headChain = (int32_t*)(base + (2 * qc + 2) + (2 * tc + 2) + (qc + 2) / 2 + (tc + 2) / 2);
            L.headLen = (uint16_t*)(base + (2 * qc + 2) + (2 * tc + 2) + (qc + 2) / 2 + (tc + 2) / 2 + (qc + 2));
            L.qcap = (int)big_q;
            L.tcap = (int)big_t;
        }
        __builtin_amdgcn_wave_barrier();
    }
    MChainPool P;
    P.stride = pstride;
    P.a = poolA + (size_t)gw * M_CHAINS * pstride;
    P.b = poolB + (size_t)gw * M_CHAINS * pstride;
    uint16_t* chainLen = poolLen + (size_t)gw * M_CHAINS;
    // algorithmic bytes of this wave's windows: posting words the index query gathered for them, 2 x 8 x SW per
    // prefiltered candidate + 4 per candidate out, 8 per seed of both sides of every pair that is chained, 8 per chain link out
    unsigned long long algb = 0;
    // profiling build (make PROF=1, DP_MAP_PROF=1 prints them): ticks of 10 ns per phase, summed per wave, added to cursor[16 ..] at the end
#ifdef DP_PROF_BUILD
    unsigned long long mp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mlast = wall_clock64();
    const unsigned long long mstart = mlast;
#define MP_TICK(i_)                                                  \
    {                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        const unsigned long long n_ = wall_clock64();                \
        mp[i_] += n_ - mlast;                                        \
        mlast = n_;                                                  \
    }
#else
#define MP_TICK(i_)
#endif
    // (pairs dealt out by wave number.  A ticket - one returning atomic per pair on one word - was tried in round 5 to shorten the
    // launch's slowest wave, 305 us against a mean of 185: 4 096 same-address atomics per launch cost every pair 7 us and the slowest wave
    // got longer)
    const uint32_t waves = gridDim.x * M_WAVES;
    for (uint32_t pair = gw; pair < n_pairs; pair += waves) {
        // performMapping :494-501
        int thr[2];
        for (int s = 0; s < 2; s++) {
            const uint32_t w = 2 * pair + s;
            const int ns = (int)((woff[w + 1] - woff[w]) / 2);
            thr[s] = ns / 5 < 5 ? 5 : ns / 5;
            if (thr_io && thr_io[2 * pair + s] >= 0) thr[s] = thr_io[2 * pair + s];
        }
        uint32_t seq = 0;
        for (int s = (phase == 1 ? 1 : 0); s < (phase == 0 ? 1 : 2); s++) {
            const uint32_t w = 2 * pair + s;
            const int32_t* qSeg = wsegs + woff[w];
            const int qN = (int)(woff[w + 1] - woff[w]);
            const int qLen = (int)wlen[w];
            const u64* qset = wsets + (uint64_t)w * SW;
            if (qmeta[4 * w + 0] < 5 || qmeta[4 * w + 2]) continue;  // Matches() returned nothing
            if (words_read) algb += 8ull * words_read[w];
            for (uint32_t wi = 0; wi < W; wi++) {
                u64 mask = cand[(uint64_t)w * W + wi];
                while (mask) {
                    const int b = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const uint32_t t = wi * 64 + (uint32_t)b;
                    const u64* tset = seedsets + (uint64_t)t * SW;
                    int c = 0;
                    for (uint32_t x = lane; x < SW; x += 64) c += __popcll(tset[x] & qset[x]);
                    c = wave_sum(c);
                    algb += 16ull * SW + 4ull;
                    MP_TICK(0)  // candidate word walk + prefilter (two set rows per candidate)
                    if (c < thr[s]) continue;  // CountIntersectionTo(seedSet, min) < min (:521, :560)
                    const dp_seq_ref r = refs[t];
                    const int32_t* tSeg = segs + r.seg_off;
                    const int tN = (int)(2 * r.n_seeds + 1);
                    algb += 4ull * (unsigned long long)(qN + tN);
                    int thrS = thr[s], thrOther = thr[1 - s];
                    uint32_t err = 0;
                    // Match (:361-394): s = seq.Reduced(querySet), q = query.Reduced(seqSet) - on all 64 lanes
                    const int minMatch = thr[s];
                    const int nT = m_reduce_wave<uint16_t>(tSeg, tN, qset, k, minMatch, L.t, L.tIdx, L.tmax(), &err);
                    MP_TICK(1)  // Reduced() of the target chunk
                    const int nQ = nT < 0 ? -1 : m_reduce_wave<uint16_t>(qSeg, qN, tset, k, minMatch, L.q, L.qIdx, L.qmax(), &err);
                    MP_TICK(2)  // Reduced() of the query window
                    // dynamicMatch: the probes on 64 lanes, the walk's decisions on lane 0 (DP_MAP_ONE_LANE=1: all of it on lane 0 as
                    // before round 4)
                    int nGoodW = 0;
                    if (nT >= 0 && nQ >= 0 && !one_lane) nGoodW = m_dynamic_match_wave(L, 2 * nQ + 1, 2 * nT + 1, minMatch, k, P, chainLen, &err);
                    MP_TICK(3)  // dynamicMatch + extendChain
                    if (!one_lane) {
                        // records and chains out - by the whole wave (round 5: on lane 0 alone this was 42 us of a wave's 185: the two seed
                        // offsets of the flank test are sums over the query's gaps - up to a hundred dependent global loads each - and the
                        // chain went out element by element)
                        const int nsQ = qN >> 1;
                        // (lane 0 wrote the chains: its stores are made visible at the L2, the other lanes read them there)
                        // (a workgroup-scope fence: the stores have left the wave - an agent-scope release fence here writes the L2 back and made
                        // every phase of every wave 1.4 x as long)
                        if (nGoodW > 0) __threadfence_block();
                        for (int g = 0; g < nGoodW; g++) {
                            const int ch = L.good[g];
                            const int len = (int)m_ld16_agent(&chainLen[ch]);
                            const uint16_t* ca = P.A(ch);
                            const uint16_t* cb = P.B(ch);
                            const int a0 = L.qIdx[m_ld16_agent(&ca[0])], aL = L.qIdx[m_ld16_agent(&ca[len - 1])];
                            // GetSeedOffset(a0) = seg[0] + sum_{j=1..a0} (seg[2j] + k); GetSeedOffsetFromEnd(aL) = seg[n-1] + sum_{j=aL+1..ns-1} (seg[2j] + k)
                            int so = 0, se = 0;
                            for (int j = 1 + lane; j <= a0; j += 64) so += qSeg[2 * j] + k;
                            for (int j = aL + 1 + lane; j <= nsQ - 1; j += 64) se += qSeg[2 * j] + k;
                            const int qOffset = qSeg[0] + wave_sum(so);
                            const int qInset = qSeg[qN - 1] + wave_sum(se);
                            // qOffset+qInset > (Len*2)/3 -> skipped, and does not ratchet (:536-538, 576-578)
                            if (qOffset + qInset > (qLen * 2) / 3) continue;
                            uint32_t ri = 0, off = 0;
                            if (lane == 0) {
                                ri = atomicAdd(&cursor[0], 1u);
                                off = atomicAdd(&cursor[1], (uint32_t)len);
                            }
                            ri = (uint32_t)__shfl((int)ri, 0, 64);
                            off = (uint32_t)__shfl((int)off, 0, 64);
                            if (ri < rec_cap && off + (uint32_t)len <= int_cap) {
                                if (lane == 0) {
                                    MapRec rec = {w, t, off, (uint32_t)len, seq};
                                    recs[ri] = rec;
                                }
                                for (int x = lane; x < len; x += 64) {
                                    ma[off + x] = L.qIdx[m_ld16_agent(&ca[x])];
                                    mb[off + x] = L.tIdx[m_ld16_agent(&cb[x])];
                                }
                            } else if (lane == 0) {
                                cursor[3] = 1;
                            }
                            seq++;
                            algb += 8ull * (unsigned long long)len;
                            const int limit = (len * 4) / 5;
                            if (limit > thrS) thrS = limit;
                            if (s == 0 && limit > thrOther) thrOther = limit;  // forward also raises minRCMatches (:547-549)
                        }
                        if (lane == 0 && err) atomicOr(&cursor[2], err);
                    } else if (lane == 0) {
                        if (nT >= 0 && nQ >= 0) {
                            const int nGood = m_dynamic_match(L, 2 * nQ + 1, 2 * nT + 1, minMatch, k, P, chainLen, &err);
                            for (int g = 0; g < nGood; g++) {
                                const int ch = L.good[g];
                                const int len = chainLen[ch];
                                const uint16_t* ca = P.A(ch);
                                const uint16_t* cb = P.B(ch);
                                const int a0 = L.qIdx[ca[0]], aL = L.qIdx[ca[len - 1]];
                                // qOffset+qInset > (Len*2)/3 -> skipped, and does not ratchet (:536-538, 576-578)
                                const int qOffset = m_seed_offset(qSeg, a0, k);
                                const int qInset = m_seed_offset_from_end(qSeg, qN, aL, k);
                                if (qOffset + qInset > (qLen * 2) / 3) continue;
                                const uint32_t ri = atomicAdd(&cursor[0], 1u);
                                const uint32_t off = atomicAdd(&cursor[1], (uint32_t)len);
                                if (ri < rec_cap && off + (uint32_t)len <= int_cap) {
                                    MapRec rec = {w, t, off, (uint32_t)len, seq};
                                    recs[ri] = rec;
                                    for (int x = 0; x < len; x++) {
                                        ma[off + x] = L.qIdx[ca[x]];
                                        mb[off + x] = L.tIdx[cb[x]];
                                    }
                                } else {
                                    cursor[3] = 1;
                                }
                                seq++;
                                algb += 8ull * (unsigned long long)len;
                                const int limit = (len * 4) / 5;
                                if (limit > thrS) thrS = limit;
                                if (s == 0 && limit > thrOther) thrOther = limit;  // forward also raises minRCMatches (:547-549)
                            }
                        }
                        if (err) atomicOr(&cursor[2], err);
                    }
                    thr[s] = __shfl(thrS, 0, 64);
                    thr[1 - s] = __shfl(thrOther, 0, 64);
                    seq = (uint32_t)__shfl((int)seq, 0, 64);
                    MP_TICK(4)  // flank test, records and chains out (lane 0)
#ifdef DP_PROF_BUILD
                    mp[6]++;
#endif
                }
            }
        }
        if (thr_io && lane == 0) {
            thr_io[2 * pair] = thr[0];
            thr_io[2 * pair + 1] = thr[1];
        }
    }
    if (lane == 0 && algb) atomicAdd((unsigned long long*)(cursor + 8), algb);  // (the chain links were counted on lane 0 only)
#ifdef DP_PROF_BUILD
    if (lane == 0 && prof) {
        MP_TICK(5)  // whatever is left: thresholds, skipped windows
        for (int i = 0; i < 7; i++) atomicAdd(&prof[i], mp[i]);
        atomicAdd(&prof[7], 1ull);                    // waves
        atomicMax(&prof[8], wall_clock64() - mstart);  // the launch's slowest wave
        atomicAdd(&prof[9], wall_clock64() - mstart);
    }
#endif
#undef MP_TICK
}

// ---------------------------------------------------------------------------------------------------------------
// A18, the parallel part of AddSingleSeeds (seeds/seeds.go:160-200).  The reference walks the mapping reference in windows of
// seed_rate bases: a window in which no k-mer is a seed yet gets its best-valued k-mer as a new seed.  "Is a seed yet" makes the
// walk sequential - but only k-mers that are the BEST of some window can ever be seeds, so everything else about a window can be
// computed for all windows at once: its best k-mer, and the k-mers of its count region that are the best of ANY window (its
// candidates: five or six of ~seed_rate).  The host then walks the windows in order and probes only the candidates against the
// seeds added so far (host_map.cpp): 3.9 s of one host thread per 375 Mb of reference -> a few kernels + 15 M probes.
// The count region is CountKmersBetween's (sequence.go:332-337 + asm:81-203: whole bytes, the parent's skipBack, the do-while group
// loop - the arithmetic of the host mirror and the oracle), bases beyond the sequence read as zero.
struct SsGeom {
    long long len, seed_rate;
    int k, skip_back;
    uint32_t n_windows;
};
__device__ __forceinline__ uint32_t ss_code(const uint8_t* __restrict__ p, long long pos, long long len) {
    return pos < len ? (uint32_t)((p[pos >> 2] >> (6 - 2 * (int)(pos & 3))) & 3u) : 0u;
}
__device__ __forceinline__ void ss_region(const SsGeom& G, long long i, long long* p0, long long* P) {
    const long long startB = (i + 3) / 4, endB = (i + G.seed_rate) / 4;
    const long long nb = endB - startB;
    const long long nk = 4 * (nb - 1) - G.skip_back - G.k + 1;
    long long groups = (nk & ~(long long)3) / 4;
    if (groups < 1) groups = 1;
    *p0 = startB * 4;
    *P = 4 + 4 * groups + (nk & 3);
}
// best k-mer of every window (first maximum of the value table over the k-mers that start at i .. i + seed_rate - k) + its bit in `bits`
__global__ void ss_best_kernel(const uint8_t* __restrict__ packed, SsGeom G, const double* __restrict__ values, uint32_t* __restrict__ best,
                               uint32_t* __restrict__ bits) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= G.n_windows) return;
    const long long i = (long long)w * G.seed_rate, end = i + G.seed_rate;
    const uint32_t mask = (uint32_t)(((unsigned long long)1 << (2 * G.k)) - 1);
    uint32_t km = 0;
    for (int j = 0; j < G.k; j++) km = (km << 2) | ss_code(packed, i + j, G.len);
    double bv = values[km];
    uint32_t bk = km;
    for (long long j = i + G.k; j < end; j++) {
        km = ((km << 2) | ss_code(packed, j, G.len)) & mask;
        const double v = values[km];
        if (v > bv) {
            bv = v;
            bk = km;
        }
    }
    best[w] = bk;
    atomicOr(&bits[bk >> 5], 1u << (bk & 31));
}
// pass 0: how many k-mers of the window's count region are somebody's best; pass 1: write them at off[w]
template <bool WRITE>
__global__ void ss_cand_kernel(const uint8_t* __restrict__ packed, SsGeom G, const uint32_t* __restrict__ bits, uint32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ off, uint32_t* __restrict__ cand) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= G.n_windows) return;
    long long p0, P;
    ss_region(G, (long long)w * G.seed_rate, &p0, &P);
    const uint32_t mask = (uint32_t)(((unsigned long long)1 << (2 * G.k)) - 1);
    uint32_t km = 0, n = 0;
    const uint32_t at = WRITE ? off[w] : 0u;
    for (int j = 0; j < G.k; j++) km = (km << 2) | ss_code(packed, p0 + j, G.len);
    for (long long j = 0; j < P; j++) {
        if (j) km = ((km << 2) | ss_code(packed, p0 + j + G.k - 1, G.len)) & mask;
        if ((bits[km >> 5] >> (km & 31)) & 1u) {
            if (WRITE) cand[at + n] = km;
            n++;
        }
    }
    if (!WRITE) cnt[w] = n;
}

extern "C" int dp_single_seed_candidates(dp_ctx* ctx, uint32_t read, int k, int64_t seed_rate, dp_single_seed_batch* out) {
    if (!ctx || !out || k < 4 || k > 15 || seed_rate < 1) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_single_seed_candidates: bad arguments") : DP_ERR_ARG;
    if (read >= ctx->n_reads) return dp_fail(ctx, DP_ERR_ARG, "dp_single_seed_candidates: read index out of range");
    const dp_ctx* src = ctx->owner ? ctx->owner : ctx;
    const size_t nk = (size_t)1 << (2 * k);
    if (!src->d_values.p || src->n_values != nk) return dp_fail(ctx, DP_ERR_STATE, "dp_single_seed_candidates: no value table of this k resident");
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    SsGeom G;
    G.len = (long long)src->h_len[read];
    G.seed_rate = seed_rate;
    G.k = k;
    G.skip_back = 4 - (int)(G.len % 4);  // top-level sequence: finalLen = len % 4 (0 when len % 4 == 0: sequence.go:70,88)
    // windows: for (i = 0; i < len - seed_rate; i += seed_rate)
    const long long span = G.len - seed_rate;
    G.n_windows = span > 0 ? (uint32_t)((span + seed_rate - 1) / seed_rate) : 0u;
    out->n_windows = G.n_windows;
    if (!G.n_windows) return DP_OK;
    const uint8_t* packed = (const uint8_t*)src->d_packed.p + src->h_boff[read];
    void *d_bits = nullptr, *d_best = nullptr, *d_cnt = nullptr, *d_off = nullptr, *d_cand = nullptr, *d_tmp = nullptr;
    auto cleanup = [&] {
        for (void* p : {d_bits, d_best, d_cnt, d_off, d_cand, d_tmp})
            if (p) dp_dev_free(p);
    };
#define DSS(x)                                                                        \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            cleanup();                                                                \
            return dp_fail(ctx, DP_ERR_HIP, "dp_single_seed_candidates: " #x, e_);    \
        }                                                                             \
    } while (0)
    const uint32_t nw = G.n_windows, blocks = (nw + 255) / 256;
    DSS(dp_dev_malloc(&d_bits, nk / 8 + 64));
    DSS(dp_dev_malloc(&d_best, (size_t)nw * 4));
    DSS(dp_dev_malloc(&d_cnt, ((size_t)nw + 1) * 4));
    DSS(dp_dev_malloc(&d_off, ((size_t)nw + 1) * 4));
    DSS(hipMemsetAsync(d_bits, 0, nk / 8 + 64, ctx->stream));
    DSS(hipMemsetAsync(d_cnt, 0, ((size_t)nw + 1) * 4, ctx->stream));
    hipLaunchKernelGGL(ss_best_kernel, dim3(blocks), dim3(256), 0, ctx->stream, packed, G, (const double*)src->d_values.p, (uint32_t*)d_best,
                       (uint32_t*)d_bits);
    hipLaunchKernelGGL(ss_cand_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, packed, G, (const uint32_t*)d_bits, (uint32_t*)d_cnt,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr);
    DSS(hipGetLastError());
    size_t tb = 0;
    DSS(rocprim::exclusive_scan(nullptr, tb, (const uint32_t*)d_cnt, (uint32_t*)d_off, 0u, (size_t)nw + 1, rocprim::plus<uint32_t>(), ctx->stream));
    DSS(dp_dev_malloc(&d_tmp, tb + 64));
    DSS(rocprim::exclusive_scan(d_tmp, tb, (const uint32_t*)d_cnt, (uint32_t*)d_off, 0u, (size_t)nw + 1, rocprim::plus<uint32_t>(), ctx->stream));
    uint32_t total = 0;
    DSS(hipMemcpyAsync(&total, (const uint32_t*)d_off + nw, 4, hipMemcpyDeviceToHost, ctx->stream));
    DSS(dp_stream_sync(ctx));
    DSS(dp_dev_malloc(&d_cand, (size_t)total * 4 + 64));
    hipLaunchKernelGGL(ss_cand_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, packed, G, (const uint32_t*)d_bits, (uint32_t*)nullptr,
                       (const uint32_t*)d_off, (uint32_t*)d_cand);
    DSS(hipGetLastError());
    // results to the host: [best nw | off nw + 1 | cand total] in ordinary host memory of the context - the caller reads them in a
    // sequential walk on the CPU, and reading the library's pinned blocks from the CPU costs ~50 ns per access (45 ms for config 3's
    // 115 k windows, ten times the walk itself)
    ctx->ss_host.resize((size_t)2 * nw + 1 + total + 16);
    uint32_t* h = ctx->ss_host.data();
    DSS(hipMemcpyAsync(h, d_best, (size_t)nw * 4, hipMemcpyDeviceToHost, ctx->stream));
    DSS(hipMemcpyAsync(h + nw, d_off, ((size_t)nw + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (total) DSS(hipMemcpyAsync(h + 2 * (size_t)nw + 1, d_cand, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    DSS(dp_stream_sync(ctx));
#undef DSS
    cleanup();
    out->best = h;
    out->cand_off = h + nw;
    out->cand = h + 2 * (size_t)nw + 1;
    return DP_OK;
}

int dp_map_windows_impl(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len, uint32_t nw, int k,
                        dp_chain_batch* out, int phase, int32_t* thr_io) {
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_map_windows before dp_round_begin");
    if (nw & 1) return dp_fail(ctx, DP_ERR_ARG, "dp_map_windows: windows come in (forward, reverse-complement) pairs");
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    if (pin_reserve(ctx, ctx->h_moff, 16)) return DP_ERR_HIP;
    ((uint64_t*)ctx->h_moff.p)[0] = 0;
    out->off = (const uint64_t*)ctx->h_moff.p;
    const uint32_t W = ctx->W, SW = ctx->SW, M = ctx->n_seqs;
    if (nw == 0 || M == 0) return DP_OK;
    uint32_t* d_qmeta = nullptr;
    u64* d_words = nullptr;
    int32_t* d_mc = nullptr;
    uint32_t mc_n = 0;
    uint32_t* d_qcnt_unused = nullptr;
    int rc = DP_OK;
    if (phase == 1 && ctx->map_stage_windows == nw && ctx->map_stage_valid) {
        // the reverse pass of a shard follows its forward pass on the same windows: the query stage's results are still there
        d_qmeta = (uint32_t*)ctx->d_qmeta.p;
    } else {
        rc = dp_query_stage(ctx, w_segs, w_off, nw, 0.25, &d_qmeta, &d_words, &d_mc, &mc_n, &d_qcnt_unused);  // Matches(.., 0.25) :502-503
        if (rc != 0) return rc;
        ctx->map_stage_windows = nw;
        ctx->map_stage_valid = phase == 0;
    }
    int32_t* d_thr = nullptr;
    if (thr_io) {
        if (dev_reserve(ctx, ctx->d_sa, (size_t)nw * 4 + 64)) return DP_ERR_HIP;
        d_thr = (int32_t*)ctx->d_sa.p;
        DP_HIP(hipMemcpyAsync(d_thr, thr_io, (size_t)nw * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (dev_reserve(ctx, ctx->d_cursor, 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_sched, (size_t)nw * 4 + 16)) return DP_ERR_HIP;
    DP_HIP(hipMemcpyAsync(ctx->d_sched.p, w_len, (size_t)nw * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t n_pairs = nw / 2;
    const uint32_t blocks = std::min<uint32_t>(256, (n_pairs + M_WAVES - 1) / M_WAVES);
    const size_t waves = (size_t)blocks * M_WAVES;
    size_t poolElems = waves * M_CHAINS * M_QMAX;
    if (dev_reserve(ctx, ctx->d_pool, poolElems * 2 * 2 + waves * M_CHAINS * 2 + 64)) return DP_ERR_HIP;
    uint16_t* poolA = (uint16_t*)ctx->d_pool.p;
    uint16_t* poolB = poolA + poolElems;
    uint16_t* poolLen = poolB + poolElems;
    // (round 6) the BIG variant: run again over the batch when the ordinary kernel reports that a reduced window or chunk did not fit
    bool big = false;
    uint32_t big_blocks = blocks, big_q = 0, big_t = 0;
    uint32_t rec_cap = std::max<uint32_t>(1u << 16, (uint32_t)(ctx->d_mrec.cap / sizeof(MapRec)));
    uint32_t int_cap = std::max<uint32_t>(1u << 21, (uint32_t)(ctx->d_ma.cap / 4));
    uint32_t cur[16];
    float total_ms = 0;
    // profiling build + DP_MAP_PROF=1: per-phase sums of the launch's waves (map_kernel's MP_TICK), printed per call
    unsigned long long* d_mprof = nullptr;
#ifdef DP_PROF_BUILD
    static const bool map_prof = dp_debug("map_prof");
    if (map_prof) {
        if (dev_reserve(ctx, ctx->d_sb, 16 * 8 + 64)) return DP_ERR_HIP;
        d_mprof = (unsigned long long*)ctx->d_sb.p;
    }
#endif
    const int one_lane = dp_tune("map_one_lane", 0) ? 1 : 0;  // (tests: dynamicMatch on one lane)
    for (;;) {
        if (dev_reserve(ctx, ctx->d_mrec, (size_t)rec_cap * sizeof(MapRec))) return DP_ERR_HIP;
        if (dev_reserve(ctx, ctx->d_ma, (size_t)int_cap * 4)) return DP_ERR_HIP;
        if (dev_reserve(ctx, ctx->d_mb, (size_t)int_cap * 4)) return DP_ERR_HIP;
        DP_HIP(hipMemsetAsync(ctx->d_cursor.p, 0, 64, ctx->stream));
        if (d_mprof) DP_HIP(hipMemsetAsync(d_mprof, 0, 16 * 8, ctx->stream));
        DP_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
        if (!big)
            hipLaunchKernelGGL(map_kernel<false>, dim3(blocks), dim3(64 * M_WAVES), 0, ctx->stream, ctx->qsegs_dev,
                               ctx->qoff_dev, (const uint32_t*)ctx->d_sched.p, n_pairs, (const u64*)ctx->d_qsets.p,
                               (const uint32_t*)d_qmeta, (const u64*)ctx->d_cand.p, (const dp_seq_ref*)ctx->d_seqrefs.p,
                               (const int32_t*)ctx->d_segs.p, (const u64*)ctx->d_seedsets.p, W, SW, k, poolA, poolB, poolLen,
                               (MapRec*)ctx->d_mrec.p, rec_cap, (int32_t*)ctx->d_ma.p, (int32_t*)ctx->d_mb.p, int_cap,
                               (uint32_t*)ctx->d_cursor.p, phase, d_thr, one_lane, (const u64*)d_words, d_mprof, (uint32_t*)nullptr, 0u, 0u);
        else
            hipLaunchKernelGGL(map_kernel<true>, dim3(big_blocks), dim3(64 * M_WAVES), 0, ctx->stream, ctx->qsegs_dev,
                               ctx->qoff_dev, (const uint32_t*)ctx->d_sched.p, n_pairs, (const u64*)ctx->d_qsets.p,
                               (const uint32_t*)d_qmeta, (const u64*)ctx->d_cand.p, (const dp_seq_ref*)ctx->d_seqrefs.p,
                               (const int32_t*)ctx->d_segs.p, (const u64*)ctx->d_seedsets.p, W, SW, k, poolA, poolB, poolLen,
                               (MapRec*)ctx->d_mrec.p, rec_cap, (int32_t*)ctx->d_ma.p, (int32_t*)ctx->d_mb.p, int_cap,
                               (uint32_t*)ctx->d_cursor.p, phase, d_thr, one_lane, (const u64*)d_words, d_mprof, (uint32_t*)ctx->d_qbig.p,
                               big_q, big_t);
        DP_HIP(hipGetLastError());
        DP_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
        DP_HIP(hipMemcpyAsync(cur, ctx->d_cursor.p, 64, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
        float ms = 0;
        hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]);
        total_ms += ms;
        if (cur[3] || cur[0] > rec_cap || cur[1] > int_cap) {
            rec_cap = std::max(rec_cap * 2, cur[0] + 1024);
            int_cap = std::max(int_cap * 2, cur[1] + 1024);
            continue;
        }
        if ((cur[2] & 1u) && !big) {
            // a reduced window or chunk beyond the LDS arrays: the whole batch again with the working set in global memory, sized from the
            // batch's longest window and the index's longest chunk (16-bit indices: 65 535 seeds)
            uint64_t mq = 1;
            for (uint32_t w = 0; w < nw; w++) mq = std::max<uint64_t>(mq, (w_off[w + 1] - w_off[w]) / 2);
            big_q = (uint32_t)std::min<uint64_t>(65535, mq);
            big_t = std::min<uint32_t>(65535u, std::max<uint32_t>(1u, ctx->max_seq_seeds));
            if (big_q <= M_QMAX && big_t <= M_TMAX) break;  // (cannot be: the error stands)
            big_blocks = std::min<uint32_t>(blocks, 64);
            const size_t bw = (size_t)big_blocks * M_WAVES;
            if (dev_reserve(ctx, ctx->d_qbig, bw * m_big_words(big_q, big_t) * 4 + 64)) return DP_ERR_HIP;
            poolElems = bw * M_CHAINS * big_q;
            if (dev_reserve(ctx, ctx->d_pool, poolElems * 2 * 2 + bw * M_CHAINS * 2 + 64)) return DP_ERR_HIP;
            poolA = (uint16_t*)ctx->d_pool.p;
            poolB = poolA + poolElems;
            poolLen = poolB + poolElems;
            big = true;
            continue;
        }
        break;
    }
    float qms = 0;
    hipEventElapsedTime(&qms, ctx->ev[4], ctx->ev[5]);
    out->kernel_ms = (double)total_ms + (double)qms;
    if (d_mprof) {
        unsigned long long h[16];
        if (hipMemcpy(h, d_mprof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[7]) {
            const double w = (double)h[7];
            fprintf(stderr, "[map prof] %u window pairs on %llu waves, kernel %.1f us, slowest wave %.1f us, mean wave %.1f us | us per wave: prefilter %.1f reduce target %.1f "
                            "reduce query %.1f dynamicMatch %.1f out %.1f rest %.1f | candidates chained per wave %.1f\n",
                    n_pairs, h[7], 1e3 * total_ms, h[8] / 100.0, h[9] / 100.0 / w, h[0] / 100.0 / w, h[1] / 100.0 / w, h[2] / 100.0 / w, h[3] / 100.0 / w,
                    h[4] / 100.0 / w, h[5] / 100.0 / w, (double)h[6] / w);
        }
    }
    {
        unsigned long long ab = 0;
        memcpy(&ab, &cur[8], 8);
        out->alg_bytes = (double)ab;
    }
    if (thr_io) DP_HIP(hipMemcpy(thr_io, d_thr, (size_t)nw * 4, hipMemcpyDeviceToHost));
    if (cur[2]) {
        char msg[160];
        snprintf(msg, sizeof msg, "map chaining exceeded a device capacity (bits %u: 1 reduced sequence, 2 chain pool, 4 good-chain list)", cur[2]);
        return dp_fail(ctx, DP_ERR_CAPACITY, msg);
    }
    std::vector<uint32_t> qm((size_t)nw * 4);
    DP_HIP(hipMemcpy(qm.data(), d_qmeta, (size_t)nw * 16, hipMemcpyDeviceToHost));
    for (uint32_t w = 0; w < nw; w++)
        if (qm[4 * w + 2] & 1) return dp_fail(ctx, DP_ERR_CAPACITY, "window with more than 65535 usable seeds");
    const uint32_t nm = cur[0], ni = cur[1];
    if (pin_reserve(ctx, ctx->h_mrec, (size_t)nm * sizeof(MapRec) + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_ma, (size_t)ni * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mb, (size_t)ni * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mq, (size_t)nm * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mt, (size_t)nm * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_moff, ((size_t)nm + 1) * 8)) return DP_ERR_HIP;
    std::vector<int32_t> ta(ni), tb(ni);
    if (nm) {
        DP_HIP(hipMemcpyAsync(ctx->h_mrec.p, ctx->d_mrec.p, (size_t)nm * sizeof(MapRec), hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(hipMemcpyAsync(ta.data(), ctx->d_ma.p, (size_t)ni * 4, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(hipMemcpyAsync(tb.data(), ctx->d_mb.p, (size_t)ni * 4, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
    }
    MapRec* recs = (MapRec*)ctx->h_mrec.p;
    std::vector<uint32_t> order(nm);
    for (uint32_t i = 0; i < nm; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (recs[a].window != recs[b].window) return recs[a].window < recs[b].window;
        return recs[a].seq < recs[b].seq;
    });
    uint32_t* mw = (uint32_t*)ctx->h_mq.p;
    uint32_t* mt = (uint32_t*)ctx->h_mt.p;
    uint64_t* moff = (uint64_t*)ctx->h_moff.p;
    int32_t* fa = (int32_t*)ctx->h_ma.p;
    int32_t* fb = (int32_t*)ctx->h_mb.p;
    uint64_t pos = 0;
    for (uint32_t i = 0; i < nm; i++) {
        const MapRec& r = recs[order[i]];
        mw[i] = r.window;
        mt[i] = r.target;
        moff[i] = pos;
        memcpy(fa + pos, ta.data() + r.off, (size_t)r.len * 4);
        memcpy(fb + pos, tb.data() + r.off, (size_t)r.len * 4);
        pos += r.len;
    }
    moff[nm] = pos;
    out->n_chains = nm;
    out->window = mw;
    out->target = mt;
    out->off = moff;
    out->match_a = fa;
    out->match_b = fb;
    return DP_OK;
}
