// libdownpore_hip.so — A16 (part): the seed-space multiple alignment at the heart of multiAligner.Consensus
// (seeds/alignment.go:23-268) for many query windows at once.  One wave per group of sequences; lane i owns sequence i
// (its cursor pos/offs/gaps, its support counters and its match list), the consensus grows one seed per iteration.
// The host keeps what surrounds it (trimming the matched targets, Reduced(), trimToBestSeed, PAF): see host_overlap.cpp.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dp_common.h"
#include <string>

#include "dp_launch.h"

#define CA_WAVES 4
#define CA_CAP 6144  // ints of one group staged in LDS

#define CA_RL(v_, l_) __builtin_amdgcn_readlane((v_), (l_))
#define RFLc(v_) __builtin_amdgcn_readfirstlane(v_)
typedef uint64_t u64;

// gapRange seeds/alignment.go:411-424
__device__ __forceinline__ void ca_gap_range(int gap, int k, int& mn, int& mx) {
    const int m0 = (gap * 2) / 3 - k, x0 = (gap * 3) / 2 + k + 1;
    const bool neg = m0 < 0, small = !neg && x0 < 20;
    mx = neg ? (x0 < 0 ? 0 : x0) : (small ? 20 : x0);
    mn = neg ? -k : (small ? 0 : m0);
}

// flags per group: 1 = not computed (more than 64 sequences, too large for LDS, or a value left the 32-bit safe range)
__global__ __launch_bounds__(64 * CA_WAVES) void consensus_align_kernel(const int32_t* __restrict__ segs, const uint64_t* __restrict__ seq_off,
                                                                        const uint32_t* __restrict__ group_off, uint32_t n_groups, int k,
                                                                        int32_t* __restrict__ cons, const uint64_t* __restrict__ cons_off,
                                                                        uint32_t* __restrict__ cons_len, int32_t* __restrict__ match_a,
                                                                        int32_t* __restrict__ match_b, uint32_t* __restrict__ match_len,
                                                                        uint32_t* __restrict__ flags) {
    __shared__ int32_t stage[CA_WAVES][CA_CAP];
    int32_t* S = stage[threadIdx.x >> 6];
    const int lane = dp_lane();
    const uint32_t g = blockIdx.x * CA_WAVES + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const uint32_t s0 = group_off[g], s1 = group_off[g + 1];
    const int ns = __builtin_amdgcn_readfirstlane((int)(s1 - s0));
    const uint64_t base0 = seq_off[s0];
    const uint64_t total = seq_off[s1] - base0;
    if (lane == 0) {
        flags[g] = 0;
        cons_len[g] = 0;
    }
    if (ns > 64 || total > CA_CAP) {
        if (lane == 0) flags[g] = 1;
        return;
    }
    for (uint64_t i = lane; i < total; i += 64) S[i] = segs[base0 + i];
    const bool mine = lane < ns;
    const int b = mine ? (int)(seq_off[s0 + lane] - base0) : 0;                      // my sequence starts at S[b]
    const int sl = mine ? (int)(seq_off[s0 + lane + 1] - seq_off[s0 + lane]) : 0;  // ints; 0 = no reduced sequence
    const uint64_t out_base = mine ? seq_off[s0 + lane] : 0;                         // my match list starts here
    int pos = -1, offs = 0, gaps = 50, supported = 0, dist = 0, mlen = 0;
    int clen = 0;
    bool bad = false;  // a quantity left the range in which 32-bit arithmetic equals the reference's 64-bit ints
    const int kLim = 1 << 28;
    int32_t* my_cons = cons + cons_off[g];
    for (;;) {
        int near = 100000;
        // ---- state of every sequence's next seed (constant during the support scan)
        const int p2s = pos + 1;
        const bool okS = sl > 0 && p2s < sl / 2;
        const int od = okS ? S[b + p2s * 2] - offs : 0;
        supported = 0;
        const bool fin = !mine || sl == 0 || pos >= (sl - 1) / 2 - 1;
        int fCount = __popcll(__ballot(fin && mine));
        int d = 0, nextSeed = 0, minD = 0, maxD = 0;
        if (!fin) {
            d = S[b + pos * 2 + 2] - offs;
            dist = d;
            nextSeed = S[b + pos * 2 + 3];
            ca_gap_range(d + gaps, k, minD, maxD);
            minD -= gaps;
            maxD -= gaps;
        }
        if (od >= kLim || od <= -kLim || gaps >= kLim || d >= kLim || d <= -kLim || dist >= kLim || dist <= -kLim) bad = true;
        if (__ballot(bad)) break;
        // ---- support of every proposer, in sequence order (`near` shrinks as proposers are seen, :70-76)
        unsigned long long cand = __ballot(!fin);
        bool memoOk = false;
        int memoD = 0, memoSeed = 0, memoMin = 0, memoMax = 0, memoCnt = 0, memoSum = 0;
        bool fnd = false;
        int val = 0;
        while (cand) {
            const int i = __builtin_ctzll(cand);
            cand &= cand - 1;
            const int di = CA_RL(d, i);
            if (!(di < near && di > -k)) continue;
            const int seedI = CA_RL(nextSeed, i), minI = CA_RL(minD, i), maxI = CA_RL(maxD, i);
            if (near > maxI) near = maxI;
            if (!(memoOk && memoD == di && memoSeed == seedI && memoMin == minI && memoMax == maxI)) {
                // every OTHER sequence looks for seedI inside its distance window (:101-131); the proposer's own lane
                // takes part too and is subtracted below (the result depends only on (d, seed, window) and the lane)
                fnd = false;
                val = 0;
                if (okS) {
                    int min2, max2;
                    ca_gap_range(di + gaps, k, min2, max2);
                    if (min2 > minI) min2 = minI;
                    if (max2 < maxI) max2 = maxI;
                    int p2 = p2s, otherD = od;
                    while (otherD < min2 && p2 < sl / 2) {
                        p2++;
                        otherD += S[b + p2 * 2] + k;
                    }
                    while (otherD < max2 && p2 < sl / 2) {
                        if (S[b + p2 * 2 + 1] == seedI) {
                            fnd = true;
                            val = otherD;
                            break;
                        }
                        p2++;
                        otherD += S[b + p2 * 2] + k;
                    }
                }
                memoCnt = __popcll(__ballot(fnd));
                memoSum = wave_sum(fnd ? val : 0);
                memoOk = true;
                memoD = di;
                memoSeed = seedI;
                memoMin = minI;
                memoMax = maxI;
            }
            if (lane == i) {
                supported = 1 + memoCnt - (fnd ? 1 : 0);
                dist += memoSum - (fnd ? val : 0);
            }
        }
        if (fCount >= ns) break;
        // ---- the seed to append: first proposer with support > 1, replaced by a later one with the same seed and more
        //      support or with another seed at a smaller mean distance (:141-160)
        int minseed = -1, mindist = 0, minsup = 0, selMin = 0, selMax = 0;
        {
            unsigned long long sup = __ballot(supported > 1);
            while (sup) {
                const int i = __builtin_ctzll(sup);
                sup &= sup - 1;
                const int si = CA_RL(supported, i);
                const int dv = CA_RL(dist, i) / si;
                const int seed = CA_RL(nextSeed, i);
                if (minseed == -1 || (minseed == seed && si > minsup) || (minseed != seed && mindist > dv)) {
                    minsup = si;
                    mindist = dv;
                    minseed = seed;
                    const int gi = CA_RL(gaps, i);
                    ca_gap_range(dv + gi, k, selMin, selMax);
                    selMin -= gi;
                    selMax -= gi;
                }
            }
        }
        if (minseed == -1) {  // nobody is supported: advance the nearest sequence by one seed (:162-187)
            const int dvv = supported > 1 ? dist / supported : dist;
            const bool can = mine && sl > 0 && pos < ns / 2;  // len(segments)/2 is len(seqs)/2 (:170)
            int best = can ? dvv : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
            if (best >= 100000) break;
            const int minIndex = __builtin_ctzll(__ballot(can && dvv == best));  // first of the minimal ones
            if (mine && sl > 0) {
                gaps += best;
                offs += best;
            }
            if (lane == minIndex) {
                gaps = 0;
                offs = 0;
                pos++;
            }
            continue;
        }
        if (lane == 0) {
            my_cons[clen] = mindist;
            my_cons[clen + 1] = minseed;
        }
        clen += 2;
        // ---- every sequence tries to follow (:197-247)
        bool finC = true;
        if (mine && sl > 0) {
            int matchDex = pos + 1;
            if (matchDex < sl / 2) {
                int min2, max2;
                ca_gap_range(mindist + gaps, k, min2, max2);
                if (min2 > selMin) min2 = selMin;
                if (max2 < selMax) max2 = selMax;
                int otherD = S[b + matchDex * 2] - offs;
                while (otherD < min2 && matchDex < sl / 2) {
                    matchDex++;
                    otherD += S[b + matchDex * 2] + k;
                }
                bool found = false;
                while (otherD < max2 && matchDex < sl / 2) {
                    if (S[b + matchDex * 2 + 1] == minseed) {
                        pos = matchDex;
                        offs = 0;
                        gaps = 0;
                        match_a[out_base + mlen] = clen / 2 - 1;
                        match_b[out_base + mlen] = matchDex;  // index into the REDUCED sequence; the host maps it back
                        mlen++;
                        found = true;
                        break;
                    }
                    matchDex++;
                    otherD += S[b + matchDex * 2] + k;
                }
                finC = false;
                if (!found) {
                    gaps += mindist;
                    offs += mindist;
                    int p = pos;
                    while (p < sl / 2 && offs > S[b + p * 2 + 2] + 50) {
                        offs -= S[b + p * 2 + 2] + k;
                        p++;
                        pos++;
                    }
                    if (p >= sl / 2) finC = true;
                }
            }
        }
        if (__popcll(__ballot(finC && mine)) >= ns) break;
    }
    if (__ballot(bad)) {
        if (lane == 0) flags[g] = 1;
        return;
    }
    if (lane == 0) {
        my_cons[clen] = 0;
        cons_len[g] = (uint32_t)clen + 1;
    }
    if (mine) match_len[s0 + lane] = (uint32_t)mlen;
}

extern "C" int dp_consensus_align(dp_ctx* ctx, const int32_t* segs, const uint64_t* seq_off, const uint32_t* group_off,
                                  uint32_t n_groups, int k, dp_consensus_batch* out) {
    if (!ctx || !out || (n_groups && (!segs || !seq_off || !group_off))) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_consensus_align: bad arguments") : DP_ERR_ARG;
    memset(out, 0, sizeof(*out));
    out->n_groups = n_groups;
    if (n_groups == 0) return DP_OK;
    hipSetDevice(ctx->device);
    const uint32_t n_seqs = group_off[n_groups];
    const uint64_t n_ints = seq_off[n_seqs];
    // consensus of group g gets (ints of the group + 2) slots
    const size_t b_segs = n_ints * 4, b_soff = ((size_t)n_seqs + 1) * 8, b_goff = ((size_t)n_groups + 1) * 4, b_coff = ((size_t)n_groups + 1) * 8;
    const size_t in_bytes = b_soff + b_coff + b_segs + b_goff + 64;
    if (pin_reserve(ctx, ctx->h_cin, in_bytes)) return DP_ERR_HIP;
    uint8_t* hin = (uint8_t*)ctx->h_cin.p;
    uint64_t* h_soff = (uint64_t*)hin;
    uint64_t* h_coff = (uint64_t*)(hin + b_soff);
    int32_t* h_segs = (int32_t*)(hin + b_soff + b_coff);
    uint32_t* h_goff = (uint32_t*)(hin + b_soff + b_coff + b_segs);
    memcpy(h_soff, seq_off, b_soff);
    memcpy(h_segs, segs, b_segs);
    memcpy(h_goff, group_off, b_goff);
    uint64_t cpos = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        h_coff[g] = cpos;
        cpos += (seq_off[group_off[g + 1]] - seq_off[group_off[g]]) + 2;
    }
    h_coff[n_groups] = cpos;
    const size_t b_cons = cpos * 4, b_clen = (size_t)n_groups * 4, b_m = n_ints * 4, b_mlen = (size_t)n_seqs * 4, b_flag = (size_t)n_groups * 4;
    const size_t out_bytes = b_cons + b_clen + 2 * b_m + b_mlen + b_flag + 64;
    if (dev_reserve(ctx, ctx->d_cin, in_bytes)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_cout, out_bytes)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_cout, out_bytes + b_coff)) return DP_ERR_HIP;
    uint8_t* din = (uint8_t*)ctx->d_cin.p;
    uint8_t* dout = (uint8_t*)ctx->d_cout.p;
    DP_HIP(hipMemcpyAsync(din, hin, in_bytes - 64, hipMemcpyHostToDevice, ctx->stream));
    int32_t* d_cons = (int32_t*)dout;
    uint32_t* d_clen = (uint32_t*)(dout + b_cons);
    int32_t* d_ma = (int32_t*)(dout + b_cons + b_clen);
    int32_t* d_mb = (int32_t*)(dout + b_cons + b_clen + b_m);
    uint32_t* d_mlen = (uint32_t*)(dout + b_cons + b_clen + 2 * b_m);
    uint32_t* d_flag = (uint32_t*)(dout + b_cons + b_clen + 2 * b_m + b_mlen);
    DP_HIP(dp_mark(ctx, 0));
    hipLaunchKernelGGL(consensus_align_kernel, dim3((n_groups + CA_WAVES - 1) / CA_WAVES), dim3(64 * CA_WAVES), 0, ctx->stream,
                       (const int32_t*)(din + b_soff + b_coff), (const uint64_t*)din, (const uint32_t*)(din + b_soff + b_coff + b_segs),
                       n_groups, k, d_cons, (const uint64_t*)(din + b_soff), d_clen, d_ma, d_mb, d_mlen, d_flag);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 1));
    uint8_t* hout = (uint8_t*)ctx->h_cout.p;
    DP_HIP(hipMemcpyAsync(hout, dout, out_bytes - 64, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    memcpy(hout + out_bytes, h_coff, b_coff);
    float ms = 0;
    ms = dp_elapsed(ctx, 0, 1);
    out->kernel_ms = ms;
    out->cons = (const int32_t*)hout;
    out->cons_len = (const uint32_t*)(hout + b_cons);
    out->match_a = (const int32_t*)(hout + b_cons + b_clen);
    out->match_b = (const int32_t*)(hout + b_cons + b_clen + b_m);
    out->match_len = (const uint32_t*)(hout + b_cons + b_clen + 2 * b_m);
    out->flags = (const uint32_t*)(hout + b_cons + b_clen + 2 * b_m + b_mlen);
    out->cons_off = (const uint64_t*)(hout + out_bytes);
    return DP_OK;
}

// =====================================================================================================================
// A16 + A17 on the device: BuildConsensus (overlap/combine.go:163-193), multiAligner.Consensus (seeds/alignment.go:23-268),
// NewSeedContig / trimToBestSeed (combine.go:21-133) and the numbers of finalCheckWorker's PAF lines
// (commands/overlap.go:197-233) for every query window of a round, straight from the device-resident output of the
// chaining stage.  One wave per query window ("group": forward query 2g, reverse-complement query 2g+1).
//   1. matches of the group in arrival order (forward query's, then the rc query's; targets ascending);
//   2. per match: un-reverse-complement (rc query), GetBasesCovered filter (< 25 bases on either side drops it), Trimmed()
//      of the target to the query-aligned span -> the trimmed sequences of the group, in LDS;
//   3. seeds shared by >= 2 of them (GetSharedIDs(.., 2, true)), Reduced() of every sequence;
//   4. the seed-space multiple alignment (same loop as consensus_align_kernel above: lane i owns sequence i);
//   5. parts with < 3 matched seeds removed (swap with last, :258-266), trimToBestSeed, contig fields, PAF numbers,
//      SetIgnore decisions.
// What does not fit the LDS layout (more than 64 sequences, more than CF_T ints, values outside the 32-bit safe range)
// is flagged and left to the host path, which then fetches the matches.
// Two layouts of the wave's LDS block.  The small one (int16 elements, a quarter of the capacities' bytes: ~21 KB) holds what a
// query window of `downpore overlap` nearly always is - twenty to forty trimmed targets of fifteen to thirty seeds - and lets a CU
// hold seven of these waves instead of two; a window that does not fit it (more ints, a gap or offset beyond 16 bits, more than
// 32767 seeds in the round) is put on a list and done by a second launch with the large layout (int32, 57 KB: round 2's), and
// what does not fit that either is flagged for the host path as before.  Same code, same results, whichever layout ran.
// Round 4 - a third, huge layout (int32, 12 288 ints each for the trimmed and the reduced sequences: 144 KB, one wave per CU) for
// the dense-seed regime (k = 10: every trimmed target of a 1 kb window carries ~100 seeds, twenty to forty of them are 4-8 k ints;
// nearly every window overflowed the large layout and was done on the host from ~90 MB of fetched segments per round).  It is
// launched behind the large one once a round of the context has had a window the large layout could not hold.
// TIER: 0 = small, 1 = large, 2 = huge.  (HASH >= the most seeds T can hold: the open-addressed table must never fill up.)
template <int TIER>
struct CFCfg;
template <>
struct CFCfg<1> {
    typedef int32_t elem_t;
    enum { T = 4096, R = 4096, CONS = 1024, A = 256, M = 256, HASH = 4096, HSHIFT = 20 };
};
template <>
struct CFCfg<0> {
    typedef int16_t elem_t;
    enum { T = 2048, R = 2048, CONS = 256, A = 128, M = 128, HASH = 2048, HSHIFT = 21 };
};
template <>
struct CFCfg<2> {
    typedef int32_t elem_t;
    enum { T = 12288, R = 12288, CONS = 1024, A = 256, M = 256, HASH = 8192, HSHIFT = 19 };
};

template <int TIER>
struct CFWaveT {
    typedef CFCfg<TIER> C;
    typedef typename C::elem_t elem_t;
    elem_t T[C::T];               // trimmed sequences of the group, [gap, seed, ..., gap] each
    elem_t R[C::R];               // their Reduced() forms
    uint16_t Rmap[C::R / 2];
    union {
        uint32_t hash[C::HASH];  // (seed + 1) << 8 | shared << 7 | first sequence
        struct {
            uint16_t cmA[C::R / 2], cmB[C::R / 2];  // consensus matches of sequence s at [rb[s]/2 .. ): consensus index, index in T_s
            int32_t cons[C::CONS + 2];
            int32_t front[C::CONS / 8 + 2], backc[C::CONS / 8 + 2];
        };
    };
    int32_t GA[C::A + 2];     // G(i) = sum_{j=1..i} (a[2j] + k) over the forward query
    uint32_t mpair[C::M];     // pair slot of every match of the group
    int32_t tLen[64], tOff[64], tIns[64], tId[64];
    uint16_t tb[64], tN[64], rb[64], rN[64], mLen[64], cum[64];
    uint8_t tRc[64], ord[64];
};

// seeds/sequence.go:1190 GetBaseIndex for one part: MA/MB = its consensus matches, before = how many of them lie at or before
// aIndex (MA ascends: the caller bisects), CG = prefix sums of the consensus' (gap + k), sb = trimmed target
template <class E>
__device__ __forceinline__ void cf_base_index(const uint16_t* MA, const uint16_t* MB, int before, int aIndex, const int32_t* CG,
                                              const E* sb, int nb, int k, int& indexOut, int& basesOut) {
    if (before == 0) {
        int offset = CG[MA[0]] - CG[aIndex];  // sum over i = aIndex + 1 .. MA[0] of sa[2i] + k
        int bIndex = MB[0];
        for (int i = bIndex * 2; i > 0 && offset > 0; i -= 2) {
            offset -= sb[i] + k;
            bIndex--;
        }
        indexOut = bIndex;
        basesOut = -offset;
        return;
    }
    before--;
    int bIndex = MB[before];
    const int ma = MA[before];
    if (aIndex == ma) {
        indexOut = bIndex;
        basesOut = 0;
        return;
    }
    int offset = CG[aIndex] - CG[ma];  // sum over i = ma + 1 .. aIndex
    for (int i = bIndex * 2 + 2; i < nb && offset >= sb[i]; i += 2) {
        offset -= sb[i] + k;
        bIndex++;
    }
    indexOut = bIndex >= nb / 2 ? bIndex - 1 : bIndex;
    basesOut = offset;
}

struct ConsFullArgs {
    const uint32_t* recs;      // MRec[pairs] as 4 x u32: q, t, off, len
    const int32_t *ma, *mb;
    const uint32_t* pbase;     // [nq + 1]
    const int32_t* qsegs;
    const uint64_t* qoff;
    uint32_t n_groups;
    const dp_seq_ref* refs;
    const int32_t* segs;
    const dp_seq_meta* smeta;
    const int32_t* rc_of;
    const int32_t* anchors;    // [2 * pairs] match_anchor_kernel
    const int32_t* cover;      // [2 * pairs] likewise: GetBasesCovered of the match on the query and on the target side (INT32_MIN / + 1: a list the reference would panic on)
    const uint32_t* read_len;
    int k, overlap_size;
    dp_paf_rec* paf;           // [pairs]: lines of group g start at slot pbase[2g]
    uint32_t* ignore_ids;      // [pairs]: likewise
    dp_group_meta* gmeta;      // [n_groups]
    unsigned long long* dbg;   // DP_CONS_DEBUG: per group 8 time stamps (wall_clock64, 100 MHz)
    uint32_t flag_every;       // DP_CONS_FLAG_EVERY=n (test hook): every n-th window is left to the host path
    uint32_t out_cap;          // slots of paf / ignore_ids (a bound when the chaining stage's pair count is not known yet)
    uint32_t rec_cap;          // records the chaining stage's buffers hold (its pair count may exceed them: the stage is then repeated)
    const uint32_t* nseq_src;  // chunk count + overflow flag of dp_index_build_chunked (device), or null
    uint32_t* nseq_dst;        // ... and where the host reads them (pinned, with the rest of the output)
    const uint32_t *in_count, *in_list;  // windows this launch works through: how many, their numbers (null: every window)
    uint32_t *out_count, *out_list;      // windows this layout cannot hold are listed here for the next one (null: flagged for the host
                                         // path); the counters are zeroed by the anchors launch
    uint32_t spin_ticks;       // DP_CONS_SPIN=us (experiment): every window sleeps this long before it ends
    bool copy_nseq;            // this launch hands the chunk count of dp_index_build_chunked on to the host's block (the first one)
};

template <int TIER>
struct consensus_full_kernel {
    enum { THREADS = 64 };
    static __device__ void run(const ConsFullArgs A) {
    constexpr bool SMALL = TIER == 0;
    typedef CFWaveT<TIER> LW;
    typedef typename LW::C CF;
    typedef typename LW::elem_t elem_t;
    __shared__ LW L;
    const int lane = dp_lane();
    const int k = A.k;
    const u64 lanesBelow = (1ull << lane) - 1ull;
    // the first layout of a call does every window; each later one takes what its predecessor listed
    const bool fromList = A.in_list != nullptr;
    const uint32_t nWork = fromList ? min(*A.in_count, A.n_groups) : A.n_groups;
    if (A.copy_nseq && blockIdx.x == 0 && lane < 2 && A.nseq_src) A.nseq_dst[lane] = A.nseq_src[lane];
    // a window this layout cannot hold: listed for the next one (which writes its group record), flagged for the host after the last
#define CF_NOFIT(why_)                                                               \
    {                                                                                \
        if (A.out_list) {                                                            \
            gm.flag = 3; /* listed for the next layout (which overwrites this record) */ \
            if (lane == 0) {                                                         \
                A.out_list[atomicAdd(A.out_count, 1u)] = g;                          \
                A.gmeta[g] = gm;                                                     \
            }                                                                        \
        } else {                                                                     \
            gm.flag = 1;                                                             \
            gm.reserved = (why_); /* (diagnosis: DP_CONS_WHY=1 prints a histogram) */ \
            if (lane == 0) A.gmeta[g] = gm;                                          \
        }                                                                            \
        continue;                                                                    \
    }
    for (uint32_t wi = blockIdx.x; wi < nWork; wi += gridDim.x) {
        const uint32_t g = fromList ? A.in_list[wi] : wi;
        __builtin_amdgcn_wave_barrier();
        const uint32_t qf = 2 * g, qr = 2 * g + 1;
        const uint32_t P0 = A.pbase[qf], P1 = A.pbase[qr + 1];
        dp_group_meta gm = {P0, 0, 0, 0, 0, 0, 0, 0};
        if (P1 > A.rec_cap || P1 < P0) {  // (only with a pending chaining stage that overflowed: everything is redone)
            gm.flag = 2;
            if (lane == 0) A.gmeta[g] = gm;
            continue;
        }
#define CF_TICK(i_) if (A.dbg && lane == 0) A.dbg[16 * (size_t)g + (i_)] = wall_clock64()
        CF_TICK(0);
        // ---- 1. matches of the group
        int nm = 0;
        bool tooMany = false;
        for (uint32_t pb = P0; pb < P1; pb += 64) {
            const uint32_t p = pb + lane;
            const bool has = p < P1 && A.recs[4 * (size_t)p + 3] != 0;
            const u64 m = __ballot(has);
            if (has) {
                const int at = nm + __popcll(m & lanesBelow);
                if (at < CF::M) L.mpair[at] = p;
            }
            nm += __popcll(m);
        }
        gm.n_matches = (uint32_t)nm;
        if (nm > CF::M) tooMany = true;
        if (nm <= 1) {  // the reference only builds a consensus for queries with more than one hit (commands/overlap.go:170)
            if (lane == 0) A.gmeta[g] = gm;
            continue;
        }
        // ---- forward query: prefix sums of its gaps
        const int32_t* aSeg = A.qsegs + A.qoff[qf];
        const int nA = RFLc((int)(A.qoff[qf + 1] - A.qoff[qf]));
        const int sA = nA >> 1;
        if (sA > CF::A || sA < 1) tooMany = true;
        if (A.flag_every && (g % A.flag_every) == 0) tooMany = true;  // (test hook: goes the whole way through the layouts to the host path)
        if (tooMany) CF_NOFIT(1u)  // more matches / query seeds than the layout holds
        {
            int run = 0;
            for (int base = 0; base < sA; base += 64) {
                const int i = base + lane;  // G(i) for i >= 1 adds a[2i] + k
                const int v = (i >= 1 && i < sA) ? aSeg[2 * i] + k : 0;
                const int incl = wave_incl_sum(v);
                if (i < sA) L.GA[i] = run + incl;
                run += __shfl(incl, 63, 64);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int a_first = aSeg[0], a_last = aSeg[nA - 1], GA_end = L.GA[sA - 1];
        CF_TICK(1);
#ifdef CF_TRIMPROF
        unsigned long long tpT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp0 = 0;
#define CF_TP(i_) if (A.dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); const unsigned long long n_ = wall_clock64(); tpT[i_] += n_ - tp0; tp0 = n_; }
        if (A.dbg) tp0 = wall_clock64();
#else
#define CF_TP(i_)
#endif
        // ---- 2. per match: filter + Trimmed().  One lane per match; everything a lane does is its own little loop.
        int nseq = 0, tUsed = 0;
        bool bad = false, laneWide = false;
        uint32_t badWhy = 0;
        // algorithmic bytes of this window (reported in its group record): per match its record (16), its chain
        // (two int32 per matched seed), its anchors (8) and its target's view (16); per kept match the chunk's fields (16) and the
        // trimmed stretch of its segments; the forward query's segments; out: 40 per PAF line, 4 per SetIgnore id, 32 for the record
        unsigned algB = 0;
        for (int m0 = 0; m0 < nm; m0 += 64) {
            const int mi = m0 + lane;
            bool keep = false, laneBad = false;
            uint32_t laneWhy = 0;  // (diagnosis only: which check sent the window to the host path)
            int nT = 0, startSeed = 0, startOffset = 0, endOffset = 0, offset = 0, inset = 0, ns = 0;
            bool isRc = false;
            const int32_t* S = nullptr;
            uint32_t t = 0;
            int anc0 = 0, anc1 = 0;
            dp_seq_meta sm = {};
            if (mi < nm) {
                const uint32_t p = L.mpair[mi];
                const uint32_t rq = A.recs[4 * (size_t)p], off = A.recs[4 * (size_t)p + 2];
                t = A.recs[4 * (size_t)p + 1];
                const int len = (int)A.recs[4 * (size_t)p + 3];
                algB += 16u + 8u * (unsigned)len + 8u + 16u;
                isRc = rq == qr;
                // (asked for here, used after the chain walk: two trips to memory that used to start when the walk was over)
                anc0 = A.anchors[2 * (size_t)p], anc1 = A.anchors[2 * (size_t)p + 1];
                sm = A.smeta[t];
                const int32_t* MA = A.ma + off;
                const int32_t* MB = A.mb + off;
                const dp_seq_ref ref = A.refs[t];
                S = A.segs + ref.seg_off;
                ns = (int)ref.n_seeds;
                // GetBasesCovered on both sides (seeds/sequence.go:830): computed per matched pair by the anchors launch
                // (match_anchor_kernel, dp_overlap.hip), read here
                const int ca = A.cover[2 * (size_t)p], cb = A.cover[2 * (size_t)p + 1];
                if (ca == DP_NO_ANCHOR) laneBad = true, laneWhy = 10u;           // first pair outside the query / the target
                else if (ca == DP_NO_ANCHOR + 1) laneBad = true, laneWhy = 11u;  // a later one: the reference would panic inside GetBasesCovered
                CF_TP(0)
                CF_TP(1)
                if (!laneBad && ca >= 25 && cb >= 25) {
                    // indices in the forward query / in X (X = the target, or its reverse complement for a match of the rc query)
                    const int m_first = MA[0], m_last = MA[len - 1], t_first = MB[0], t_last = MB[len - 1];
                    const int a0 = isRc ? sA - 1 - m_last : m_first, aL = isRc ? sA - 1 - m_first : m_last;
                    startSeed = isRc ? ns - 1 - t_last : t_first;
                    int endSeed = isRc ? ns - 1 - t_first : t_last;
                    startOffset = a_first + L.GA[a0];          // forward query's GetSeedOffset(MatchA[0])
                    endOffset = a_last + GA_end - L.GA[aL];    // ... GetSeedOffsetFromEnd(MatchA[last])
                    // X.GetSeedOffset(startSeed) / X.GetSeedOffsetFromEnd(endSeed) from the match's anchors on the forward
                    // target (match_anchor_kernel): R.seedOffset(i) = S.seedOffsetFromEnd(ns-1-i) and vice versa
                    int anchorStart = isRc ? anc1 : anc0, anchorEnd = isRc ? anc0 : anc1;
                    if (anchorStart == DP_NO_ANCHOR || anchorEnd == DP_NO_ANCHOR) laneBad = true, laneWhy = 12u;
                    // X.seg[2t] = S[2t] (forward) or S[2(ns - t)] (reverse complement)
                    while (startSeed > 0) {
                        const int gp = (isRc ? S[2 * (ns - startSeed)] : S[2 * startSeed]) + k;
                        if (startOffset < gp) break;
                        startOffset -= gp;
                        anchorStart -= gp;
                        startSeed--;
                    }
                    while (endSeed < ns - 1) {
                        const int gp = (isRc ? S[2 * (ns - endSeed - 1)] : S[2 * endSeed + 2]) + k;
                        if (endOffset < gp) break;
                        endOffset -= gp;
                        anchorEnd -= gp;
                        endSeed++;
                    }
                    if (startSeed > endSeed) laneBad = true, laneWhy = laneWhy ? laneWhy : 13u;
                    offset = anchorStart - startOffset;
                    inset = anchorEnd - endOffset;
                    nT = 2 * (endSeed - startSeed) + 3;
                    keep = !laneBad;
                }
            }
            CF_TP(2)
            if (__ballot(laneBad)) {
                bad = true;
                badWhy = (uint32_t)__shfl((int)laneWhy, __builtin_ctzll(__ballot(laneBad)), 64);
                break;
            }
            const u64 keepMask = __ballot(keep);
            const int incl = wave_incl_sum(keep ? nT : 0);
            const int total = __shfl(incl, 63, 64);
            const int nKeep = __popcll(keepMask);
            if (nseq + nKeep > 64 || tUsed + total > CF::T) {
                bad = true;
                badWhy = nseq + nKeep > 64 ? 8u : 9u;
                break;
            }
            CF_TP(3)
            if (keep) {
                const int sq = nseq + __popcll(keepMask & lanesBelow);
                const int tb = tUsed + incl - nT;
                algB += 16u + 4u * (unsigned)nT;
                const int nB = 2 * ns + 1;
                int wide = 0;  // (small layout: a value that does not fit 16 bits sends the window to the large one)
                for (int j0 = 0; j0 < nT; j0 += 16) {  // (sixteen loads in flight per trip, then - reverse complement - their look-ups)
                    int vv[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const int x = 2 * startSeed + j0 + u;  // index in X
                        vv[u] = j0 + u < nT ? (isRc ? S[nB - 1 - x] : S[x]) : 0;
                    }
                    if (isRc) {
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            const int x = 2 * startSeed + j0 + u;
                            if (j0 + u < nT && (x & 1)) vv[u] = A.rc_of[vv[u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        if (j0 + u >= nT) continue;
                        L.T[tb + j0 + u] = (elem_t)vv[u];
                        wide |= (vv[u] != (int)(elem_t)vv[u]);
                    }
                }
                L.T[tb] = (elem_t)startOffset;
                L.T[tb + nT - 1] = (elem_t)endOffset;
                wide |= (startOffset != (int)(elem_t)startOffset) | (endOffset != (int)(elem_t)endOffset);
                if (SMALL && wide) laneWide = true;
                L.tb[sq] = tb;
                L.tN[sq] = nT;
                L.tLen[sq] = sm.length - offset - inset;
                L.tOff[sq] = isRc ? sm.offset + inset : sm.offset + offset;
                L.tIns[sq] = isRc ? sm.inset + offset : sm.inset + inset;
                L.tId[sq] = (int32_t)sm.read;
                L.tRc[sq] = isRc ? 1 : 0;
            }
            CF_TP(4)
            nseq += nKeep;
            tUsed += total;
        }
        if (bad || __ballot(laneWide)) CF_NOFIT(bad ? badWhy : 3u)  // a list the reference would panic on (2), > 64 sequences (8), trimmed ints beyond T (9) | 16-bit overflow
        if (nseq <= 1) {  // BuildConsensus needs more than one sequence (combine.go:183)
            if (lane == 0) A.gmeta[g] = gm;
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        CF_TICK(2);
        // ---- 3. seeds shared by >= 2 sequences (GetSharedIDs(.., 2, true)), Reduced() of every sequence (seeds/sequence.go:85).
        //         Round 4: the hash is filled and asked by ALL lanes over ALL seeds of the window (seed f of the window belongs to
        //         sequence own[f]; a lane takes every 64th seed - thirteen of a mean window's eight hundred - where a lane per
        //         sequence walked the forty to sixty of the longest one, one hash probe after the other), the verdict goes into the
        //         seed's own top bit, and ONE walk per sequence - loads eight seeds ahead, no probes - writes the reduced form
        //         (its room is sized for "every seed kept", so nobody counts first).
        {
            const uint4 z = {0u, 0u, 0u, 0u};
            for (int i = lane; i < CF::HASH / 4; i += 64) ((uint4*)L.hash)[i] = z;
        }
        const bool mine = lane < nseq;
        const int tb_ = mine ? L.tb[lane] : 0;
        const int nsT_ = mine ? (L.tN[lane] >> 1) : 0;
        constexpr uint32_t SHARED_BIT = 1u << (8 * (int)sizeof(elem_t) - 1);  // (seed ids stay below it: 15 bits in the small layout, 31 otherwise)
        uint8_t* const own = (uint8_t*)L.Rmap;  // (Rmap is written by the walk below, when nobody asks for owners any more)
        int totalSeeds;
        {
            const int incl = wave_incl_sum_dpp(nsT_);
            totalSeeds = __builtin_amdgcn_readlane(incl, 63);
            if (mine) L.cum[lane] = (uint16_t)(incl - nsT_);
            for (int sq_ = 0; sq_ < nseq; sq_++) {
                const int c0 = __builtin_amdgcn_readlane(incl - nsT_, sq_), n0 = __builtin_amdgcn_readlane(nsT_, sq_);
                for (int i = lane; i < n0; i += 64) own[c0 + i] = (uint8_t)sq_;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // fill: eight seeds per lane and trip - their owners, then their places, then the seeds themselves are loaded together,
        // the eight first probes (a compare-and-swap each) travel together; only a probe that met another seed walks on alone
        for (int f0 = 0; f0 < totalSeeds; f0 += 512) {
            int sq8[8], at8[8];
            uint32_t sd8[8], old8[8], h8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int f = f0 + 64 * u + lane;
                sq8[u] = f < totalSeeds ? (int)own[f] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int f = f0 + 64 * u + lane;
                at8[u] = sq8[u] >= 0 ? (int)L.tb[sq8[u]] + 2 * (f - (int)L.cum[sq8[u]]) + 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) sd8[u] = sq8[u] >= 0 ? (uint32_t)L.T[at8[u]] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                h8[u] = (sd8[u] * 2654435761u) >> CF::HSHIFT;
                old8[u] = sq8[u] >= 0 ? atomicCAS(&L.hash[h8[u]], 0u, ((sd8[u] + 1) << 8) | (uint32_t)sq8[u]) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (sq8[u] < 0) continue;
                uint32_t e = old8[u], h = h8[u];
                const uint32_t mineV = ((sd8[u] + 1) << 8) | (uint32_t)sq8[u];
                while (e != 0) {  // (0: the slot was free and is this seed's now)
                    if ((e >> 8) == sd8[u] + 1) {
                        if ((e & 63u) != (uint32_t)sq8[u] && !(e & 128u)) atomicOr(&L.hash[h], 128u);
                        break;
                    }
                    h = (h + 1) & (CF::HASH - 1);
                    e = atomicCAS(&L.hash[h], 0u, mineV);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ask: the same walk over all seeds; a shared seed gets its top bit set where it lies
        for (int f0 = 0; f0 < totalSeeds; f0 += 512) {
            int sq8[8], at8[8];
            uint32_t sd8[8], e8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int f = f0 + 64 * u + lane;
                sq8[u] = f < totalSeeds ? (int)own[f] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int f = f0 + 64 * u + lane;
                at8[u] = sq8[u] >= 0 ? (int)L.tb[sq8[u]] + 2 * (f - (int)L.cum[sq8[u]]) + 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) sd8[u] = sq8[u] >= 0 ? (uint32_t)L.T[at8[u]] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) e8[u] = sq8[u] >= 0 ? L.hash[(sd8[u] * 2654435761u) >> CF::HSHIFT] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (sq8[u] < 0) continue;
                uint32_t e = e8[u], h = (sd8[u] * 2654435761u) >> CF::HSHIFT;
                while (e != 0 && (e >> 8) != sd8[u] + 1) {
                    h = (h + 1) & (CF::HASH - 1);
                    e = L.hash[h];
                }
                if (e & 128u) L.T[at8[u]] = (elem_t)(sd8[u] | SHARED_BIT);
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            // room for sequence s in R: as if every seed were kept (2 * seeds + 2 ints: even, Rmap / cm arrays are indexed by rb >> 1)
            const int slot = mine ? 2 * nsT_ + 2 : 0;
            const int incl = wave_incl_sum_dpp(slot);
            const int totalR = __builtin_amdgcn_readlane(incl, 63);
            if (totalR + 2 >= CF::R) CF_NOFIT(4u)
            if (mine) {
                const int rb = incl - slot;
                int prev = -1, r = 0, offset = nsT_ > 0 ? (int)L.T[tb_] : 0;
                for (int i0 = 0; i0 < nsT_; i0 += 8) {
                    int sv[8], gp[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool in = i0 + u < nsT_;
                        sv[u] = in ? (int)L.T[tb_ + 2 * (i0 + u) + 1] : 0;
                        gp[u] = in ? (int)L.T[tb_ + 2 * (i0 + u) + 2] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (i0 + u >= nsT_) continue;
                        const int seed = (int)((uint32_t)sv[u] & (SHARED_BIT - 1u));
                        if (((uint32_t)sv[u] & SHARED_BIT) && seed != prev) {
                            L.R[rb + 2 * r] = (elem_t)offset;
                            laneWide |= SMALL && offset != (int)(elem_t)offset;
                            L.R[rb + 2 * r + 1] = (elem_t)seed;
                            L.Rmap[(rb >> 1) + r] = (uint16_t)(i0 + u);
                            r++;
                            offset = gp[u];
                            prev = seed;
                        } else {
                            offset += gp[u] + k;
                        }
                    }
                }
                if (r >= 1) {
                    L.R[rb + 2 * r] = (elem_t)offset;
                    laneWide |= SMALL && offset != (int)(elem_t)offset;
                }
                L.rb[lane] = rb;
                L.rN[lane] = r >= 1 ? 2 * r + 1 : 0;
            }
        }
        if (__ballot(laneWide)) CF_NOFIT(5u)
        __builtin_amdgcn_wave_barrier();
        CF_TICK(3);
        // ---- 4. the alignment (multiAligner.Consensus :52-247); lane i owns sequence i.  hash[] is dead from here on.
        const int b = mine ? L.rb[lane] : 0;
        const int sl = mine ? L.rN[lane] : 0;
        const int mbase = b >> 1;
        const elem_t* S = L.R;
        int pos = -1, offs = 0, gaps = 50, supported = 0, dist = 0, mlen = 0, clen = 0;
        const int kLim = 1 << 28;
        const int ns = nseq;
        unsigned dbgUni = 0, dbgGen = 0, dbgProp = 0;  // (DP_CONS_DEBUG: uniform steps, general steps, proposers looked at)
        unsigned long long dbgT[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (general steps: ticks before the scan, in it, in the selection, in the update; scan runs, walk trips of the scan, of the update)
        unsigned long long dbgT0 = 0;
#define CF_T(i_) if (A.dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wall_clock64(); dbgT[i_] += n_ - dbgT0; dbgT0 = n_; }
        for (;;) {
            int near = 100000;
            const int p2s = pos + 1;
            const bool okS = sl > 0 && p2s < sl / 2;
            // (the next seed of the sequence and its distance: loaded once per step, together - the uniform test, the proposal, the
            // first probe of every search and of the update all look at this pair; each used to fetch it again, a trip to the LDS
            // of ~130 cycles on the step's critical path every time)
            const int sdNext = okS ? S[b + p2s * 2 + 1] : -1;
            const int od = okS ? S[b + p2s * 2] - offs : 0;
            {
                // Uniform step: every sequence that still has a seed is in step (gap 0) and shows the same seed at the same
                // distance.  The general code below then does nothing but agree: each of them proposes (d = o0 < near, which
                // only drops to maxD(o0) > o0), finds the seed in every other one at once (o0 lies inside gapRange(o0) for
                // every -k < o0), so supported = nValid >= 2 and dist / supported = o0 exactly; the first proposer wins the
                // selection and the update finds the seed at pos + 1 of every sequence.  Same state, consensus and matches
                // (the host mirror takes the same short cut, host_seq.cpp; 80 % of the steps at e = 0).
                const u64 okMask = __ballot(okS);
                const int nValid = __popcll(okMask);
                if (nValid >= 2) {
                    const int f = __builtin_ctzll(okMask);
                    const int o0 = CA_RL(od, f), sd0 = CA_RL(sdNext, f);
                    const bool same = !okS || (od == o0 && sdNext == sd0 && gaps == 0);
                    if (__ballot(same) == ~0ull && o0 > -k && o0 < 100000) {
                        if (clen + 2 >= CF::CONS) {
                            bad = true;
                            break;
                        }
                        if (lane == 0) {
                            L.cons[clen] = o0;
                            L.cons[clen + 1] = sd0;
                        }
                        clen += 2;
                        dbgUni++;
                        if (okS) {
                            pos = p2s;
                            offs = 0;
                            dist = nValid * o0;  // what the support scan leaves behind (a later step may read it, :170-176)
                            L.cmA[mbase + mlen] = (uint16_t)(clen / 2 - 1);
                            L.cmB[mbase + mlen] = L.Rmap[mbase + p2s];
                            mlen++;
                        }
                        continue;
                    }
                }
            }
            supported = 0;
            dbgGen++;
            if (A.dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); dbgT0 = wall_clock64(); }
            const bool fin = !mine || sl == 0 || pos >= (sl - 1) / 2 - 1;
            int fCount = __popcll(__ballot(fin && mine));
            int d = 0, nextSeed = 0, minD = 0, maxD = 0;
            if (!fin) {  // (!fin == okS: pos + 1 < seeds; so the proposal is the pair above)
                d = od;
                dist = d;
                nextSeed = sdNext;
                ca_gap_range(d + gaps, k, minD, maxD);
                minD -= gaps;
                maxD -= gaps;
            }
            if (od >= kLim || od <= -kLim || gaps >= kLim || d >= kLim || d <= -kLim || dist >= kLim || dist <= -kLim) bad = true;
            if (__ballot(bad)) break;
            // Support of every proposer, in sequence order (`near` shrinks as proposers are seen, :70-76).  Proposers that follow
            // each other with the same proposal {d, seed, window} - at e = 0 nearly all of them - are one step here: the first
            // sets `near`, the others meet the very same test after it (near does not move again inside the run), the search
            // (which depends on the proposal and the searching lane only) is made once, every accepted member takes its numbers.
            CF_T(0)
            unsigned long long cand = __ballot(!fin);
            bool fnd = false, memoOk = false;
            int val = 0, memoD = 0, memoSeed = 0, memoMin = 0, memoMax = 0, cntAll = 0, sumAll = 0;
            while (cand) {
                dbgT[4]++;
                const int i = __builtin_ctzll(cand);
                const int di = CA_RL(d, i);
                const int seedI = CA_RL(nextSeed, i), minI = CA_RL(minD, i), maxI = CA_RL(maxD, i);
                const unsigned long long same = __ballot(!fin && d == di && nextSeed == seedI && minD == minI && maxD == maxI) & cand;
                const unsigned long long others = cand & ~same;
                const unsigned long long run = others ? (same & ((1ull << __builtin_ctzll(others)) - 1ull)) : same;  // (holds bit i)
                cand &= ~run;
                if (!(di < near && di > -k)) continue;  // (so would every member of the run: same d, same near)
                if (near > maxI) near = maxI;
                const unsigned long long accepted = di < near ? run : (1ull << i);  // the members after the first meet the new near
                if (!(memoOk && memoD == di && memoSeed == seedI && memoMin == minI && memoMax == maxI)) {  // (else: the search just made)
                memoOk = true, memoD = di, memoSeed = seedI, memoMin = minI, memoMax = maxI;
                dbgProp++;
                fnd = false;
                val = 0;
                if (okS) {
                    int min2, max2;
                    ca_gap_range(di + gaps, k, min2, max2);
                    if (min2 > minI) min2 = minI;
                    if (max2 < maxI) max2 = maxI;
                    int p2 = p2s, otherD = od, curSeed = sdNext;
                    while (otherD < min2 && p2 < sl / 2) {
                        if (A.dbg) dbgT[5]++;
                        p2++;
                        otherD += S[b + p2 * 2] + k;
                        curSeed = S[b + p2 * 2 + 1];  // (one int past the sequence when p2 reaches its end: inside its room in R, never used)
                    }
                    while (otherD < max2 && p2 < sl / 2) {
                        if (curSeed == seedI) {
                            fnd = true;
                            val = otherD;
                            break;
                        }
                        p2++;
                        otherD += S[b + p2 * 2] + k;
                        curSeed = S[b + p2 * 2 + 1];
                    }
                }
                cntAll = __popcll(__ballot(fnd));
                sumAll = wave_sum_dpp(fnd ? val : 0);
                }
                if ((accepted >> lane) & 1ull) {
                    supported = 1 + cntAll - (fnd ? 1 : 0);
                    dist += sumAll - (fnd ? val : 0);
                }
            }
            CF_T(1)
            if (fCount >= ns) break;
            int minseed = -1, mindist = 0, minsup = 0, selMin = 0, selMax = 0;
            // (mean distance and window of every supported proposer at once - one vector division per step instead of one per
            // proposer; the walk over them keeps the order dependence of :141-160)
            const int myDv = supported > 1 ? dist / supported : dist;
            {
                int myMin, myMax;
                ca_gap_range(myDv + gaps, k, myMin, myMax);
                myMin -= gaps;
                myMax -= gaps;
                int winner = -1;
                unsigned long long sup = __ballot(supported > 1);
                while (sup) {
                    const int i = __builtin_ctzll(sup);
                    const int si = CA_RL(supported, i);
                    const int dv = CA_RL(myDv, i);
                    const int seed = CA_RL(nextSeed, i);
                    // (supported proposers that follow each other with the same {seed, support, mean distance} are decided by the
                    // first of them: taken, the rest find their own numbers in place; not taken, the rest meet the same state)
                    const unsigned long long same = __ballot(supported == si && myDv == dv && nextSeed == seed) & sup;
                    const unsigned long long others = sup & ~same;
                    sup &= ~(others ? (same & ((1ull << __builtin_ctzll(others)) - 1ull)) : same);
                    if (minseed == -1 || (minseed == seed && si > minsup) || (minseed != seed && mindist > dv)) {
                        minsup = si;
                        mindist = dv;
                        minseed = seed;
                        winner = i;
                    }
                }
                if (winner >= 0) {
                    selMin = CA_RL(myMin, winner);
                    selMax = CA_RL(myMax, winner);
                }
            }
            CF_T(2)
            if (minseed == -1) {
                dbgT[7]++;
                const int dvv = myDv;
                const bool can = mine && sl > 0 && pos < ns / 2;
                int best = can ? dvv : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
                if (best >= 100000) break;
                const int minIndex = __builtin_ctzll(__ballot(can && dvv == best));
                if (mine && sl > 0) {
                    gaps += best;
                    offs += best;
                }
                if (lane == minIndex) {
                    gaps = 0;
                    offs = 0;
                    pos++;
                }
                continue;
            }
            if (clen + 2 >= CF::CONS) {
                bad = true;
                break;
            }
            if (lane == 0) {
                L.cons[clen] = mindist;
                L.cons[clen + 1] = minseed;
            }
            clen += 2;
            bool finC = true;
            if (mine && sl > 0) {
                int matchDex = pos + 1;
                if (matchDex < sl / 2) {
                    int min2, max2;
                    ca_gap_range(mindist + gaps, k, min2, max2);
                    if (min2 > selMin) min2 = selMin;
                    if (max2 < selMax) max2 = selMax;
                    int otherD = od, curSeed = sdNext;  // (matchDex == p2s)
                    while (otherD < min2 && matchDex < sl / 2) {
                        matchDex++;
                        otherD += S[b + matchDex * 2] + k;
                        curSeed = S[b + matchDex * 2 + 1];
                    }
                    bool found = false;
                    while (otherD < max2 && matchDex < sl / 2) {
                        if (curSeed == minseed) {
                            pos = matchDex;
                            offs = 0;
                            gaps = 0;
                            L.cmA[mbase + mlen] = (uint16_t)(clen / 2 - 1);
                            L.cmB[mbase + mlen] = L.Rmap[mbase + matchDex];  // index in the trimmed sequence
                            mlen++;
                            found = true;
                            break;
                        }
                        matchDex++;
                        otherD += S[b + matchDex * 2] + k;
                        curSeed = S[b + matchDex * 2 + 1];
                    }
                    finC = false;
                    if (!found) {
                        if (A.dbg) dbgT[6]++;
                        gaps += mindist;
                        offs += mindist;
                        int p = pos;
                        while (p < sl / 2 && offs > S[b + p * 2 + 2] + 50) {
                            offs -= S[b + p * 2 + 2] + k;
                            p++;
                            pos++;
                        }
                        if (p >= sl / 2) finC = true;
                    }
                }
            }
            CF_T(3)
            if (__popcll(__ballot(finC && mine)) >= ns) break;
        }
        if (A.dbg) {
            int t5 = (int)dbgT[5], t6 = (int)dbgT[6];
            for (int o = 32; o > 0; o >>= 1) t5 = max(t5, __shfl_xor(t5, o, 64)), t6 = max(t6, __shfl_xor(t6, o, 64));
            dbgT[5] = (unsigned long long)t5, dbgT[6] = (unsigned long long)t6;
        }
        if (A.dbg && lane == 0) {
            A.dbg[16 * (size_t)g + 6] = ((unsigned long long)dbgUni << 32) | dbgGen;
            A.dbg[16 * (size_t)g + 7] = ((unsigned long long)nseq << 32) | dbgProp;
#ifdef CF_TRIMPROF
            for (int i = 0; i < 8; i++) A.dbg[16 * (size_t)g + 8 + i] = tpT[i];
#else
            for (int i = 0; i < 8; i++) A.dbg[16 * (size_t)g + 8 + i] = dbgT[i];
#endif
        }
        if (__ballot(bad)) CF_NOFIT(6u)  // consensus longer than CONS, a value outside the safe range
        if (lane == 0) L.cons[clen] = 0;
        if (mine) L.mLen[lane] = mlen;
        __builtin_amdgcn_wave_barrier();
        CF_TICK(4);
#ifdef CF_P5PROF  // (diagnosis build: make EXTRA=-DCF_P5PROF; sub-phase ticks of phase 5 take the place of the general steps' in the debug record)
        unsigned long long p5T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p50 = 0;
#define CF_P5(i_) if (A.dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); const unsigned long long n_ = wall_clock64(); p5T[i_] += n_ - p50; p50 = n_; }
        if (A.dbg) p50 = wall_clock64();
#else
#define CF_P5(i_)
#endif
        // ---- 5. parts with fewer than 3 matched seeds leave by swap-with-last, from the back (:258-266)
        // (when the walk reaches slot i, ord[i] still is i - a removal writes the slot being visited, nothing else - so the test needs no
        // load: one ballot says who leaves, lane 0 moves the last one into each hole, highest hole first)
        int np = nseq;
        {
            const bool gone = lane < nseq && (L.rN[lane] == 0 || L.mLen[lane] < 3);
            u64 holes = __ballot(gone);
            if (lane < nseq) L.ord[lane] = (uint8_t)lane;
            __builtin_amdgcn_wave_barrier();
            if (holes) {
                if (lane == 0) {
                    int last = nseq;
                    while (holes) {
                        const int i = 63 - __clzll((long long)holes);
                        L.ord[i] = L.ord[last - 1];
                        last--;
                        holes ^= 1ull << i;
                    }
                }
                np = nseq - __popcll(__ballot(gone));
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (np <= 1) {
            if (lane == 0) A.gmeta[g] = gm;
            continue;
        }
        // trimToBestSeed (combine.go:21-111)
        const int cS = clen / 2;                  // ms[0].SeqA.GetNumSeeds()
        const int upto = cS / 4;
        const int minMatch = np < 5 ? np : 5;
        for (int i = lane; i < upto; i += 64) {
            L.front[i] = 0;
            L.backc[i] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        CF_P5(0)
        const bool part = lane < np;
        const int sq = part ? L.ord[lane] : 0;
        const int mb0 = part ? (L.rb[sq] >> 1) : 0;
        int pl = part ? L.mLen[sq] : 0;
        const int myId = part ? L.tId[sq] : 0;
        const int mySeqLen = part ? (int)A.read_len[myId] : 0;  // (asked for here, needed at the very end)
        // CG[i] = sum_{j = 1 .. i} (cons[2j] + k), i = 0 .. cS: every walk over the consensus' gaps below is a difference of two of these
        // (in the room of the reduced sequences, which nobody reads any more)
        int32_t* CG = (int32_t*)L.R;
        static_assert(sizeof(L.R) >= (CF::CONS / 2 + 2) * sizeof(int32_t), "the consensus' prefix sums live in the reduced sequences' room");
        {
            int run = 0;
            for (int base = 0; base <= cS; base += 64) {
                const int i = base + lane;
                const int v = (i >= 1 && i <= cS) ? L.cons[2 * i] + k : 0;
                const int incl = wave_incl_sum(v);
                if (i <= cS) CG[i] = run + incl;
                run += __shfl(incl, 63, 64);
            }
        }
        if (part) {
            // a part's consensus indices ascend strictly (one consensus seed per step of the alignment): the front histogram sees a
            // prefix of them, the back one a suffix
            const uint16_t* MA = L.cmA + mb0;
            for (int j = 0; j < pl; j++) {
                const int a = MA[j];
                if (a >= upto) break;
                atomicAdd(&L.front[a], 1);
            }
            for (int j = pl - 1; j >= 1; j--) {
                const int b = cS - 1 - (int)MA[j];
                if (b >= upto) break;
                if (b >= 0) atomicAdd(&L.backc[b], 1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        CF_P5(1)
        int bestCount = 0, bestScore = 0, bestIndex = upto, backCount = 0, backScore = 0, backIndex = cS - upto - 1;
        for (int i = 0; i < upto; i++) {
            const int count = L.front[i], bCount = L.backc[i];
            if (count - i >= bestScore || (bestCount < minMatch && count >= minMatch)) {
                bestCount = count;
                bestScore = count - i;
                bestIndex = i;
            }
            if (bCount - i >= backScore || (backCount < minMatch && bCount >= minMatch)) {
                backCount = bCount;
                backScore = bCount - i;
                backIndex = cS - 1 - i;
            }
        }
        // consensus := Trimmed(0, bestIndex, 0, backIndex): only its extent matters below (gaps at both ends become 0)
        int cStart = bestIndex, cEnd = backIndex;
        while (cStart > 0 && 0 >= L.cons[2 * cStart] + k) cStart--;
        while (cEnd < cS - 1 && 0 >= L.cons[2 * cEnd + 2] + k) cEnd++;
        const int naC = 2 * cEnd + 3 - 2 * cStart;  // ints of the trimmed consensus
        int pOffset = 0, pInset = 0, pN = 0, ident = 0, badBack = 0;
        bool panic = false, partBad = false;
        CF_P5(2)
        if (part) {
            const uint16_t* MA = L.cmA + mb0;
            const uint16_t* MB = L.cmB + mb0;
            const elem_t* sb = L.T + L.tb[sq];
            const int nT = L.tN[sq], nsT = nT >> 1;
            int index, bases, bIndex, backBases;
            // GetBaseIndex's "matches up to aIndex", for both ends at once, by bisection
            int lo1 = 0, n1 = pl, lo2 = 0, n2 = pl;
            while ((n1 | n2) != 0) {
                const int h1 = n1 >> 1, h2 = n2 >> 1;
                const int v1 = MA[lo1 + h1], v2 = MA[lo2 + h2];  // (a finished search reads MA[lo] and moves nothing)
                const bool le1 = n1 > 0 && v1 <= bestIndex, le2 = n2 > 0 && v2 <= backIndex;
                lo1 = le1 ? lo1 + h1 + 1 : lo1;
                n1 = le1 ? n1 - h1 - 1 : h1;
                lo2 = le2 ? lo2 + h2 + 1 : lo2;
                n2 = le2 ? n2 - h2 - 1 : h2;
            }
            cf_base_index(MA, MB, lo1, bestIndex, CG, sb, nT, k, index, bases);
            cf_base_index(MA, MB, lo2, backIndex, CG, sb, nT, k, bIndex, backBases);
            CF_P5(3)
            if (bases > -k && index < nsT - 1) {
                bases = (sb[2 * index + 2] + k) - bases;
                index++;
            } else if (bases < 0) {
                bases = -bases + k;
            }
            // parts[j] = match.SeqB.Trimmed(bases, index, backBases, bIndex)
            int startSeed = index, endSeed = bIndex, startOffset = bases, endOffset = backBases;
            while (startSeed > 0 && startOffset >= sb[2 * startSeed] + k) {
                startOffset -= sb[2 * startSeed] + k;
                startSeed--;
            }
            while (endSeed < nsT - 1 && endOffset >= sb[2 * endSeed + 2] + k) {
                endOffset -= sb[2 * endSeed + 2] + k;
                endSeed++;
            }
            if (startSeed < 0 || endSeed >= nsT || startSeed > endSeed) {
                partBad = true;
            } else {
                int so = sb[0], se = sb[nT - 1];
                for (int j = 1; j <= startSeed; j++) so += sb[2 * j] + k;
                for (int j = endSeed + 1; j <= nsT - 1; j++) se += sb[2 * j] + k;
                const int o2 = so - startOffset, i2 = se - endOffset;
                const bool rc = L.tRc[sq] != 0;
                pOffset = rc ? L.tOff[sq] + i2 : L.tOff[sq] + o2;
                pInset = rc ? L.tIns[sq] + o2 : L.tIns[sq] + i2;
                pN = 2 * (endSeed - startSeed) + 3;
                CF_P5(4)
                // the part's matches: keep [front, back], rebase (combine.go:84-110)
                int front = 0;
                while (front < pl && (int)MB[front] < index) front++;
                int back = pl - 1;
                while (back >= 0 && (int)MB[back] > bIndex) back--;
                if (back + 1 > pl || back < front) {  // "Bad back:" diagnostic, suppressed and counted (HISTORY.md 2.5)
                    badBack = 1;
                    if (back + 1 < front) back = front - 1;
                }
                // GetBasesCovered(k) of the trimmed match, consensus side (commands/overlap.go:214): this part's line uses the
                // PREVIOUS part's match (Matches[id-1]); computed here, handed to the next lane below
                const int Ln = back - front + 1;
                if (Ln <= 0) {
                    panic = true;
                } else {
                    int countA = Ln * k;
                    // a pair of neighbours adds the consensus' gaps between them where those are negative: tc = the trimmed consensus
                    // (cons from cStart on, its two end gaps 0), so d1 = CG difference - k, less the end gap when the pair ends on it.
                    // Four pairs a trip: their loads travel together, only the running sum and the "previous" indices are carried
                    const int mEnd = (naC - 1) >> 1;
                    int prevA = (int)MA[front] - bestIndex, prevB = (int)MB[front] - index;
                    int cgPrev = CG[min(max(cStart + prevA, 0), cS)];
                    for (int j = front + 1; j <= back; j += 4) {
                        int sv[4], s2v[4], cg[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int jj = min(j + u, back);
                            sv[u] = (int)MA[jj] - bestIndex;
                            s2v[u] = (int)MB[jj] - index;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) cg[u] = CG[min(max(cStart + sv[u], 0), cS)];
                        const int endGap = L.cons[2 * (cStart + mEnd)];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (j + u <= back) {
                                const int s = sv[u], s2 = s2v[u];
                                if (s * 2 >= naC || s2 * 2 >= pN || prevA * 2 + 2 >= naC || prevB * 2 + 2 >= pN || prevA < 0 || prevB < 0) panic = true;
                                int d1;
                                if (s > prevA) d1 = cg[u] - cgPrev - k - (s == mEnd ? endGap : 0);
                                else d1 = (prevA + 1 == mEnd || panic) ? 0 : L.cons[2 * (cStart + prevA + 1)];  // (does not happen: MA ascends)
                                if (d1 < 0) countA += d1;
                                prevA = s;
                                prevB = s2;
                                cgPrev = cg[u];
                            }
                        }
                    }
                    ident = panic ? 0 : countA;
                }
            }
        }
        CF_P5(5)
        if (__ballot(partBad)) CF_NOFIT(7u)
        // contig + PAF numbers (combine.go:113-133, commands/overlap.go:199-231)
        const int myRc = part ? L.tRc[sq] : 0;
        const int myStart = pOffset, myLen = mySeqLen - pOffset - pInset;
        const int q_id = __shfl(myId, 0, 64), q_rc = __shfl(myRc, 0, 64), q_len = __shfl(mySeqLen, 0, 64);
        const int q_start = __shfl(myStart, 0, 64), q_l = __shfl(myLen, 0, 64);
        const int identPrev = __shfl_up(ident, 1, 64);
        const bool panicPrev = __shfl_up(panic ? 1 : 0, 1, 64) != 0;
        const bool line = part && lane >= 1;
        bool ign = false;
        if (lane == 0) ign = q_len <= A.overlap_size * 2;
        if (line) {
            const int start = myStart, end = myStart + myLen;
            int covered = A.overlap_size;
            if (end - start > A.overlap_size) covered = end - start;
            ign = (long long)mySeqLen * 9 <= (long long)covered * 10;
            dp_paf_rec r;
            r.q_read = (uint32_t)q_id;
            r.t_read = (uint32_t)myId;
            r.q_len = q_len;
            r.q_start = q_start;
            r.q_end = q_start + q_l;
            r.t_len = mySeqLen;
            r.t_start = start;
            r.t_end = end;
            r.ident = identPrev;
            r.minus = (q_rc != myRc) ? 1u : 0u;
            if (P0 + (uint32_t)(lane - 1) < A.out_cap) A.paf[P0 + (uint32_t)(lane - 1)] = r;
        }
        const u64 ignMask = __ballot(ign && part);
        if (ign && part) {
            const uint32_t at = P0 + (uint32_t)__popcll(ignMask & lanesBelow);
            if (at < A.out_cap) A.ignore_ids[at] = (uint32_t)myId;
        }
        // (slots are the pairs of the window's two queries: P0 .. P1; out_cap is a bound when the pair count is not known yet)
        if (P1 > A.out_cap) gm.flag = 2;  // output did not fit: the caller repeats the call with the exact size
        gm.n_lines = (uint32_t)(np - 1);
        gm.n_ignore = (uint32_t)__popcll(ignMask);
        gm.reserved = (uint32_t)wave_sum((int)algB) + 4u * (uint32_t)nA + 40u * gm.n_lines + 4u * gm.n_ignore + 32u;
        gm.bad_back = (uint32_t)__popcll(__ballot(badBack != 0));
        gm.empty_match = (uint32_t)__popcll(__ballot(line && panicPrev));
        if (lane == 0) A.gmeta[g] = gm;
        if (A.spin_ticks) {
            const unsigned long long t0_ = wall_clock64();
            while (wall_clock64() - t0_ < A.spin_ticks) __builtin_amdgcn_s_sleep(8);
        }
        CF_P5(6)
#ifdef CF_P5PROF
        if (A.dbg && lane == 0)
            for (int i = 0; i < 8; i++) A.dbg[16 * (size_t)g + 8 + i] = p5T[i];
#endif
        CF_TICK(5);
    }
#undef CF_NOFIT
}
};

// Device consensus of the round whose chaining stage (dp_find_overlaps) last ran on this context.
int dp_consensus_paf_impl(dp_ctx* ctx, const dp_seq_meta* metas, uint32_t n_seqs, const int32_t* rc_of, uint32_t n_seeds, int k,
                          int overlap_size, dp_paf_batch* out) {
    memset(out, 0, sizeof(*out));
    hipSetDevice(ctx->device);
    // a chaining stage left pending (dp_find_overlaps, want_candidates bit 2) is evaluated in this call's wait: its pair count is
    // not known yet, so the output is sized from a bound - twice the previous round's count - and the kernel reports a slot
    // beyond it; then, or when the stage itself had to be run again with larger buffers, this call is simply made once more
    const bool pending = dp_find_pending(ctx);
    const uint32_t nq = ctx->last_nq, ng = nq / 2;
    uint32_t np = ctx->n_pairs;
    if (pending) np = std::min<uint32_t>(dp_find_pair_cap(ctx), std::max<uint32_t>(4096u, 2 * ctx->cons_prev_pairs));
    out->n_groups = ng;
    if ((metas && n_seqs != ctx->n_seqs) || n_seeds != ctx->n_seeds) return dp_fail(ctx, DP_ERR_ARG, "dp_consensus_paf: metas / rc_of do not match the round's index");
    if (!metas && !ctx->chunks_on_device) return dp_fail(ctx, DP_ERR_ARG, "dp_consensus_paf: metas == NULL needs an index made by dp_index_build_chunked");
    if (!metas) n_seqs = 0;  // (the chunks' fields are on the device already: only rc_of travels)
    if (nq & 1) return dp_fail(ctx, DP_ERR_ARG, "dp_consensus_paf: queries must come in (forward, reverse complement) pairs");
    if (ng == 0 || (!ctx->find_valid && !pending)) {
        if (!ctx->find_valid && ng) return dp_fail(ctx, DP_ERR_STATE, "dp_consensus_paf before dp_find_overlaps");
        return DP_OK;
    }
    const size_t b_meta = (size_t)n_seqs * sizeof(dp_seq_meta), b_rc = (size_t)n_seeds * 4;
    if (pin_reserve(ctx, ctx->h_cin, b_meta + b_rc + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_cin, b_meta + b_rc + 64)) return DP_ERR_HIP;
    if (metas) memcpy(ctx->h_cin.p, metas, b_meta);
    memcpy((uint8_t*)ctx->h_cin.p + b_meta, rc_of, b_rc);
    const dp_fetch_region cin_fetch = {ctx->d_cin.p, ctx->h_cin.p, b_meta + b_rc};  // (brought over by the anchors launch below)
    const size_t b_paf = (size_t)np * sizeof(dp_paf_rec), b_ign = (size_t)np * 4, b_gm = ((size_t)ng * sizeof(dp_group_meta) + 15) & ~(size_t)15;
    // The kernel's output - group records, PAF records, ignore ids: written once, read by nobody on the device - goes straight
    // into the pinned host block (the kernel's stores cross the link; the wait below is the stream's), and the chunk count of
    // dp_index_build_chunked rides along: two copies per round that never were submitted (HISTORY.md 5.3: a copy costs more
    // than its bytes)
    if (pin_reserve(ctx, ctx->h_cout, b_paf + b_ign + b_gm + 96)) return DP_ERR_HIP;
    uint8_t* dout = (uint8_t*)ctx->h_cout.p;
    ConsFullArgs A;
    A.recs = (const uint32_t*)ctx->d_mrec.p;
    A.ma = (const int32_t*)dp_chain_a(ctx);
    A.mb = (const int32_t*)dp_chain_b(ctx);
    A.pbase = (const uint32_t*)((const uint64_t*)ctx->d_pbase.p + nq + 1);
    A.qsegs = ctx->qsegs_dev;
    A.qoff = (const uint64_t*)ctx->qoff_dev;
    A.n_groups = ng;
    A.refs = (const dp_seq_ref*)ctx->d_seqrefs.p;
    A.segs = (const int32_t*)ctx->d_segs.p;
    A.smeta = metas ? (const dp_seq_meta*)ctx->d_cin.p : (const dp_seq_meta*)ctx->d_chunk_meta.p;
    {
        static const uint32_t flag_every = (uint32_t)dp_tune("cons_flag_every", 0);  // (test hook: every n-th window goes to the host path)
        A.flag_every = flag_every;
        static const uint32_t spin = 0u;
        A.spin_ticks = spin;
    }
    A.rc_of = (const int32_t*)((const uint8_t*)ctx->d_cin.p + b_meta);
    // small LDS layout first (int16: needs every seed id below 2^15), the large one for what it lists; DP_CONS_LAYOUTS=nosmall: large only.
    // The huge layout follows the large one from the round after the first in which a window did not fit the large one
    // (ctx->cons_huge; DP_CONS_LAYOUTS=huge / nohuge: always / never) - the sparse regime never pays its launch.
    // DP_CONS_LAYOUTS (tests; read per call: they switch it between jobs of one process): "nosmall" = the large layout for every window,
    // "eager" = the large layout launched behind the small one in every round, "huge" / "nohuge" = the huge layout always / never
    const char* lay_env = getenv("DP_CONS_LAYOUTS");
    const std::string lay = lay_env ? lay_env : "";
    const bool small_off = lay.find("nosmall") != std::string::npos;
    const bool use_small = !small_off && n_seeds <= 32767;
    const bool use_huge = lay.find("nohuge") != std::string::npos ? false : lay.find("huge") != std::string::npos ? true : ctx->cons_huge;
    // lists: [count of list 1 | count of list 2 | list 1: ng entries | list 2: ng entries] (both counts zeroed by the anchors launch)
    uint32_t* lists = nullptr;
    if (use_small || use_huge) {
        if (dev_reserve(ctx, ctx->d_cretry, (2 * (size_t)ng + 2) * 4 + 16)) return DP_ERR_HIP;
        lists = (uint32_t*)ctx->d_cretry.p;
    }
    {
        int rc = dp_match_anchors_launch(ctx, &cin_fetch, lists);
        if (rc != 0) return rc;
    }
    A.anchors = (const int32_t*)ctx->d_manchor.p;
    A.cover = A.anchors + ctx->mcover_off;
    A.read_len = (const uint32_t*)ctx->d_len.p;
    A.k = k;
    A.overlap_size = overlap_size;
    A.gmeta = (dp_group_meta*)dout;
    A.paf = (dp_paf_rec*)(dout + b_gm);
    A.ignore_ids = (uint32_t*)(dout + b_gm + b_paf);
    A.out_cap = np;
    A.rec_cap = pending ? dp_find_pair_cap(ctx) : ctx->n_pairs;
    static const bool cons_debug = dp_debug("cons");
    A.dbg = nullptr;
    if (cons_debug) {
        DP_HIP(dp_dev_malloc((void**)&A.dbg, (size_t)ng * 128));
        DP_HIP(hipMemsetAsync(A.dbg, 0, (size_t)ng * 128, ctx->stream));
    }
    uint32_t* h_nseq = (uint32_t*)((uint8_t*)ctx->h_cout.p + ((b_paf + b_ign + b_gm + 15) & ~(size_t)15));
    h_nseq[0] = ctx->n_seqs;
    h_nseq[1] = 0;
    A.nseq_src = ctx->chunks_on_device ? (const uint32_t*)ctx->d_nseqs.p : nullptr;
    A.nseq_dst = h_nseq;
    bool lazy_large = false;
    const bool lazy_off = lay.find("eager") != std::string::npos;
    DP_HIP(dp_mark(ctx, 0));
    {
        // small -> large -> huge: each layout works through what its predecessor listed and lists what it cannot hold for its
        // successor; the last one flags for the host path.  Only the first launch does every window.
        const uint32_t *in_count = nullptr, *in_list = nullptr;
        uint32_t* next_count = lists;
        uint32_t* next_list = lists ? lists + 2 : nullptr;
        auto stage = [&](bool last) {
            A.in_count = in_count;
            A.in_list = in_list;
            A.copy_nseq = in_list == nullptr;
            A.out_count = last ? nullptr : next_count;
            A.out_list = last ? nullptr : next_list;
            if (!last) {
                in_count = next_count;
                in_list = next_list;
                next_count = lists + 1;
                next_list = lists + 2 + ng;
            }
        };
        if (use_small) {
            stage(false);
            dp_launch<consensus_full_kernel<0>>(ctx, dim3(std::min<uint32_t>(ng, 4096)), dim3(64), A);
        }
        // The large layout behind the small one nearly always finds its list empty (config 2: a window in a few hundred rounds) and
        // still was a launch of 4 us alone, 12 us under five slots, in every round: it is launched at once only while the context's
        // recent rounds needed it (cons_large_rounds), otherwise after the wait - if the small layout's records say "listed" (flag 3)
        lazy_large = use_small && !use_huge && ctx->cons_large_rounds == 0 && !lazy_off;
        if (!lazy_large) {
            stage(!use_huge);
            dp_launch<consensus_full_kernel<1>>(ctx, dim3(std::min<uint32_t>(ng, use_small ? 96u : 4096u)), dim3(64), A);
            if (use_huge) {
                stage(true);
                // (144 KB of LDS per wave: one per CU; behind the small + large layouts it mostly finds its list short or empty)
                dp_launch<consensus_full_kernel<2>>(ctx, dim3(std::min<uint32_t>(ng, use_small ? 64u : 256u)), dim3(64), A);
            }
        }
    }
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 1));
    DP_HIP(dp_stream_sync(ctx));
    if (use_small) {
        // windows the small layout listed: with the large layout not launched yet, it runs now (and the next rounds launch it at once)
        const dp_group_meta* gms = (const dp_group_meta*)ctx->h_cout.p;
        bool listed = false;
        for (uint32_t g = 0; g < ng && !listed; g++) listed = gms[g].flag == 3;
        if (listed && lazy_large) {
            A.in_count = lists;
            A.in_list = lists + 2;
            A.copy_nseq = false;
            A.out_count = nullptr;
            A.out_list = nullptr;
            dp_launch<consensus_full_kernel<1>>(ctx, dim3(std::min<uint32_t>(ng, 96u)), dim3(64), A);
            DP_HIP(hipGetLastError());
            DP_HIP(dp_stream_sync(ctx));
        }
        if (listed) ctx->cons_large_rounds = 64;
        else if (ctx->cons_large_rounds > 0) ctx->cons_large_rounds--;
    }
    if (h_nseq[1]) return dp_fail(ctx, DP_ERR_CAPACITY, "dp_index_build_chunked: chunk bound exceeded");
    if (pending) {
        bool reran = false;
        if (int rc = dp_find_complete(ctx, &reran)) return rc;
        const bool overflow = ctx->n_pairs > np;  // (some window's slots lay beyond the bound: its group carries flag 2)
        ctx->cons_prev_pairs = ctx->n_pairs;
        if (reran || overflow) return dp_consensus_paf_impl(ctx, metas, metas ? ctx->n_seqs : 0, rc_of, n_seeds, k, overlap_size, out);
    } else {
        ctx->cons_prev_pairs = ctx->n_pairs;
    }
    dp_find_stats(ctx, &out->query_kernel_ms, &out->chain_kernel_ms, &out->query_bytes, &out->chain_bytes);
    out->index_kernel_ms = ctx->index_marked ? (double)dp_elapsed(ctx, 8, 9) : 0.0;
    ctx->index_marked = false;
    out->n_indexed = h_nseq[0];
    float ms = 0;
    ms = dp_elapsed(ctx, 0, 1);
    out->kernel_ms = ms;
    if (cons_debug) {  // per-phase time of the groups that ran to the end: mean and maximum, in microseconds
        std::vector<unsigned long long> h((size_t)ng * 16);
        hipMemcpy(h.data(), A.dbg, (size_t)ng * 128, hipMemcpyDeviceToHost);
        dp_dev_free(A.dbg);
        double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}, tot = 0, totmx = 0;
        uint32_t cnt = 0;
        for (uint32_t g = 0; g < ng; g++) {
            if (!h[16 * (size_t)g + 5]) continue;
            cnt++;
            for (int i = 0; i < 5; i++) {
                const double d = (double)(h[16 * (size_t)g + i + 1] - h[16 * (size_t)g + i]) / 100.0;
                sum[i] += d;
                mx[i] = std::max(mx[i], d);
            }
            const double t = (double)(h[16 * (size_t)g + 5] - h[16 * (size_t)g]) / 100.0;
            tot += t;
            totmx = std::max(totmx, t);
        }
        {
            double su = 0, sg = 0, sp = 0, sn = 0;
            uint32_t slow = 0;
            double slowT = 0;
            for (uint32_t g = 0; g < ng; g++) {
                if (!h[16 * (size_t)g + 5]) continue;
                su += (double)(h[16 * (size_t)g + 6] >> 32);
                sg += (double)(h[16 * (size_t)g + 6] & 0xffffffffu);
                sp += (double)(h[16 * (size_t)g + 7] & 0xffffffffu);
                sn += (double)(h[16 * (size_t)g + 7] >> 32);
                const double t = (double)(h[16 * (size_t)g + 4] - h[16 * (size_t)g + 3]);
                if (t > slowT) slowT = t, slow = g;
            }
            {
                uint32_t sg2 = 0;
                double st2 = 0;
                for (uint32_t g = 0; g < ng; g++) {
                    if (!h[16 * (size_t)g + 5]) continue;
                    const double t = (double)(h[16 * (size_t)g + 5] - h[16 * (size_t)g]);
                    if (t > st2) st2 = t, sg2 = g;
                }
                if (cnt)
                    fprintf(stderr, "[cons] slowest window (%.1f us): gather+query %.1f trim %.1f shared+reduce %.1f align %.1f contig+paf %.1f, %llu sequences\n", st2 / 100.0,
                            (h[16 * (size_t)sg2 + 1] - h[16 * (size_t)sg2]) / 100.0, (h[16 * (size_t)sg2 + 2] - h[16 * (size_t)sg2 + 1]) / 100.0,
                            (h[16 * (size_t)sg2 + 3] - h[16 * (size_t)sg2 + 2]) / 100.0, (h[16 * (size_t)sg2 + 4] - h[16 * (size_t)sg2 + 3]) / 100.0,
                            (h[16 * (size_t)sg2 + 5] - h[16 * (size_t)sg2 + 4]) / 100.0, h[16 * (size_t)sg2 + 7] >> 32);
            }
            if (cnt)
                fprintf(stderr, "[cons] steps per window: %.1f uniform, %.1f general (%.1f proposer searches), %.1f sequences | slowest align %.1f us: %llu uniform, %llu general, %llu searches, %llu sequences\n",
                        su / cnt, sg / cnt, sp / cnt, sn / cnt, slowT / 100.0, h[16 * (size_t)slow + 6] >> 32, h[16 * (size_t)slow + 6] & 0xffffffffu,
                        h[16 * (size_t)slow + 7] & 0xffffffffu, h[16 * (size_t)slow + 7] >> 32);
        }
        {
            double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t g = 0; g < ng; g++)
                if (h[16 * (size_t)g + 5])
                    for (int i = 0; i < 8; i++) t[i] += (double)h[16 * (size_t)g + 8 + i];
            if (cnt)
                fprintf(stderr, "[cons] general steps, per window: before the scan %.2f us, scan %.2f, selection %.2f, update %.2f | scan runs %.1f, lane-trips of the scan's first walk %.1f, lanes not found in the update %.1f, steps without a seed %.1f\n",
                        t[0] / cnt / 100.0, t[1] / cnt / 100.0, t[2] / cnt / 100.0, t[3] / cnt / 100.0, t[4] / cnt, t[5] / cnt, t[6] / cnt, t[7] / cnt);
        }
#ifdef CF_P5PROF
        {
            double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t g = 0; g < ng; g++)
                if (h[16 * (size_t)g + 5])
                    for (int i = 0; i < 8; i++) t[i] += (double)h[16 * (size_t)g + 8 + i];
            if (cnt)
                fprintf(stderr, "[cons] contig+paf, us per window: leave %.2f, prefix sums + histograms %.2f, best seeds %.2f, base indices %.2f, trimmed ends + offsets %.2f, kept matches + identity %.2f, lines %.2f\n",
                        t[0] / cnt / 100.0, t[1] / cnt / 100.0, t[2] / cnt / 100.0, t[3] / cnt / 100.0, t[4] / cnt / 100.0, t[5] / cnt / 100.0, t[6] / cnt / 100.0);
        }
#endif
        if (cnt)
            fprintf(stderr, "[cons] kernel %.3f ms, %u of %u groups complete | us mean/max: gather+query %.1f/%.1f trim %.1f/%.1f shared+reduce %.1f/%.1f "
                            "align %.1f/%.1f contig+paf %.1f/%.1f | group total %.1f/%.1f\n",
                    ms, cnt, ng, sum[0] / cnt, mx[0], sum[1] / cnt, mx[1], sum[2] / cnt, mx[2], sum[3] / cnt, mx[3], sum[4] / cnt, mx[4], tot / cnt, totmx);
    }
    const uint8_t* h = (const uint8_t*)ctx->h_cout.p;
    {
        static const bool why = dp_debug("cons_why");
        if (why) {
            const dp_group_meta* gms = (const dp_group_meta*)h;
            uint32_t hist[16] = {0}, nf = 0;
            for (uint32_t g = 0; g < ng; g++)
                if (gms[g].flag == 1) hist[gms[g].reserved & 15u]++, nf++;
            if (nf)
                fprintf(stderr, "[cons] %u of %u windows left to the host (huge layout %s): capacity %u | bad list (first index %u, index %u, anchors %u, start > end %u) | > 64 sequences %u | T %u | 16 bit %u | R %u | 16 bit R %u | consensus %u | parts %u\n",
                        nf, ng, use_huge ? "on" : "off", hist[1], hist[10], hist[11], hist[12], hist[13], hist[8], hist[9], hist[3], hist[4], hist[5], hist[6], hist[7]);
        }
    }
    if (!ctx->cons_huge) {  // a window left to the host path: the next rounds of this context try the huge layout before that
        const dp_group_meta* gms = (const dp_group_meta*)h;
        for (uint32_t g = 0; g < ng; g++)
            if (gms[g].flag == 1) {
                ctx->cons_huge = true;
                break;
            }
    }
    out->groups = (const dp_group_meta*)h;
    out->paf = (const dp_paf_rec*)(h + b_gm);
    out->ignore_ids = (const uint32_t*)(h + b_gm + b_paf);
    return DP_OK;
}

extern "C" int dp_consensus_paf(dp_ctx* ctx, const dp_seq_meta* metas, uint32_t n_seqs, const int32_t* rc_of, uint32_t n_seeds, int k,
                                int overlap_size, dp_paf_batch* out) {
    if (!ctx || !out || (n_seqs && !metas) || (n_seeds && !rc_of)) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_consensus_paf: bad arguments") : DP_ERR_ARG;
    return dp_consensus_paf_impl(ctx, metas, n_seqs, rc_of, n_seeds, k, overlap_size, out);
}
