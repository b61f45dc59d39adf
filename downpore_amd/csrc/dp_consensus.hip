// libdownpore_hip.so — A16 (part): the seed-space multiple alignment at the heart of multiAligner.Consensus
// (seeds/alignment.go:23-268) for many query windows at once.  One wave per group of sequences; lane i owns sequence i
// (its cursor pos/offs/gaps, its support counters and its match list), the consensus grows one seed per iteration.
// The host keeps what surrounds it (trimming the matched targets, Reduced(), trimToBestSeed, PAF): see host_overlap.cpp.
#include <algorithm>
#include <cstring>

#include "dp_common.h"

#define CA_WAVES 4
#define CA_CAP 6144  // ints of one group staged in LDS

#define CA_RL(v_, l_) __builtin_amdgcn_readlane((v_), (l_))

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
    DP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(consensus_align_kernel, dim3((n_groups + CA_WAVES - 1) / CA_WAVES), dim3(64 * CA_WAVES), 0, ctx->stream,
                       (const int32_t*)(din + b_soff + b_coff), (const uint64_t*)din, (const uint32_t*)(din + b_soff + b_coff + b_segs),
                       n_groups, k, d_cons, (const uint64_t*)(din + b_soff), d_clen, d_ma, d_mb, d_mlen, d_flag);
    DP_HIP(hipGetLastError());
    DP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    uint8_t* hout = (uint8_t*)ctx->h_cout.p;
    DP_HIP(hipMemcpyAsync(hout, dout, out_bytes - 64, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    memcpy(hout + out_bytes, h_coff, b_coff);
    float ms = 0;
    hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
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
