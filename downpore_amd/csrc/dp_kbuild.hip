// libdownpore_hip.so — building the resident k-mer position index (dp_kindex.hip) and the k-mer histogram (A22) in one
// go, as a most-significant-digit radix sort whose every pass streams HBM with coalesced traffic.
//
// The first version scattered each of the ~10^9 k-mer positions straight to its bucket (one 8-byte store to a random
// address each: ~95 ms at config 2, 1 % of the HBM roofline) after counting them with 10^9 global atomics (37 ms).  Here
// an entry (k-mer << 36 | absolute base index) goes through three passes instead:
//   pass 1  by the k-mer's top B1 bits, straight from the packed reads;
//   pass 2  by its next B2 bits, inside every pass-1 bucket;
//   pass 3  one workgroup per (B1+B2)-bit sub-partition: the remaining R = 2k - B1 - B2 bits are counted in LDS - those
//           counts ARE the k-mer histogram of the sub-partition's 2^R k-mers - and the entries leave in final order as
//           (read << 32 | position in the read).
// Passes 1 and 2 sort a tile of 4096 entries by digit in LDS, reserve room in every digit's bucket with one atomic per
// (tile, digit), and copy runs out: consecutive lanes write consecutive addresses.  Pass 3 places a whole sub-partition in
// LDS when it fits (<= 8192 entries, the digit widths are chosen for that) and writes it out sequentially.
// Algorithmic bytes of the build: packed reads once + 8 B per k-mer start written + the 4^k histogram and offset tables;
// each pass moves 16 B per entry (read + write), which is what its own roofline fraction is computed from.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dp_common.h"

typedef unsigned long long kb_u64;

#define KB_TILE 16384       // entries per tile of passes 1/2
#define KB_THREADS 1024
#define KB_P3_CAP 8192     // entries pass 3 sorts inside LDS
// An entry in flight = k-mer << (rbits + pbits) | read << pbits | position in the read: what the index finally stores travels
// with the k-mer from pass 1 on (the read of a position is known for free there), so the last pass converts nothing.  Needs
// 2k + rbits + pbits <= 64: 26 + 17 + 14 at config 2, 26 + 21 + 14 at config 5's read set.

struct KbGeom {
    int k, b1, b2, r;  // digit widths: b1 + b2 + r = 2k
    int pbits, rbits;  // payload below the k-mer: position in the read, read id
    // round 5: a rank of a multi-GPU job builds the index of the k-mers whose pass-1 digit lies in [dlo, dhi) only (the ranks' parts are
    // all-gathered afterwards, dp_kindex.hip); one rank: [0, 2^b1)
    uint32_t dlo, dhi;
};

// Round 4: entries are stored as narrow as their bits allow.  An entry in flight is a 64-bit word in registers and LDS; between the
// passes it lives in two streams - the low 32 bits, and the bits above them in a stream of 0, 1, 2 or 4 bytes per entry (both written
// in the same coalesced runs).  After pass 1 an entry needs 2k - b1 + rbits + pbits bits (config 2: 48 - 49: six or eight bytes),
// after pass 2 r + rbits + pbits (40: five bytes), and the index itself rbits + pbits (31: four bytes; 34 at config 4: five) -
// 32 bytes moved per entry over the three passes instead of 48, and an index of half the size for the rounds to walk.
struct KbBuf {
    uint32_t* lo;
    void* hi;
    int hb;  // bytes per entry in `hi`: 0, 1, 2 or 4; 8: `lo` is one stream of 64-bit words
};
__device__ __forceinline__ kb_u64 kb_ld(const KbBuf B, kb_u64 i) {
    if (B.hb == 8) return ((const kb_u64*)B.lo)[i];
    kb_u64 v = B.lo[i];
    if (B.hb == 1) v |= (kb_u64)((const uint8_t*)B.hi)[i] << 32;
    else if (B.hb == 2) v |= (kb_u64)((const uint16_t*)B.hi)[i] << 32;
    else if (B.hb == 4) v |= (kb_u64)((const uint32_t*)B.hi)[i] << 32;
    return v;
}
__device__ __forceinline__ void kb_st(const KbBuf B, kb_u64 i, kb_u64 v) {
    if (B.hb == 8) {
        ((kb_u64*)B.lo)[i] = v;
        return;
    }
    B.lo[i] = (uint32_t)v;
    if (B.hb == 1) ((uint8_t*)B.hi)[i] = (uint8_t)(v >> 32);
    else if (B.hb == 2) ((uint16_t*)B.hi)[i] = (uint16_t)(v >> 32);
    else if (B.hb == 4) ((uint32_t*)B.hi)[i] = (uint32_t)(v >> 32);
}

// read of every 1024-base block of the packed layout (reads start on 64-base boundaries, so a 32-position group never
// straddles two reads): gread[j] = read whose [boff*4, boff_next*4) range holds base 1024*j
__global__ void kb_group_table(const uint64_t* __restrict__ boff, uint32_t n_reads, uint32_t* __restrict__ gread) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t a0 = boff[r] * 4, a1 = boff[r + 1] * 4;
    for (uint64_t j = (a0 + 1023) >> 10; (j << 10) < a1; j++) gread[j] = r;
}

__device__ __forceinline__ uint32_t kb_read_of(uint64_t a, const uint32_t* __restrict__ gread, const uint64_t* __restrict__ boff) {
    uint32_t r = gread[a >> 10];
    while (boff[r + 1] * 4 <= a) r++;
    return r;
}

// the up to 16 k-mers a thread owns in passes over the packed reads: half a 32-position group.  Returns the valid mask.
__device__ __forceinline__ uint32_t kb_load16(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                              const uint32_t* __restrict__ len, const uint32_t* __restrict__ gread, uint64_t n_groups,
                                              uint64_t g, int half, int k, uint32_t kmer[16], uint32_t* read_out = nullptr,
                                              uint64_t* start_out = nullptr) {
    if (g >= n_groups) return 0u;
    const uint64_t a = g * 32 + (uint64_t)half * 16;
    const uint32_t r = kb_read_of(g * 32, gread, boff);
    const uint32_t L = len[r];
    if (L < (uint32_t)k) return 0u;
    const uint64_t a0 = boff[r] * 4, a1 = a0 + (L - k + 1);
    if (read_out) *read_out = r;
    if (start_out) *start_out = a0;
    if (a >= a1) return 0u;
    const uint32_t* p = (const uint32_t*)(packed + g * 8) + half;
    const uint32_t w0 = __builtin_bswap32(p[0]), w1 = __builtin_bswap32(p[1]);
    const int sh = 32 - 2 * k;
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t win = j ? __builtin_amdgcn_alignbit(w0, w1, 32 - 2 * j) : w0;
        kmer[j] = win >> sh;
        if (a + j >= a0 && a + j < a1) mask |= 1u << j;
    }
    return mask;
}

// ---- pass 1 --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(KB_THREADS) void kb_count1(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                                        const uint32_t* __restrict__ len, const uint32_t* __restrict__ gread,
                                                        uint64_t n_groups, KbGeom G, kb_u64* __restrict__ cnt1) {
    __shared__ uint32_t hist[1024];
    const int nb = 1 << G.b1;
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) hist[i] = 0;
    __syncthreads();
    const int dsh = 2 * G.k - G.b1;
    const uint64_t tiles = (n_groups * 2 + KB_THREADS - 1) / KB_THREADS;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t h = t * KB_THREADS + threadIdx.x;
        uint32_t kmer[16];
        const uint32_t m = kb_load16(packed, boff, len, gread, n_groups, h >> 1, (int)(h & 1), G.k, kmer);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((m >> j) & 1) atomicAdd(&hist[kmer[j] >> dsh], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += KB_THREADS)
        if (hist[i]) atomicAdd(&cnt1[i], (kb_u64)hist[i]);
}

// shared tail of passes 1 and 2: the tile's entries sit unsorted in registers `e[]` with validity mask `m`; sort by digit in
// LDS, reserve room per digit with one atomic, copy runs out
template <int E>
__device__ __forceinline__ void kb_tile_out(const kb_u64 (&e)[E], uint32_t m, int dshift, uint32_t dmask, uint32_t* hist /*[1024]*/,
                                            uint32_t* lstart /*[1024]*/, kb_u64* gbase /*[1024]*/, kb_u64* sorted /*[KB_TILE]*/,
                                            kb_u64* __restrict__ cursor, const KbBuf out) {
    const int nb = (int)dmask + 1;
    for (int i = threadIdx.x; i < nb; i += KB_THREADS) hist[i] = 0;
    __syncthreads();
    // (the count's own atomic hands every entry its rank inside its digit - kept in registers, so the placing below is a plain LDS
    // store: one LDS atomic per entry instead of two, round 5)
    uint32_t rk[E];
#pragma unroll
    for (int j = 0; j < E; j++) rk[j] = ((m >> j) & 1) ? atomicAdd(&hist[(uint32_t)(e[j] >> dshift) & dmask], 1u) : 0u;
    __syncthreads();
    // exclusive scan of the digit counts (nb <= 1024: four bins per thread), global reservation, cursors back to zero
    {
        const int per = (nb + KB_THREADS - 1) / KB_THREADS;
        uint32_t loc[4] = {0, 0, 0, 0}, s = 0;
        for (int u = 0; u < per; u++) {
            const int b = threadIdx.x * per + u;
            loc[u] = b < nb ? hist[b] : 0u;
            s += loc[u];
        }
        const int lane = dp_lane(), wave = threadIdx.x >> 6;
        __shared__ uint32_t wsum[KB_THREADS / 64];
        uint32_t x = (uint32_t)wave_incl_sum((int)s);
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t base = x - s;
        for (int w = 0; w < wave; w++) base += wsum[w];
        for (int u = 0; u < per; u++) {
            const int b = threadIdx.x * per + u;
            if (b < nb) {
                lstart[b] = base;
                if (loc[u]) gbase[b] = atomicAdd(&cursor[b], (kb_u64)loc[u]);
                base += loc[u];
            }
        }
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < E; j++)
        if ((m >> j) & 1) {
            const uint32_t d = (uint32_t)(e[j] >> dshift) & dmask;
            sorted[lstart[d] + rk[j]] = e[j];
        }
    __syncthreads();
    total = lstart[nb - 1] + hist[nb - 1];
    for (uint32_t i = threadIdx.x; i < total; i += KB_THREADS) {
        const kb_u64 v = sorted[i];
        const uint32_t d = (uint32_t)(v >> dshift) & dmask;
        kb_st(out, gbase[d] + (kb_u64)(i - lstart[d]), v);
    }
    __syncthreads();
}

__global__ __launch_bounds__(KB_THREADS) void kb_part1(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                                       const uint32_t* __restrict__ len, const uint32_t* __restrict__ gread,
                                                       uint64_t n_groups, KbGeom G, kb_u64* __restrict__ cursor, const KbBuf out) {
    __shared__ uint32_t hist[1024], lstart[1024];
    __shared__ kb_u64 gbase[1024];
    __shared__ kb_u64 sorted[KB_TILE];
    const int pay = G.pbits + G.rbits;
    const int dsh = pay + 2 * G.k - G.b1;
    const uint32_t dmask = (1u << G.b1) - 1u;
    const uint64_t tiles = (n_groups * 2 + KB_THREADS - 1) / KB_THREADS;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t h = t * KB_THREADS + threadIdx.x;
        uint32_t kmer[16], rd = 0;
        uint64_t a0 = 0;
        const uint32_t m = kb_load16(packed, boff, len, gread, n_groups, h >> 1, (int)(h & 1), G.k, kmer, &rd, &a0);
        kb_u64 e[16];
        const kb_u64 a = (h >> 1) * 32 + (h & 1) * 16;
        // (lanes outside the read's k-mer range are masked out: their payload may be anything)
        const kb_u64 low = ((kb_u64)rd << G.pbits) + (a - a0);
        uint32_t mine = m;
        const int ksh = 2 * G.k - G.b1;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            e[j] = ((kb_u64)kmer[j] << pay) | (low + (kb_u64)j);
            const uint32_t d = kmer[j] >> ksh;
            if (d < G.dlo || d >= G.dhi) mine &= ~(1u << j);  // (another rank's k-mer)
        }
        kb_tile_out<16>(e, mine, dsh, dmask, hist, lstart, gbase, sorted, cursor, out);
    }
}

// ---- pass 2 (inside every pass-1 bucket) ---------------------------------------------------------------------------
// tile_start[b] = first tile of bucket b (tiles of KB_TILE entries; buckets do not share tiles), [nb1] = total
__device__ __forceinline__ uint32_t kb_bucket_of_tile(const uint32_t* __restrict__ tile_start, int nb1, uint32_t t) {
    int lo = 0, hi = nb1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tile_start[mid] <= t) lo = mid;
        else hi = mid;
    }
    return (uint32_t)lo;
}

__global__ __launch_bounds__(KB_THREADS) void kb_count2(const KbBuf in, const kb_u64* __restrict__ base1,
                                                        const uint32_t* __restrict__ tile_start, KbGeom G, kb_u64* __restrict__ cnt2) {
    __shared__ uint32_t hist[1024];
    const int nb1 = 1 << G.b1, nb2 = 1 << G.b2;
    const uint32_t n_tiles = tile_start[nb1];
    const int dsh = G.pbits + G.rbits + 2 * G.k - G.b1 - G.b2;
    const uint32_t dmask = (uint32_t)nb2 - 1u;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t b = kb_bucket_of_tile(tile_start, nb1, t);
        const kb_u64 lo = base1[b] + (kb_u64)(t - tile_start[b]) * KB_TILE, hi = min(base1[b + 1], lo + KB_TILE);
        for (int i = threadIdx.x; i < nb2; i += KB_THREADS) hist[i] = 0;
        __syncthreads();
        for (kb_u64 i = lo + threadIdx.x; i < hi; i += KB_THREADS) atomicAdd(&hist[(uint32_t)(kb_ld(in, i) >> dsh) & dmask], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < nb2; i += KB_THREADS)
            if (hist[i]) atomicAdd(&cnt2[(kb_u64)b * nb2 + i], (kb_u64)hist[i]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(KB_THREADS) void kb_part2(const KbBuf in, const kb_u64* __restrict__ base1,
                                                       const uint32_t* __restrict__ tile_start, KbGeom G, kb_u64* __restrict__ cursor2,
                                                       const KbBuf out) {
    __shared__ uint32_t hist[1024], lstart[1024];
    __shared__ kb_u64 gbase[1024];
    __shared__ kb_u64 sorted[KB_TILE];
    const int nb1 = 1 << G.b1, nb2 = 1 << G.b2;
    const uint32_t n_tiles = tile_start[nb1];
    const int dsh = G.pbits + G.rbits + 2 * G.k - G.b1 - G.b2;
    const uint32_t dmask = (uint32_t)nb2 - 1u;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t b = kb_bucket_of_tile(tile_start, nb1, t);
        const kb_u64 lo = base1[b] + (kb_u64)(t - tile_start[b]) * KB_TILE, hi = min(base1[b + 1], lo + KB_TILE);
        kb_u64 e[KB_TILE / KB_THREADS];
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < KB_TILE / KB_THREADS; j++) {
            const kb_u64 i = lo + (kb_u64)j * KB_THREADS + threadIdx.x;  // coalesced
            e[j] = 0;
            if (i < hi) {
                e[j] = kb_ld(in, i);
                m |= 1u << j;
            }
        }
        kb_tile_out<KB_TILE / KB_THREADS>(e, m, dsh, dmask, hist, lstart, gbase, sorted, cursor2 + (kb_u64)b * nb2, out);
    }
}

// ---- pass 3: one workgroup per sub-partition ------------------------------------------------------------------------
// counts[kmer] (the histogram), off[kmer] (bucket starts of the index) and the index entries in final order
// The index entry: fmt 4 = read << pbits | position in 32 bits; 5 = the same in 40 (low 32 in `lo`, the rest in a byte stream);
// 8 = read << 32 | position in 64 (what the atomic scatter build of dp_kindex.hip writes too)
struct KbPosOut {
    void* lo;
    uint8_t* hi;
    int fmt;
};
__device__ __forceinline__ void kb_pos_st(const KbPosOut P, kb_u64 i, kb_u64 v, const KbGeom& G) {
    const kb_u64 pay = v & (((kb_u64)1 << (G.pbits + G.rbits)) - 1);
    if (P.fmt == 8) {
        ((kb_u64*)P.lo)[i] = ((pay >> G.pbits) << 32) | (pay & (((kb_u64)1 << G.pbits) - 1));
    } else {
        ((uint32_t*)P.lo)[i] = (uint32_t)pay;
        if (P.fmt == 5) P.hi[i] = (uint8_t)(pay >> 32);
    }
}
__global__ __launch_bounds__(KB_THREADS) void kb_final(const KbBuf in, const kb_u64* __restrict__ base2, uint32_t n_sub,
                                                       KbGeom G, const uint32_t* __restrict__ gread, const uint64_t* __restrict__ boff,
                                                       uint32_t* __restrict__ counts, kb_u64* __restrict__ off, const KbPosOut pos) {
    extern __shared__ kb_u64 kb_dyn[];  // sorted[KB_P3_CAP] | hist[2^r] | lstart[2^r]
    const int nb = 1 << G.r;
    kb_u64* sorted = kb_dyn;
    uint32_t* hist = (uint32_t*)(kb_dyn + KB_P3_CAP);
    uint32_t* lstart = hist + nb;
    const uint32_t dmask = (uint32_t)nb - 1u;
    const int pay = G.pbits + G.rbits;
    for (uint32_t sp = blockIdx.x; sp < n_sub; sp += gridDim.x) {
        const kb_u64 lo = base2[sp], hi = base2[sp + 1];
        const kb_u64 n = hi - lo;
        for (int i = threadIdx.x; i < nb; i += KB_THREADS) hist[i] = 0;
        __syncthreads();
        // (a sub-partition that fits the LDS buffer - the digit widths are chosen so that nearly all do - is read once: its entries
        // stay in registers between the counting and the placing walk)
        constexpr int KB_HOLD = KB_P3_CAP / KB_THREADS;
        kb_u64 held[KB_HOLD];
        uint32_t rk[KB_HOLD];  // (rank of the entry inside its k-mer, from the counting atomic itself: placing needs no second one)
        const bool hold = n <= KB_P3_CAP;
        if (hold) {
#pragma unroll
            for (int u = 0; u < KB_HOLD; u++) {
                const kb_u64 i = lo + (kb_u64)u * KB_THREADS + threadIdx.x;
                held[u] = i < hi ? kb_ld(in, i) : 0;
                rk[u] = i < hi ? atomicAdd(&hist[(uint32_t)(held[u] >> pay) & dmask], 1u) : 0u;
            }
        } else {
            for (kb_u64 i = lo + threadIdx.x; i < hi; i += KB_THREADS) atomicAdd(&hist[(uint32_t)(kb_ld(in, i) >> pay) & dmask], 1u);
        }
        __syncthreads();
        // histogram + offsets of this sub-partition's 2^r k-mers (k-mer = sp << r | bin): coalesced
        {
            const int per = (nb + KB_THREADS - 1) / KB_THREADS;  // <= 16
            uint32_t s = 0;
            for (int u = 0; u < per; u++) {
                const int b = threadIdx.x * per + u;
                if (b < nb) s += hist[b];
            }
            const int lane = dp_lane(), wave = threadIdx.x >> 6;
            __shared__ uint32_t wsum[KB_THREADS / 64];
            uint32_t x = (uint32_t)wave_incl_sum((int)s);
            if (lane == 63) wsum[wave] = x;
            __syncthreads();
            uint32_t base = x - s;
            for (int w = 0; w < wave; w++) base += wsum[w];
            for (int u = 0; u < per; u++) {
                const int b = threadIdx.x * per + u;
                if (b < nb) {
                    lstart[b] = base;
                    base += hist[b];
                }
            }
        }
        __syncthreads();
        const kb_u64 kbase = (kb_u64)sp << G.r;
        for (int b = threadIdx.x; b < nb; b += KB_THREADS) {
            counts[kbase + b] = hist[b];
            off[kbase + b] = lo + lstart[b];
        }
        if (hold) {
#pragma unroll
            for (int u = 0; u < KB_HOLD; u++) {
                const kb_u64 i = lo + (kb_u64)u * KB_THREADS + threadIdx.x;
                if (i < hi) {
                    const kb_u64 v = held[u];
                    const uint32_t d = (uint32_t)(v >> pay) & dmask;
                    sorted[lstart[d] + rk[u]] = v;
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < (uint32_t)n; i += KB_THREADS) {  // consecutive lanes, consecutive addresses
                kb_pos_st(pos, lo + i, sorted[i], G);
            }
        } else {  // larger than the LDS buffer: entries go straight to their slot (the region is this workgroup's alone)
            __syncthreads();
            for (int i = threadIdx.x; i < nb; i += KB_THREADS) hist[i] = 0;  // placement cursors
            __syncthreads();
            for (kb_u64 i = lo + threadIdx.x; i < hi; i += KB_THREADS) {
                const kb_u64 v = kb_ld(in, i);
                const uint32_t d = (uint32_t)(v >> pay) & dmask;
                kb_pos_st(pos, lo + lstart[d] + atomicAdd(&hist[d], 1u), v, G);
            }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[(kb_u64)n_sub << G.r] = base2[n_sub];
}

// a rank's share of the pass-1 buckets: the other ranks' counts go to zero before the scan (their tiles, sub-partitions and k-mers then
// are empty on this rank and cost nothing in the later passes)
__global__ void kb_mask_digits(kb_u64* __restrict__ cnt1, uint32_t nb1, uint32_t dlo, uint32_t dhi) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < nb1 && (d < dlo || d >= dhi)) cnt1[d] = 0;
}

__global__ void kb_excl_scan_small(const kb_u64* __restrict__ in, uint32_t n, kb_u64* __restrict__ out) {
    // n <= 2^20 entries, one workgroup: each thread scans a contiguous slice, sixteen entries - one cache line - at a time (with one
    // entry per step the 1 024 slices' lines did not survive in the L1 until their next entry was asked for: 0.25 ms for 2 MB)
    __shared__ kb_u64 part[1024];
    const uint32_t per = (((n + 1023) / 1024) + 15u) & ~15u;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    kb_u64 s = 0;
    for (uint32_t i = lo; i < hi; i += 16) {
        kb_u64 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = i + j < hi ? in[i + j] : 0;
#pragma unroll
        for (int j = 0; j < 16; j++) s += v[j];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // (ten steps instead of one thread's walk over 1 024 sums)
        const kb_u64 v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    kb_u64 run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; i += 16) {
        kb_u64 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = i + j < hi ? in[i + j] : 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (i + j < hi) out[i + j] = run;
            run += v[j];
        }
    }
    if (threadIdx.x == 1023) out[n] = run;
}

// The scan of the sub-partition counts (2^18 of them at config 2) on many workgroups: one workgroup needed 0.23 - 0.25 ms for 2 MB in and
// 2 MB out (a single CU's share of the memory system), three small launches need a tenth of it.  partial[g] = sum of chunk g.
__global__ __launch_bounds__(1024) void kb_scan_partials(const kb_u64* __restrict__ in, uint32_t n, uint32_t chunk, kb_u64* __restrict__ partial) {
    __shared__ kb_u64 red[16];
    const uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    kb_u64 s = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) s += in[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (dp_lane() == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        kb_u64 t = 0;
        for (int w = 0; w < 16; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// out[i] = pbase[chunk of i] + sum of the chunk's entries in front of i; out[n] = the total
__global__ __launch_bounds__(1024) void kb_scan_write(const kb_u64* __restrict__ in, uint32_t n, uint32_t chunk, const kb_u64* __restrict__ pbase,
                                                      kb_u64* __restrict__ out) {
    __shared__ kb_u64 part[1024];
    const uint32_t lo = min(n, blockIdx.x * chunk), hi = min(n, lo + chunk);
    kb_u64 run = pbase[blockIdx.x];
    for (uint32_t b0 = lo; b0 < hi; b0 += 1024) {
        const uint32_t i = b0 + threadIdx.x;
        const kb_u64 v = i < hi ? in[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const kb_u64 u = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        if (i < hi) out[i] = run + part[threadIdx.x] - v;
        run += part[1023];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = run;
}

// Builds counts (uint32 [4^k]), off (uint64 [4^k + 1]) and the index entries for the context's resident reads.  d_counts, d_off
// are the caller's; *d_pos_out (and *d_pos_hi_out for format 5) are allocated here (ownership passes to the caller), the scratch is
// released before returning.  *fmt_out / *pbits_out say how an entry is stored (KbPosOut).
// Returns 1 when this path does not apply (k < 9 or > 14, more than 2^36 bases): the caller uses the atomic scatter.
int dp_kindex_build_sorted(dp_ctx* ctx, dp_ctx* ow, int k, uint32_t* d_counts, uint64_t* d_off, void** d_pos_out, void** d_pos_hi_out,
                           int* fmt_out, int* pbits_out, uint64_t* n_pos_out, float* ms_out, dp_kindex_shard* shard) {
    if (k < 9 || k > 14 || ow->n_reads == 0) return 1;
    if (dp_tune("kindex_atomic", 0)) return 1;  // (tests: the count -> offsets -> atomic scatter build of dp_kindex_ensure)
    const uint64_t n_groups = (ow->packed_bytes * 4 + 31) / 32;
    KbGeom G;
    G.k = k;
    {
        uint32_t max_len = 1;
        for (uint32_t r = 0; r < ow->n_reads; r++) max_len = std::max(max_len, ow->h_len[r]);
        G.pbits = 1;
        while (((uint64_t)1 << G.pbits) <= max_len) G.pbits++;
        G.rbits = 1;
        while (((uint64_t)1 << G.rbits) < ow->n_reads) G.rbits++;
        if (const char* e = getenv("DP_KB_MIN_PBITS")) G.pbits = std::max(G.pbits, std::min(32, atoi(e)));  // (test hook: small inputs reach the wider entry formats)
        if (2 * k + G.pbits + G.rbits > 64) return 1;  // (the entry has no room for read and position: atomic scatter build)
    }
    const int pay = G.pbits + G.rbits;
    // between the passes the entries travel as one stream of 64-bit words (two narrower streams - 4 + 1 or 2 bytes - move fewer bytes
    // and were SLOWER on MI355X: two memory instructions per lane for less than one 8-byte access moves; profiles/r04/kbuild_kernels.txt;
    // removed in round 6).  Pass 1's width: 8 bits (DP_TUNE=kb_b1=n overrides: tests)
    G.b1 = 8;
    G.b1 = (int)std::max(6L, std::min(10L, dp_tune("kb_b1", G.b1)));
    if (G.b1 > 2 * k - 2) G.b1 = 2 * k - 2;
    // pass 2's width: sub-partitions of about 6000 entries (they are sorted inside LDS when they hold <= 8192)
    const uint64_t approx = ow->total_bases;
    int b2 = std::max(1, 14 - G.b1);
    while (b2 < 10 && (approx >> (G.b1 + b2)) > 7800) b2++;
    if (G.b1 + b2 > 2 * k) b2 = 2 * k - G.b1;
    G.b2 = b2;
    G.r = 2 * k - G.b1 - G.b2;
    if (G.r > 12) {  // (k = 14 with a small read set: widen pass 2)
        G.b2 += G.r - 12;
        G.r = 12;
    }
    if (G.b2 > 10) return 1;
    const int nb1 = 1 << G.b1, nb2 = 1 << G.b2;
    G.dlo = 0;
    G.dhi = (uint32_t)nb1;
    const uint32_t n_sub = (uint32_t)nb1 * (uint32_t)nb2;
    // how the entries are stored after each pass (DP_KINDEX_WIDE=1: eight bytes throughout, the index as read << 32 | position)
    const bool wide = getenv("DP_KINDEX_WIDE") != nullptr;
    const int hb1 = 8, hb2 = 8;  // (bytes per entry of the two intermediate arrays)
    (void)wide;
    const int fmt = (wide || pay > 40) ? 8 : pay > 32 ? 5 : 4;
    // The index is written over A's low stream when that stream has the index's entry size (A is dead by pass 3): both eight bytes
    // wide (format 8) or both four (two narrow streams, formats 4 / 5).  With the default 8-byte A and a 4- or 5-byte index the index
    // gets a buffer of its own, live next to A and B.
    const bool a_lo_wide = hb1 == 8 || fmt == 8;
    const bool f_over_a = a_lo_wide == (fmt == 8);
    {
        // what is live at the build's peak (pass 3), per index entry: A + B + the index where it has a buffer of its own + format 5's
        // byte stream.  Default path at config 2: 8 + 8 + 4 = 20 B, at config 4 (format 5): 21 B = 210 GB for 10 G entries.
        const uint64_t per_entry = (uint64_t)((a_lo_wide ? 8 : 4) + (hb1 != 8 ? hb1 : 0)) + (uint64_t)(hb2 == 8 ? 8 : 4 + hb2) + (f_over_a ? 0u : 4u) + (fmt == 5 ? 1u : 0u);
        size_t free_b = 0, total_b = 0;
        hipMemGetInfo(&free_b, &total_b);
        free_b += dp_dev_cached_bytes();
        // (a rank of a sharded build sorts 1 / n_ranks of the entries; the whole index it then gathers is the caller's to fit)
        const uint64_t mine = shard && shard->n_ranks > 1 ? ow->total_bases / (uint64_t)shard->n_ranks + (ow->total_bases >> 6) : ow->total_bases;
        if ((uint64_t)free_b < mine * per_entry + ((uint64_t)6 << 30)) return 1;  // (the caller builds with the atomic scatter, or scans)
    }
    void *d_gread = nullptr, *d_small = nullptr, *d_a_lo = nullptr, *d_a_hi = nullptr, *d_b_lo = nullptr, *d_b_hi = nullptr, *d_f_hi = nullptr;
    struct Temps {
        std::vector<void**> v;
        ~Temps() {
            for (void** p : v)
                if (*p) dp_dev_free(*p);
        }
    } temps{{&d_gread, &d_small, &d_a_lo, &d_a_hi, &d_b_lo, &d_b_hi, &d_f_hi}};
    const size_t n_blocks1k = (size_t)((ow->packed_bytes * 4 + 1023) >> 10) + 2;
    DP_HIP(dp_dev_malloc(&d_gread, n_blocks1k * 4));
    // small tables: cnt1[nb1], base1[nb1+1], cur1[nb1], cnt2[n_sub], base2[n_sub+1], cur2[n_sub], tile_start[nb1+1]
    const size_t small_u64 = (size_t)nb1 * 3 + 1 + (size_t)n_sub * 3 + 1;
    DP_HIP(dp_dev_malloc(&d_small, small_u64 * 8 + ((size_t)nb1 + 1) * 4 + 64));
    kb_u64* cnt1 = (kb_u64*)d_small;
    kb_u64* base1 = cnt1 + nb1;
    kb_u64* cur1 = base1 + nb1 + 1;
    kb_u64* cnt2 = cur1 + nb1;
    kb_u64* base2 = cnt2 + n_sub;
    kb_u64* cur2 = base2 + n_sub + 1;
    uint32_t* tile_start = (uint32_t*)(cur2 + n_sub);
    DP_HIP(hipMemsetAsync(d_small, 0, small_u64 * 8 + ((size_t)nb1 + 1) * 4, ctx->stream));
    hipEvent_t e0 = ctx->ev[2], e1 = ctx->ev[3];
    DP_HIP(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(kb_group_table, dim3((ow->n_reads + 255) / 256), dim3(256), 0, ctx->stream, (const uint64_t*)ow->d_boff.p, ow->n_reads,
                       (uint32_t*)d_gread);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t grid = (uint32_t)cus * 8;
    hipLaunchKernelGGL(kb_count1, dim3(grid), dim3(KB_THREADS), 0, ctx->stream, (const uint8_t*)ow->d_packed.p, (const uint64_t*)ow->d_boff.p,
                       (const uint32_t*)ow->d_len.p, (const uint32_t*)d_gread, n_groups, G, cnt1);
    if (shard) {
        // every rank counts every k-mer's first digit (0.4 ms) and so knows every rank's share without asking: rank q takes the digits
        // [first[q], first[q + 1]) - consecutive pass-1 buckets holding about 1 / n_ranks of the entries - and sorts those alone
        std::vector<kb_u64> h_cnt((size_t)nb1);
        DP_HIP(hipMemcpyAsync(h_cnt.data(), cnt1, (size_t)nb1 * 8, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
        kb_u64 all = 0;
        for (int b = 0; b < nb1; b++) all += h_cnt[(size_t)b];
        const int N = shard->n_ranks;
        shard->digit_first.assign((size_t)N + 1, (uint32_t)nb1);
        shard->entry_first.assign((size_t)N + 1, all);
        std::vector<kb_u64> pre((size_t)nb1 + 1, 0);
        for (int b = 0; b < nb1; b++) pre[(size_t)b + 1] = pre[(size_t)b] + h_cnt[(size_t)b];
        shard->digit_first[0] = 0;
        shard->entry_first[0] = 0;
        for (int q = 1; q < N; q++) {
            const kb_u64 want = (all * (kb_u64)q + (kb_u64)N - 1) / (kb_u64)N;
            int b = (int)shard->digit_first[(size_t)q - 1];
            while (b < nb1 && pre[(size_t)b] < want) b++;
            shard->digit_first[(size_t)q] = (uint32_t)b;
            shard->entry_first[(size_t)q] = pre[(size_t)b];
        }
        shard->digit_first[(size_t)N] = (uint32_t)nb1;
        shard->entry_first[(size_t)N] = all;
        shard->kmer_shift = 2 * k - G.b1;
        shard->total = all;
        G.dlo = shard->digit_first[(size_t)shard->rank];
        G.dhi = shard->digit_first[(size_t)shard->rank + 1];
        hipLaunchKernelGGL(kb_mask_digits, dim3((nb1 + 255) / 256), dim3(256), 0, ctx->stream, cnt1, (uint32_t)nb1, G.dlo, G.dhi);
    }
    hipLaunchKernelGGL(kb_excl_scan_small, dim3(1), dim3(1024), 0, ctx->stream, (const kb_u64*)cnt1, (uint32_t)nb1, base1);
    DP_HIP(hipGetLastError());
    std::vector<kb_u64> h_base1((size_t)nb1 + 1);
    DP_HIP(hipMemcpyAsync(h_base1.data(), base1, ((size_t)nb1 + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    const uint64_t n = h_base1[(size_t)nb1];
    std::vector<uint32_t> h_tiles((size_t)nb1 + 1);
    uint64_t tt = 0;
    for (int b = 0; b < nb1; b++) {
        h_tiles[(size_t)b] = (uint32_t)tt;
        tt += (h_base1[(size_t)b + 1] - h_base1[(size_t)b] + KB_TILE - 1) / KB_TILE;
    }
    h_tiles[(size_t)nb1] = (uint32_t)tt;
    if (tt >= 0xffffffffull) return 1;
    // pass 1 writes A, pass 2 reads A and writes B, pass 3 reads B and writes the index over A's low stream (dead by then; the
    // eight-byte format takes both of A's streams' worth: it gets a buffer of its own size in place of A's low stream)
    void* d_f_lo = nullptr;
    temps.v.push_back(&d_f_lo);
    // (an allocation that fails here - the guard above works from an estimate of the free memory - is not the job's failure: the
    // temporaries go back and the caller falls back, as it does when the guard says no)
#define KB_ALLOC(p_, bytes_)                              \
    if (dp_dev_malloc(&(p_), (bytes_)) != hipSuccess) {   \
        (void)hipGetLastError();                          \
        return 1;                                         \
    }
    KB_ALLOC(d_a_lo, n * (size_t)(a_lo_wide ? 8 : 4) + 64)
    if (hb1 && hb1 != 8) KB_ALLOC(d_a_hi, n * (size_t)hb1 + 64)
    KB_ALLOC(d_b_lo, n * (size_t)(hb2 == 8 ? 8 : 4) + 64)
    if (hb2 && hb2 != 8) KB_ALLOC(d_b_hi, n * (size_t)hb2 + 64)
    if (fmt == 5) KB_ALLOC(d_f_hi, n + 64)
    if (!f_over_a) KB_ALLOC(d_f_lo, n * 4 + 64)
#undef KB_ALLOC
    const KbBuf A = {(uint32_t*)d_a_lo, d_a_hi, hb1}, B = {(uint32_t*)d_b_lo, d_b_hi, hb2};
    const KbPosOut F = {f_over_a ? d_a_lo : d_f_lo, (uint8_t*)d_f_hi, fmt};
    DP_HIP(hipMemcpyAsync(tile_start, h_tiles.data(), ((size_t)nb1 + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(hipMemcpyAsync(cur1, base1, (size_t)nb1 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(kb_part1, dim3(grid), dim3(KB_THREADS), 0, ctx->stream, (const uint8_t*)ow->d_packed.p, (const uint64_t*)ow->d_boff.p,
                       (const uint32_t*)ow->d_len.p, (const uint32_t*)d_gread, n_groups, G, cur1, A);
    hipLaunchKernelGGL(kb_count2, dim3(grid), dim3(KB_THREADS), 0, ctx->stream, A, (const kb_u64*)base1, (const uint32_t*)tile_start, G, cnt2);
    if (n_sub >= 8192) {  // (cur2 is scratch until the copy below fills it: 256 partial sums and their scan)
        const uint32_t n_wg = 256, chunk = (((n_sub + n_wg - 1) / n_wg) + 1023u) & ~1023u;
        hipLaunchKernelGGL(kb_scan_partials, dim3(n_wg), dim3(1024), 0, ctx->stream, (const kb_u64*)cnt2, n_sub, chunk, cur2);
        hipLaunchKernelGGL(kb_excl_scan_small, dim3(1), dim3(1024), 0, ctx->stream, (const kb_u64*)cur2, n_wg, cur2 + n_wg);
        hipLaunchKernelGGL(kb_scan_write, dim3(n_wg), dim3(1024), 0, ctx->stream, (const kb_u64*)cnt2, n_sub, chunk, (const kb_u64*)(cur2 + n_wg), base2);
    } else {
        hipLaunchKernelGGL(kb_excl_scan_small, dim3(1), dim3(1024), 0, ctx->stream, (const kb_u64*)cnt2, n_sub, base2);
    }
    DP_HIP(hipMemcpyAsync(cur2, base2, (size_t)n_sub * 8, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(kb_part2, dim3(grid), dim3(KB_THREADS), 0, ctx->stream, A, (const kb_u64*)base1, (const uint32_t*)tile_start, G, cur2, B);
    hipLaunchKernelGGL(kb_final, dim3(std::min<uint32_t>(n_sub, (uint32_t)cus * 16)), dim3(KB_THREADS),
                       (size_t)KB_P3_CAP * 8 + ((size_t)8 << G.r), ctx->stream, B, (const kb_u64*)base2, n_sub, G, (const uint32_t*)d_gread,
                       (const uint64_t*)ow->d_boff.p, d_counts, (kb_u64*)d_off, F);
    DP_HIP(hipGetLastError());
    DP_HIP(hipEventRecord(e1, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    if (ms_out) hipEventElapsedTime(ms_out, e0, e1);
    *d_pos_out = f_over_a ? d_a_lo : d_f_lo;  // ownership to the caller
    (f_over_a ? d_a_lo : d_f_lo) = nullptr;
    *d_pos_hi_out = d_f_hi;
    d_f_hi = nullptr;
    *fmt_out = fmt;
    *pbits_out = G.pbits;
    *n_pos_out = n;
    return DP_OK;
}
