// libdownpore_hip.so — seed index build (A13), index query = soft-union of posting bitsets (A14 + A5), exact
// intersection prefilter (A6) and overlap chaining with the per-query ratchet (A7 + A8).  CDNA4 / gfx950 only.
//
// HBM layout
//   posting  : uint64 [n_seeds][W]      row s = set of indexed-sequence indices containing seed s  (W = ceil(M/64))
//   seedsets : uint64 [M][SW]           row i = set of seeds of indexed sequence i                (SW = ceil(S/64))
//   pmeta    : uint32 [n_seeds][4]      {popcount, first non-zero word, last non-zero word, last+1} — exactly the
//                                       count/start/end an Add()-only util.IntSet carries (util/bitset.go:74-108);
//                                       an untouched set keeps NewIntSet's start=1,end=0 (:20-23).
//   cand     : uint64 [n_queries][W]    Matches() result as a bit mask (ascending bit order == ascending ids)
// Index query kernel: one wave per query; lanes own consecutive word indices, so each posting row is read as one
// coalesced 512-byte segment per wave step.  The counting ladder (util/asm_amd64.s:121-509) is held in registers.
#include <algorithm>
#include <cstring>

#include "dp_common.h"
#include "dp_launch.h"

typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------------------------
// A13

// the chaining stage's cursor block: 32 words (cursor, flags, totals) + 64 shards of its algorithmic-byte count (one counter
// would be a thousand same-address atomics per kernel: 12 ns each, in a row)
#define C_CURSOR_BYTES 640
struct index_fill_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const dp_seq_ref* __restrict__ refs, uint32_t n_seqs, const int32_t* __restrict__ segs,
                                  u64* __restrict__ posting, u64* __restrict__ seedsets, uint32_t W, uint32_t SW,
                                  const uint32_t* __restrict__ n_seqs_dev) {
    if (n_seqs_dev) n_seqs = *n_seqs_dev;  // (chunks made on the device: the launch was sized for an upper bound)
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = dp_lane();
    for (uint32_t idx = gw; idx < n_seqs; idx += waves) {
        const dp_seq_ref r = refs[idx];
        for (uint32_t i = lane; i < r.n_seeds; i += 64) {
            uint32_t seed = (uint32_t)segs[r.seg_off + 2 * (uint64_t)i + 1];
            atomicOr(&posting[(uint64_t)seed * W + (idx >> 6)], 1ull << (idx & 63));
            atomicOr(&seedsets[(uint64_t)idx * SW + (seed >> 6)], 1ull << (seed & 63));
        }
    }
}
};

// ---- round 4: the two bit matrices without atomics (sparse regime) ----------------------------------------------------------
// A scattered device-scope atomic costs a five-slot job what it costs alone (HISTORY.md 5.7) and index_fill_kernel issued two per
// indexed seed - 400 k per config-2 round - into matrices that 15 MB of stores had just cleared.  A chunk's seed-set row belongs
// to ONE wave: it is built in LDS (ds_or) and stored whole - rows need no clearing, rows beyond the chunk count are never read.
// The posting matrix is the bit transpose of the seed sets: posting[s][w] bit b = seedsets[64 w + b][s / 64] bit (s % 64).
#define IFR_SW_MAX 512  // seed-set words per row this path holds in LDS (32 768 seeds)
struct index_fill_rows_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const dp_seq_ref* __restrict__ refs, const int32_t* __restrict__ segs, u64* __restrict__ seedsets,
                               uint32_t SW, const uint32_t* __restrict__ n_seqs_dev) {
        __shared__ u64 rows[4][IFR_SW_MAX];
        u64* row = rows[threadIdx.x >> 6];
        const uint32_t n_seqs = *n_seqs_dev;
        const uint32_t waves = gridDim.x * (blockDim.x >> 6);
        const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const int lane = dp_lane();
        for (uint32_t idx = gw; idx < n_seqs; idx += waves) {
            for (uint32_t x = lane; x < SW; x += 64) row[x] = 0ull;
            __builtin_amdgcn_wave_barrier();
            const dp_seq_ref r = refs[idx];
            for (uint32_t i = lane; i < r.n_seeds; i += 64) {
                const uint32_t seed = (uint32_t)segs[r.seg_off + 2 * (uint64_t)i + 1];
                atomicOr(&row[seed >> 6], 1ull << (seed & 63));
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t x = lane; x < SW; x += 64) seedsets[(uint64_t)idx * SW + x] = row[x];
            __builtin_amdgcn_wave_barrier();
        }
    }
};
// one wave per (eight words of posting rows, FOUR consecutive words of seed-set rows): thirty-two 64 x 64 bit transposes across the lanes;
// lane j ends up with the eight words of seeds 64 (sw0 + i) + j, i = 0 .. 3 - four 64-byte stores.  (Round 6: one seed-set word per wave
// and task made every lane's 8-byte load a sector of its own - 64 rows, 2.5 KB apart - of which an eighth was used: the launch
// fetched 808 MB for a 240 MB matrix at k = 10, profiles/r05/pmc_dense.json.  Four words per lane are a lane's whole 32-byte sector,
// and the four waves of a workgroup take the four quarters of the rows' 128-byte lines.)
struct posting_transpose_kernel {
    enum { THREADS = 256, SWPT = 4 };
    static __device__ void run(const u64* __restrict__ seedsets, u64* __restrict__ posting, uint32_t n_seeds, uint32_t W, uint32_t SW,
                               const uint32_t* __restrict__ n_seqs_dev) {
        const uint32_t n_seqs = *n_seqs_dev;
        const uint32_t wchunks = (W + 7) / 8;
        const uint32_t swg = (SW + SWPT - 1) / SWPT;
        const uint32_t tasks = wchunks * swg;
        const uint32_t waves = gridDim.x * (blockDim.x >> 6);
        const int lane = dp_lane();
        for (uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < tasks; t += waves) {
            const uint32_t sw0 = (t % swg) * SWPT, w0 = (t / swg) * 8;
            u64 x[8][SWPT];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint64_t rowi = (uint64_t)(w0 + u) * 64 + (uint32_t)lane;
                const bool rv = w0 + u < W && rowi < n_seqs;
                const u64* src = seedsets + rowi * SW + sw0;
#pragma unroll
                for (int i = 0; i < SWPT; i++) x[u][i] = (rv && sw0 + i < SW) ? src[i] : 0ull;
            }
#pragma unroll
            for (int i = 0; i < SWPT; i++) {
                // 64 x 64 bit transpose across the lanes (row r of the block in lane r): six exchange stages instead of 64 ballots
                u64 out[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    u64 v = x[u][i];
                    if (__ballot(v != 0) != 0) {  // (else: sixty-four chunks none of which holds any of these sixty-four seeds)
#define PT_STAGE(j_, m_)                                                              \
    {                                                                                 \
        const u64 y_ = (u64)__shfl_xor((unsigned long long)v, (j_), 64);              \
        if ((lane & (j_)) == 0) v ^= ((((v >> (j_)) ^ y_) & (m_)) << (j_));           \
        else v ^= (((y_ >> (j_)) ^ v) & (m_));                                        \
    }
                        PT_STAGE(32, 0x00000000FFFFFFFFull)
                        PT_STAGE(16, 0x0000FFFF0000FFFFull)
                        PT_STAGE(8, 0x00FF00FF00FF00FFull)
                        PT_STAGE(4, 0x0F0F0F0F0F0F0F0Full)
                        PT_STAGE(2, 0x3333333333333333ull)
                        PT_STAGE(1, 0x5555555555555555ull)
#undef PT_STAGE
                    }
                    out[u] = v;
                }
                const uint32_t seed = (sw0 + (uint32_t)i) * 64 + (uint32_t)lane;
                if (sw0 + (uint32_t)i < SW && seed < n_seeds) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (w0 + u < W) posting[(uint64_t)seed * W + w0 + u] = out[u];
                }
            }
        }
    }
};

struct posting_meta_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const u64* __restrict__ posting, uint32_t n_seeds, uint32_t W, uint32_t* __restrict__ pmeta) {
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = dp_lane();
    for (uint32_t s = gw; s < n_seeds; s += waves) {
        int cnt = 0, first = 0x7fffffff, last = -1;
        for (uint32_t w = lane; w < W; w += 64) {
            u64 v = posting[(uint64_t)s * W + w];
            if (v) {
                cnt += __popcll(v);
                first = min(first, (int)w);
                last = max(last, (int)w);
            }
        }
        cnt = wave_sum(cnt);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            first = min(first, __shfl_xor(first, d, 64));
            last = max(last, __shfl_xor(last, d, 64));
        }
        if (lane == 0) {
            uint32_t st = 1, en = 0;  // NewIntSet(): start 1, end 0
            if (last >= 0) {
                st = (uint32_t)first;
                en = (uint32_t)last;
            }
            pmeta[4 * s + 0] = (uint32_t)cnt;
            pmeta[4 * s + 1] = st;
            pmeta[4 * s + 2] = en;
            pmeta[4 * s + 3] = en + 1;
        }
    }
}
};

int dp_index_build_impl(dp_ctx* ctx, const dp_seq_ref* seqs, uint32_t n_seqs) {
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_index_build before dp_round_begin");
    hipSetDevice(ctx->device);
    for (uint32_t i = 0; i < n_seqs; i++)
        if (seqs[i].seg_off + 2ull * seqs[i].n_seeds + 1 > ctx->n_segs)
            return dp_fail(ctx, DP_ERR_ARG, "dp_index_build: sequence view outside the scan output");
    const uint32_t S = ctx->n_seeds;
    const uint32_t W = std::max<uint32_t>(1, (n_seqs + 63) / 64), SW = std::max<uint32_t>(1, (S + 63) / 64);
    ctx->max_seq_seeds = 0;
    for (uint32_t i = 0; i < n_seqs; i++) ctx->max_seq_seeds = std::max(ctx->max_seq_seeds, seqs[i].n_seeds);
    ctx->n_seqs = n_seqs;
    ctx->W = W;
    ctx->SW = SW;
    ctx->word_base = 0;  // (a fresh index is a whole one until dp_index_set_global says otherwise)
    ctx->global_n_seqs = 0;
    ctx->chunks_on_device = false;
    if (dev_reserve(ctx, ctx->d_seqrefs, (size_t)n_seqs * sizeof(dp_seq_ref) + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_posting, (size_t)S * W * 8 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_seedsets, (size_t)n_seqs * SW * 8 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_pmeta, (size_t)S * 16 + 16)) return DP_ERR_HIP;
    {
        const dp_zero_region z[2] = {{ctx->d_posting.p, (size_t)S * W * 8 + 64}, {ctx->d_seedsets.p, (size_t)n_seqs * SW * 8 + 64}};
        if (int rc = dp_zero_regions(ctx, z, 2)) return rc;
    }
    if (n_seqs) {
        // (seqs is borrowed only for the duration of the call: the copy leaves from the context's staging block, nothing waits)
        DP_HIP(hipMemcpyAsync(ctx->d_seqrefs.p, dp_stage(ctx, seqs, (size_t)n_seqs * sizeof(dp_seq_ref)), (size_t)n_seqs * sizeof(dp_seq_ref),
                              hipMemcpyHostToDevice, ctx->stream));
        uint32_t blocks = std::min<uint32_t>(2048, (n_seqs + 3) / 4);
        dp_launch<index_fill_kernel>(ctx, dim3(blocks), dim3(256), (const dp_seq_ref*)ctx->d_seqrefs.p, n_seqs,
                           (const int32_t*)ctx->d_segs.p, (u64*)ctx->d_posting.p, (u64*)ctx->d_seedsets.p, W, SW, (const uint32_t*)nullptr);
        DP_HIP(hipGetLastError());
    }
    if (S) {
        uint32_t blocks = std::min<uint32_t>(2048, (S + 3) / 4);
        dp_launch<posting_meta_kernel>(ctx, dim3(blocks), dim3(256), (const u64*)ctx->d_posting.p, S, W,
                           (uint32_t*)ctx->d_pmeta.p);
        DP_HIP(hipGetLastError());
    }
    return DP_OK;  // (errors of the queued work surface at the next call that waits)
}

extern "C" int dp_index_build(dp_ctx* ctx, const dp_seq_ref* seqs, uint32_t n_seqs) {
    if (!ctx || (n_seqs && !seqs)) return DP_ERR_ARG;
    return dp_index_build_impl(ctx, seqs, n_seqs);
}

// ---------------------------------------------------------------------------------------------------------------
// A12 on the device: overlap.chunkWorker (overlap/overlap.go:253-318) for every surviving read of the last dp_scan_reads,
// then AddSequence + IndexSequences.  One workgroup walks the survivors in file order, 1024 at a time: every thread runs the
// reference's state machine over its read's segments twice (count the chunks, then write them at the position a block
// scan gives), so chunks come out in the reference's order without any host step.
struct ChunkParams {
    const uint32_t* s_item;
    const uint32_t* s_count;
    const u64* s_off;
    uint32_t ns, lo;
    const uint32_t* read_len;
    const int32_t* segs;
    int k;
    long long chunkSize, overlap;
    int minSeeds, inset;
    dp_seq_ref* refs;
    dp_seq_meta* metas;
    uint32_t cap;
    uint32_t* n_out;  // [0] chunks, [1] overflow, [2] ticket, [3] tiles done ([2], [3] and the tile status words are zero between launches)
    // the bit matrices index_fill_kernel (next launch) sets bits in are cleared by `zero_blocks` extra workgroups of this launch
    uint4* z_p[2];
    unsigned long long z_n16[2];
    uint32_t zero_blocks;
    // ... and they bring an announced query block over (dp_query_prestage): 16-byte words, pinned host memory -> device
    uint4* f_dst;
    const uint4* f_src;
    unsigned long long f_n16;
    // launched behind a scan that nobody has waited for yet (dp_index_prechain, round 4): the survivor count is read from the scan's
    // totals (survivors + extra items at [1]), the grid was sized from a guess, and a scan whose own guesses failed (segments
    // beyond seg_cap at [0], the sort pass's overflow word at [4], record shards full at [6]) left no segments to chunk
    const u64* scan_totals;  // null: `ns` is exact and the scan's output complete
    uint32_t n_extra, grid_tiles;
    u64 seg_cap;
    // ... and the host does not wait for the stream but for this word (pinned): that this kernel runs says that the scan's kernels
    // are done and what they wrote - the extra items' segments in the host's pinned block among it - has arrived
    uint32_t* done_flag;
    uint32_t done_seq;
};

// (round 6) CK_CACHE chunks of a thread's count pass stay in the workgroup's LDS: a read of the dense regime is two chunks, and the
// write pass used to walk its ~200 seeds a second time (4-byte loads, every lane of a wave in another read's slice)
#define CK_CACHE 2
struct ChunkCache {
    uint4 c[CK_CACHE];   // first seed, seeds, length, offset
    int32_t ins[CK_CACHE];
};
template <bool WRITE>
__device__ uint32_t chunk_one(const ChunkParams& P, uint32_t i, uint32_t at, ChunkCache* cache = nullptr) {
    const uint32_t read = P.s_item[i] + P.lo;
    const int numSeeds = (int)P.s_count[i];
    const u64 segBase = P.s_off[i];
    const int32_t* seg = P.segs + segBase;
    const int n = 2 * numSeeds + 1, k = P.k;
    const long long length = (long long)P.read_len[read];
    uint32_t made = 0;
    auto emit = [&](int first, int last, long long len, long long off, long long ins) {
        if (WRITE) {
            const uint32_t o = at + made;
            if (o < P.cap) {
                dp_seq_ref r;
                r.seg_off = segBase + 2ull * (uint32_t)first;
                r.n_seeds = (uint32_t)(last - first + 1);
                r.reserved = 0;
                P.refs[o] = r;
                dp_seq_meta m;
                m.read = read;
                m.length = (int32_t)len;
                m.offset = (int32_t)off;
                m.inset = (int32_t)ins;
                P.metas[o] = m;
            }
        } else if (cache && made < CK_CACHE) {
            cache->c[made] = make_uint4((uint32_t)first, (uint32_t)(last - first + 1), (uint32_t)(int32_t)len, (uint32_t)(int32_t)off);
            cache->ins[made] = (int32_t)ins;
        }
        made++;
    };
    auto whole = [&]() { emit(0, numSeeds - 1, length, 0, (long long)P.inset); };
    auto nextSeedOffset = [&](int idx) -> long long { return (long long)seg[idx * 2 + 2] + k; };
    const long long numChunks = length / P.chunkSize + 1;
    if (numChunks == 1 || numSeeds < P.minSeeds * 3) {
        if (numSeeds >= P.minSeeds) whole();
        return made;
    }
    int prevSeedIndex = 0;
    long long totalOffset = seg[0];  // seedOffset(0, k)
    long long lengthInBases = 0;
    for (;;) {
        int seedCount = 0;
        if (prevSeedIndex >= numSeeds - 150) {
            if (prevSeedIndex == 0) {
                whole();
            } else {
                const long long newFirstGap = nextSeedOffset(prevSeedIndex - 1) - k;
                long long fromEnd = seg[n - 1];  // seedOffsetFromEnd(prevSeedIndex, k)
                {
                    // (eight loads in flight: one gap per trip to memory was a hundred dependent trips for a read of the dense regime)
                    int x = n - 3;
                    const int stop = prevSeedIndex * 2 + 1;
                    for (; x - 14 > stop; x -= 16) {
                        int32_t g[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) g[u] = seg[x - 2 * u];
#pragma unroll
                        for (int u = 0; u < 8; u++) fromEnd += g[u] + k;
                    }
                    for (; x > stop; x -= 2) fromEnd += seg[x] + k;
                }
                lengthInBases += fromEnd + k + newFirstGap;
                emit(prevSeedIndex, numSeeds - 1, lengthInBases, totalOffset - newFirstGap, 0);
            }
            break;
        }
        // (the gaps of the next eight seeds are asked for together; the sums and the three exit tests stay in the reference's order)
        while (lengthInBases < P.chunkSize && seedCount < 100 && prevSeedIndex + seedCount < numSeeds) {
            const int at0 = prevSeedIndex + seedCount;
            const int have = min(8, min(100 - seedCount, numSeeds - at0));
            int32_t g[8];
#pragma unroll
            for (int u = 0; u < 8; u++) g[u] = u < have ? seg[(at0 + u) * 2 + 2] : 0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (u < have && lengthInBases < P.chunkSize) {
                    lengthInBases += (long long)g[u] + k;
                    seedCount++;
                }
            }
        }
        if (seedCount >= P.minSeeds) {
            const long long newFirstGap = nextSeedOffset(prevSeedIndex - 1) - k;
            lengthInBases += newFirstGap;
            emit(prevSeedIndex, prevSeedIndex + seedCount - 1, lengthInBases, totalOffset - newFirstGap,
                 length - totalOffset - lengthInBases + newFirstGap);
            totalOffset += lengthInBases - newFirstGap;
            lengthInBases = 0;
            prevSeedIndex += seedCount;
            if (prevSeedIndex >= numSeeds) break;
            for (seedCount = 0; seedCount < 5 && lengthInBases < P.overlap / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                const long long step = nextSeedOffset(prevSeedIndex);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        } else {
            prevSeedIndex += seedCount;
            for (seedCount = 0; lengthInBases < P.overlap / 2 && prevSeedIndex > 0; seedCount++) {
                prevSeedIndex--;
                const long long step = nextSeedOffset(prevSeedIndex);
                lengthInBases += step;
                totalOffset -= step;
            }
            lengthInBases = 0;
        }
    }
    return made;
}

// status word of a tile of 1024 survivors: flag << 62 | chunks (flag 1 = the tile's own count, 2 = inclusive prefix); tiles take
// their number from a ticket, so a predecessor is always running or done (the look-back cannot wait for a tile that has no CU)
struct chunk_kernel {
    enum { THREADS = 1024 };
    static __device__ void run(const ChunkParams P, unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t tile_s, base_s;
    __shared__ ChunkCache cache[THREADS];
    uint32_t ns = P.ns;
    if (P.done_flag && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(P.done_flag, P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (P.scan_totals) {
        const u64 all = P.scan_totals[1];
        ns = all > P.n_extra ? (uint32_t)(all - P.n_extra) : 0u;
        const bool scan_failed = P.scan_totals[0] > P.seg_cap || (uint32_t)P.scan_totals[4] != 0u || (uint32_t)P.scan_totals[6] != 0u;
        if (scan_failed || (ns + 1023u) / 1024u > P.grid_tiles) {  // nothing to chunk / not enough tiles: the host launches again
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                P.n_out[0] = 0u;
                P.n_out[1] = 2u;
            }
            return;
        }
    }
    // (a launch shared with other rounds has the largest round's grid: blocks beyond this round's own tiles take no ticket)
    const uint32_t n_tiles = (ns + 1023u) / 1024u;
    if (n_tiles == 0 && blockIdx.x == 0 && threadIdx.x == 0) {  // (no survivor: no tile writes the count)
        P.n_out[0] = 0u;
        P.n_out[1] = 0u;
    }
    if (blockIdx.x >= n_tiles) {
        const uint32_t zb = blockIdx.x - n_tiles;
        if (zb >= P.zero_blocks) return;
        for (unsigned long long j = (unsigned long long)zb * 1024 + threadIdx.x; j < P.f_n16; j += (unsigned long long)P.zero_blocks * 1024)
            P.f_dst[j] = P.f_src[j];
        const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
            for (unsigned long long j = (unsigned long long)zb * 1024 + threadIdx.x; j < P.z_n16[r]; j += (unsigned long long)P.zero_blocks * 1024)
                P.z_p[r][j] = zero;
        return;
    }
    if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = tile_s;
    const int lane = dp_lane(), wave = threadIdx.x >> 6;
    const uint32_t i = tile * 1024 + threadIdx.x;
    const uint32_t cnt = i < ns ? chunk_one<false>(P, i, 0, &cache[threadIdx.x]) : 0u;
    const uint32_t x = (uint32_t)wave_incl_sum_dpp((int)cnt);
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) before += wsum[w];
        total += wsum[w];
    }
    if (wave == 0) {  // (the whole wave looks back: dp_wave_lookback)
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&status[0], (2ull << 62) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&status[tile], (1ull << 62) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            excl = dp_wave_lookback(status, tile, lane);
            if (lane == 0) __hip_atomic_store(&status[tile], (2ull << 62) | (excl + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            base_s = (uint32_t)excl;
            if ((uint64_t)(tile + 1) * 1024 >= ns) {  // last tile: the totals
                const unsigned long long all = excl + total;
                P.n_out[0] = (uint32_t)min(all, (unsigned long long)P.cap);
                P.n_out[1] = all > P.cap ? 1u : 0u;
            }
        }
    }
    __syncthreads();
    if (cnt > CK_CACHE) {
        chunk_one<true>(P, i, base_s + before + x - cnt);
    } else if (cnt) {  // from the count pass's cache (own LDS slot: no barrier needed in between)
        const uint32_t read = P.s_item[i] + P.lo;
        const u64 segBase = P.s_off[i];
        for (uint32_t m_ = 0; m_ < cnt; m_++) {
            const uint32_t o = base_s + before + x - cnt + m_;
            if (o >= P.cap) break;
            const uint4 c = cache[threadIdx.x].c[m_];
            dp_seq_ref r;
            r.seg_off = segBase + 2ull * c.x;
            r.n_seeds = c.y;
            r.reserved = 0;
            P.refs[o] = r;
            dp_seq_meta m;
            m.read = read;
            m.length = (int32_t)c.z;
            m.offset = (int32_t)c.w;
            m.inset = cache[threadIdx.x].ins[m_];
            P.metas[o] = m;
        }
    }
    // the tile that finishes last puts the ticket, the status words and this counter back to zero for the next launch
    __syncthreads();
    if (threadIdx.x == 0) tile_s = __hip_atomic_fetch_add(&P.n_out[3], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __syncthreads();
    if (tile_s == n_tiles) {
        for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024) status[t] = 0ull;
        if (threadIdx.x == 0) {
            *ticket = 0u;
            P.n_out[3] = 0u;
        }
    }
}
};

// upper bound of the chunks chunkWorker makes of a read with `numSeeds` hits: every chunk but the last holds >= minSeeds seeds
// and the walk backs up at most 5 seeds (or overlap/2 bases) after each
static uint32_t chunk_cap_of(uint32_t numSeeds, long long length, long long chunkSize, int minSeeds) {
    if (length / chunkSize + 1 == 1 || (int)numSeeds < minSeeds * 3 || numSeeds <= 150) return 1;  // (<= 150 seeds: the walk ends at once)
    const int step = std::max(1, minSeeds - 5);
    return (numSeeds - 150) / (uint32_t)step + 3;
}

// Launches chunk_kernel + index_fill_kernel + posting_meta_kernel for `cap` chunks at most.  scan_totals == null: n_survivors is
// exact (the caller has the scan's result).  Otherwise (dp_index_prechain): behind a scan nobody has waited for - the survivor
// count is read on the device, n_survivors only sizes the grid.
static int index_chunked_launch(dp_ctx* ctx, int64_t chunk_size, int64_t overlap, uint32_t min_seeds, int32_t inset, uint32_t n_survivors,
                                uint32_t cap, const u64* scan_totals, uint32_t n_extra, u64 seg_cap, uint32_t* done_flag = nullptr,
                                uint32_t done_seq = 0) {
    const uint32_t n_items = ctx->scan_items;
    const uint32_t* s_item = (const uint32_t*)ctx->d_surv.p;
    const uint32_t* s_count = s_item + n_items;
    const u64* s_off = (const u64*)(s_count + n_items + (n_items & 1));
    const uint32_t S = ctx->n_seeds;
    const uint32_t W = std::max<uint32_t>(1, (cap + 63) / 64), SW = std::max<uint32_t>(1, (S + 63) / 64);
    ctx->n_seqs = cap;
    ctx->W = W;
    ctx->SW = SW;
    ctx->word_base = 0;
    ctx->global_n_seqs = 0;
    ctx->chunks_on_device = true;
    if (dev_reserve(ctx, ctx->d_seqrefs, (size_t)cap * sizeof(dp_seq_ref) + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_chunk_meta, (size_t)cap * sizeof(dp_seq_meta) + 16)) return DP_ERR_HIP;
    const uint32_t n_tiles = (n_survivors + 1023) / 1024;
    const size_t b_nseqs = 64 + ((size_t)n_tiles + 2) * 8;  // [0] chunks, [1] overflow, [2] ticket | tile status words
    {
        // [2 ..] (ticket, tiles done, tile status) are zero between launches: chunk_kernel's last tile sees to it; a new block starts zero
        const void* before = ctx->d_nseqs.p;
        if (dev_reserve(ctx, ctx->d_nseqs, b_nseqs)) return DP_ERR_HIP;
        if (ctx->d_nseqs.p != before) DP_HIP(hipMemsetAsync(ctx->d_nseqs.p, 0, ctx->d_nseqs.cap, ctx->stream));
    }
    if (dev_reserve(ctx, ctx->d_posting, (size_t)S * W * 8 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_seedsets, (size_t)cap * SW * 8 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_pmeta, (size_t)S * 16 + 16)) return DP_ERR_HIP;
    const size_t zb_post = ((size_t)S * W * 8 + 64 + 15) & ~(size_t)15, zb_sets = ((size_t)cap * SW * 8 + 64 + 15) & ~(size_t)15;
    // DP_INDEX_FILL_ROWS=1: the matrices without atomics and without clearing (index_fill_rows_kernel + posting_transpose_kernel)
    // where a seed-set row fits the wave's LDS buffer.  Built, bit-identical (tests run both) and OFF in the sparse regime: alone the three
    // launches take 8.2 + 5.8 + 14.2 us against 10.3 + 12.8 for chunk + fill with atomics (the transpose reads every row word
    // uncoalesced), and six alternating whole-job runs on one box gave 0.159 ms per round with the atomics against 0.161 without
    // (profiles/r04/ab_rows3.txt) - the 400 k atomics it removes cost less than the transpose it adds.
    // Where the matrices are large it wins: at k = 10 (100 k chunks x 20 k seeds: 250 MB each, 20 M atomics) a round's index build
    // and query stage take 1.22 ms instead of 1.67 (profiles/r04/dense_rows.txt) - the default from 4 M seed-set words (32 MB) up.
    const char* ife = getenv("DP_INDEX_FILL_ROWS");  // (read per call: tests switch it between jobs of one process; 0 / 1 force)
    const bool rows_fit = SW <= IFR_SW_MAX && S > 0;
    const bool rows_mode = rows_fit && (ife ? ife[0] == '1' : (uint64_t)cap * SW >= ((uint64_t)4 << 20));
    if (!n_survivors) {  // (no chunk_kernel launch to clear the matrices and to write the chunk count)
        const dp_zero_region z[3] = {{ctx->d_posting.p, zb_post}, {ctx->d_seedsets.p, zb_sets}, {ctx->d_nseqs.p, 8}};
        if (int rc = dp_zero_regions(ctx, z, 3)) return rc;
    }
    DP_HIP(dp_mark(ctx, 8));
    ctx->index_marked = ctx->timing_on;
    if (n_survivors) {
        ChunkParams P;
        P.s_item = s_item;
        P.s_count = s_count;
        P.s_off = s_off;
        P.ns = n_survivors;
        P.lo = ctx->chunk_lo;
        P.read_len = (const uint32_t*)ctx->d_len.p;
        P.segs = (const int32_t*)ctx->d_segs.p;
        P.k = ctx->k;
        P.chunkSize = chunk_size;
        P.overlap = overlap;
        P.minSeeds = (int)min_seeds;
        P.inset = inset;
        P.refs = (dp_seq_ref*)ctx->d_seqrefs.p;
        P.metas = (dp_seq_meta*)ctx->d_chunk_meta.p;
        P.cap = cap;
        P.n_out = (uint32_t*)ctx->d_nseqs.p;
        P.z_p[0] = (uint4*)ctx->d_posting.p;
        P.z_n16[0] = rows_mode ? 0 : zb_post / 16;
        P.z_p[1] = (uint4*)ctx->d_seedsets.p;
        P.z_n16[1] = rows_mode ? 0 : zb_sets / 16;
        P.zero_blocks = rows_mode ? 8u : (uint32_t)std::max<size_t>(1, std::min<size_t>(512, (std::max(zb_post, zb_sets) / 16 + 4095) / 4096));
        P.f_dst = nullptr;
        P.f_src = nullptr;
        P.f_n16 = 0;
        P.scan_totals = scan_totals;
        P.n_extra = n_extra;
        P.grid_tiles = n_tiles;
        P.seg_cap = seg_cap;
        P.done_flag = done_flag;
        P.done_seq = done_seq;
        if (!scan_totals && ctx->q_pre_bytes && !ctx->q_pre_fetched && ctx->h_qup.p && ctx->d_qsegs.p) {  // (both blocks are 64 bytes longer than the data)
            P.f_dst = (uint4*)ctx->d_qsegs.p;
            P.f_src = (const uint4*)ctx->h_qup.p;
            P.f_n16 = (ctx->q_pre_bytes + 15) / 16;
            ctx->q_pre_fetched = true;
        }
        dp_launch<chunk_kernel>(ctx, dim3(n_tiles + P.zero_blocks), dim3(1024), P, (unsigned long long*)((uint8_t*)ctx->d_nseqs.p + 64),
                           (uint32_t*)ctx->d_nseqs.p + 2);
        DP_HIP(hipGetLastError());
        if (cap && rows_mode) {
            const uint32_t blocks = std::min<uint32_t>(2048, (cap + 3) / 4);
            dp_launch<index_fill_rows_kernel>(ctx, dim3(blocks), dim3(256), (const dp_seq_ref*)ctx->d_seqrefs.p, (const int32_t*)ctx->d_segs.p,
                                              (u64*)ctx->d_seedsets.p, SW, (const uint32_t*)ctx->d_nseqs.p);
            const uint32_t tasks = ((W + 7) / 8) * ((SW + posting_transpose_kernel::SWPT - 1) / posting_transpose_kernel::SWPT);
            dp_launch<posting_transpose_kernel>(ctx, dim3(std::min<uint32_t>(4096, (tasks + 3) / 4)), dim3(256), (const u64*)ctx->d_seedsets.p,
                                                    (u64*)ctx->d_posting.p, S, W, SW, (const uint32_t*)ctx->d_nseqs.p);
            DP_HIP(hipGetLastError());
        } else if (cap) {
            const uint32_t blocks = std::min<uint32_t>(2048, (cap + 3) / 4);
            dp_launch<index_fill_kernel>(ctx, dim3(blocks), dim3(256), (const dp_seq_ref*)ctx->d_seqrefs.p, cap,
                               (const int32_t*)ctx->d_segs.p, (u64*)ctx->d_posting.p, (u64*)ctx->d_seedsets.p, W, SW, (const uint32_t*)ctx->d_nseqs.p);
            DP_HIP(hipGetLastError());
        }
    } else if (rows_mode && S) {
        // (cap == 0 cannot happen with survivors; no survivors: the matrices were cleared above)
    }
    if (S) {
        const uint32_t blocks = std::min<uint32_t>(2048, (S + 3) / 4);
        dp_launch<posting_meta_kernel>(ctx, dim3(blocks), dim3(256), (const u64*)ctx->d_posting.p, S, W, (uint32_t*)ctx->d_pmeta.p);
        DP_HIP(hipGetLastError());
    }
    DP_HIP(dp_mark(ctx, 9));
    return DP_OK;
}

// dp_index_prechain: the caller announces the dp_index_build_chunked call it will make after its next dp_scan_reads.  An index-mode
// scan that runs in one go (dp_kindex.hip) then launches the three kernels itself, directly behind its own and before anybody
// waits - for the survivors the device finds, in buffers sized from the previous round of this context (1.25 x its chunk bound,
// 1.5 x its survivors) - and returns as soon as its own output has arrived; the chunk stage runs while the host prepares the
// queries.  dp_index_build_chunked with the same parameters then only checks the guesses against the exact bound (and launches
// again, the old way, for the rare round that outgrew them).
extern "C" int dp_index_prechain(dp_ctx* ctx, int64_t chunk_size, int64_t overlap, uint32_t min_seeds, int32_t inset) {
    if (!ctx || chunk_size < 1) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_index_prechain: bad arguments") : DP_ERR_ARG;
    ctx->pc_armed = true;
    ctx->pc_launched = false;
    ctx->pc_chunk_size = chunk_size;
    ctx->pc_overlap = overlap;
    ctx->pc_min_seeds = min_seeds;
    ctx->pc_inset = inset;
    return DP_OK;
}
extern "C" int dp_index_prechained(const dp_ctx* ctx) { return ctx && ctx->pc_launched ? 1 : 0; }

// called by the one-go index step of dp_scan_reads once its own kernels are queued (dp_scan.hip)
int dp_index_prechain_launch(dp_ctx* ctx, const u64* scan_totals, uint32_t n_extra, u64 seg_cap, uint32_t* done_flag, uint32_t done_seq) {
    ctx->pc_launched = false;
    static const bool off = false;
    if (!ctx->pc_armed || off) return DP_OK;
    ctx->pc_armed = false;
    if (!ctx->pc_prev_cap) return DP_OK;  // (no round of this context to size from yet)
    const uint32_t cap = (uint32_t)std::min<u64>(0xfffffff0ull, (((u64)ctx->pc_prev_cap + ctx->pc_prev_cap / 4 + 64) + 63) & ~63ull);
    const uint32_t ns_guess = std::max<uint32_t>(4096u, ctx->pc_prev_ns + ctx->pc_prev_ns / 2);
    if (int rc = index_chunked_launch(ctx, ctx->pc_chunk_size, ctx->pc_overlap, ctx->pc_min_seeds, ctx->pc_inset, ns_guess, cap, scan_totals,
                                      n_extra, seg_cap, done_flag, done_seq))
        return rc;
    ctx->pc_launched = true;
    ctx->pc_cap = cap;
    ctx->pc_tiles = (ns_guess + 1023) / 1024;
    return DP_OK;
}
void dp_index_prechain_cancel(dp_ctx* ctx) { ctx->pc_launched = false; }

extern "C" int dp_index_build_chunked(dp_ctx* ctx, int64_t chunk_size, int64_t overlap, uint32_t min_seeds, int32_t inset,
                                      uint32_t n_survivors, uint32_t* n_seqs_cap) {
    if (!ctx || chunk_size < 1 || !n_seqs_cap) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_index_build_chunked: bad arguments") : DP_ERR_ARG;
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_index_build_chunked before dp_round_begin");
    if (!ctx->h_surv.p || !ctx->d_surv.p) return dp_fail(ctx, DP_ERR_STATE, "dp_index_build_chunked: no dp_scan_reads result on this context");
    hipSetDevice(ctx->device);
    const uint32_t n_items = ctx->scan_items;
    if (n_survivors > n_items) return dp_fail(ctx, DP_ERR_ARG, "dp_index_build_chunked: more survivors than scan items");
    // the library still has the survivors' hit counts and read ids on the host (pinned output of the scan)
    uint64_t cap64 = 0;
    {
        const uint32_t* h_item = (const uint32_t*)ctx->h_surv.p;  // (already turned into read ids by dp_scan_reads)
        const uint32_t* h_count = h_item + ctx->last_surv_all;
        const std::vector<uint32_t>& lens = ctx->owner ? ctx->owner->h_len : ctx->h_len;
        for (uint32_t i = 0; i < n_survivors; i++) cap64 += chunk_cap_of(h_count[i], (long long)lens[h_item[i]], chunk_size, (int)min_seeds);
    }
    if (cap64 > 0xfffffff0ull) return dp_fail(ctx, DP_ERR_CAPACITY, "dp_index_build_chunked: more than 2^32 chunks");
    ctx->pc_prev_cap = (uint32_t)cap64;
    ctx->pc_prev_ns = n_survivors;
    const bool chained = ctx->pc_launched;
    ctx->pc_launched = false;
    if (chained && chunk_size == ctx->pc_chunk_size && overlap == ctx->pc_overlap && min_seeds == ctx->pc_min_seeds && inset == ctx->pc_inset &&
        n_survivors == ctx->last_surv_all - ctx->last_n_extra && cap64 <= ctx->pc_cap && (n_survivors + 1023) / 1024 <= ctx->pc_tiles) {
        // launched behind the scan already, for exactly these survivors, with room for every chunk they can make
        ctx->pc_hits++;
        *n_seqs_cap = ctx->pc_cap;
        return DP_OK;
    }
    if (chained) ctx->pc_misses++;
    const uint32_t cap = (uint32_t)cap64;
    *n_seqs_cap = cap;
    return index_chunked_launch(ctx, chunk_size, overlap, min_seeds, inset, n_survivors, cap, nullptr, 0, 0);
}

// the chunks dp_index_build_chunked made, for a caller that needs them on the host (the host consensus path of the windows
// the device flags): exact count, views and {read, length, offset, inset}
extern "C" int dp_index_chunks(dp_ctx* ctx, dp_seq_ref* refs_out, dp_seq_meta* metas_out, uint32_t cap, uint32_t* n_out) {
    if (!ctx || !n_out) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_index_chunks: bad arguments") : DP_ERR_ARG;
    if (!ctx->chunks_on_device) return dp_fail(ctx, DP_ERR_STATE, "dp_index_chunks: the index was not built by dp_index_build_chunked");
    hipSetDevice(ctx->device);
    uint32_t nn[2] = {0, 0};
    DP_HIP(hipMemcpyAsync(nn, ctx->d_nseqs.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    if (nn[1]) return dp_fail(ctx, DP_ERR_CAPACITY, "dp_index_build_chunked: chunk bound exceeded");
    *n_out = nn[0];
    if (refs_out && metas_out) {
        if (cap < nn[0]) return dp_fail(ctx, DP_ERR_ARG, "dp_index_chunks: output too small");
        DP_HIP(hipMemcpyAsync(refs_out, ctx->d_seqrefs.p, (size_t)nn[0] * sizeof(dp_seq_ref), hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(hipMemcpyAsync(metas_out, ctx->d_chunk_meta.p, (size_t)nn[0] * sizeof(dp_seq_meta), hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
    }
    return DP_OK;
}

// ---- an index that is one shard of a larger one (map against a reference spread over several GPUs) ----------------------
extern "C" int dp_index_meta(dp_ctx* ctx, uint32_t* meta_out, uint32_t n_seeds) {
    if (!ctx || !meta_out || n_seeds != ctx->n_seeds) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_index_meta: bad arguments") : DP_ERR_ARG;
    hipSetDevice(ctx->device);
    DP_HIP(hipMemcpyAsync(meta_out, ctx->d_pmeta.p, (size_t)n_seeds * 16, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    return DP_OK;
}
extern "C" int dp_index_set_global(dp_ctx* ctx, const uint32_t* meta_global, uint32_t n_seeds, uint32_t word_base, uint32_t n_seqs_global) {
    if (!ctx || !meta_global || n_seeds != ctx->n_seeds) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_index_set_global: bad arguments") : DP_ERR_ARG;
    if (n_seqs_global < (uint64_t)word_base * 64 + ctx->n_seqs) return dp_fail(ctx, DP_ERR_ARG, "dp_index_set_global: the shard does not fit the global index");
    hipSetDevice(ctx->device);
    DP_HIP(hipMemcpyAsync(ctx->d_pmeta.p, meta_global, (size_t)n_seeds * 16, hipMemcpyHostToDevice, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    ctx->word_base = word_base;
    ctx->global_n_seqs = n_seqs_global;
    return DP_OK;
}

extern "C" int dp_index_posting_row(dp_ctx* ctx, uint32_t seed, u64* words, uint32_t cap_words, uint32_t* n_words,
                                    uint32_t* count, uint32_t* start, uint32_t* end) {
    if (!ctx || seed >= ctx->n_seeds || cap_words < ctx->W) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    uint32_t meta[4];
    DP_HIP(hipMemcpy(words, (u64*)ctx->d_posting.p + (uint64_t)seed * ctx->W, (size_t)ctx->W * 8, hipMemcpyDeviceToHost));
    DP_HIP(hipMemcpy(meta, (uint32_t*)ctx->d_pmeta.p + 4 * (size_t)seed, 16, hipMemcpyDeviceToHost));
    if (n_words) *n_words = ctx->W;
    if (count) *count = meta[0];
    if (start) *start = meta[1];
    if (end) *end = meta[2];
    return DP_OK;
}
extern "C" int dp_index_seedset_row(dp_ctx* ctx, uint32_t seq, u64* words, uint32_t cap_words, uint32_t* n_words) {
    if (!ctx || seq >= ctx->n_seqs || cap_words < ctx->SW) return DP_ERR_ARG;
    hipSetDevice(ctx->device);
    DP_HIP(hipMemcpy(words, (u64*)ctx->d_seedsets.p + (uint64_t)seq * ctx->SW, (size_t)ctx->SW * 8, hipMemcpyDeviceToHost));
    if (n_words) *n_words = ctx->SW;
    return DP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// queries: seed bitsets of the queries (matchWorker's seedSet, overlap/overlap.go:351-354)

// ---------------------------------------------------------------------------------------------------------------
// A14 + A5: SeedIndex.Matches -> util.GetSharedIDs

// exclusive scans of qcnt[q] (pairs) and qcnt[q] x seeds of query q (scratch ints) by T threads of one workgroup; qcnt was written by
// the other workgroups' atomics: read through the L2
template <int T>
__device__ __forceinline__ void pair_scan_body(const uint32_t* __restrict__ qcnt, const u64* __restrict__ qoff, uint32_t nq,
                                               uint32_t* __restrict__ pbase, u64* __restrict__ ibase, u64* __restrict__ totals,
                                               u64* shp, u64* shi) {
    const uint32_t per = (nq + T - 1) / T;
    const uint32_t lo = min(nq, threadIdx.x * per), hi = min(nq, lo + per);
    unsigned long long sp = 0, si = 0;
    for (uint32_t q = lo; q < hi; q++) {
        const uint32_t c = __hip_atomic_load(&qcnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sp += c;
        si += (unsigned long long)c * ((qoff[q + 1] - qoff[q]) / 2);
    }
    shp[threadIdx.x] = sp;
    shi[threadIdx.x] = si;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {
        unsigned long long ap = 0, ai = 0;
        if ((int)threadIdx.x >= d) {
            ap = shp[threadIdx.x - d];
            ai = shi[threadIdx.x - d];
        }
        __syncthreads();
        shp[threadIdx.x] += ap;
        shi[threadIdx.x] += ai;
        __syncthreads();
    }
    unsigned long long bp = shp[threadIdx.x] - sp, bi = shi[threadIdx.x] - si;
    for (uint32_t q = lo; q < hi; q++) {
        const uint32_t c = __hip_atomic_load(&qcnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pbase[q] = (uint32_t)bp;
        ibase[q] = bi;
        bp += c;
        bi += (unsigned long long)c * ((qoff[q + 1] - qoff[q]) / 2);
    }
    if (threadIdx.x == T - 1) {
        pbase[nq] = (uint32_t)shp[T - 1];
        ibase[nq] = shi[T - 1];
        totals[0] = shp[T - 1];
        totals[1] = shi[T - 1];
    }
}

#define Q_MAXSETS 512
#define Q_WAVES 8

struct QWave {
    uint32_t setid[Q_MAXSETS];
    uint32_t lens[Q_MAXSETS];
    uint32_t tmp_id[Q_MAXSETS];   // working copy for the order simulation
    uint32_t tmp_len[Q_MAXSETS];
    uint32_t ev_word[Q_MAXSETS + 2];
    uint16_t ev_first8[Q_MAXSETS + 2][8];
    uint32_t n_ev;
};
// Round 6: the same lists in global memory, for a query with more usable seeds than QWave holds (Matches() has no limit,
// seeds/seeds.go:336-347: `-overlap_size 6000 -num_seeds 60`, `map -query_size 8000 -seed_rate 10` are ~900 sets a query).  The BIG
// variants of the kernel run behind the ordinary ones, only when a query of the batch has more than Q_MAXSETS seeds at all, and only
// for the queries the ordinary ones flagged; a query's slice holds `stride` sets (the batch's longest query).  Ids travel in 16 bits
// (ev_first8): 65 535 sets per query.
struct QBig {
    uint32_t* setid;
    uint32_t* lens;
    uint32_t* tmp_id;
    uint32_t* tmp_len;
    uint32_t* ev_word;
    uint16_t (*ev_first8)[8];
    uint32_t n_ev;
};
#define Q_BIG_WORDS_PER_SET 9u  // 4 + 4 + 4 + 4 + 4 + 16 bytes per set of a slice, as 4-byte words
#define Q_BIG_MAXSETS 65535u

// One word of the soft union for the 4- / 8-ladder regimes: the ladder of util/asm_amd64.s:121-314 over this lane's word of
// every live posting set, D levels deep.  Per posting word and level the reference does v_j |= v_{j-1} & m: one v_and_or_b32 per
// 32-bit half here (the compiler emits an AND and an OR for the 64-bit form - 30 VALU instructions per posting word against 2 * D;
// at 3.4 TB/s of posting words the kernel was as much VALU- as HBM-bound).
__device__ __forceinline__ uint32_t q_and_or(uint32_t a, uint32_t b, uint32_t c) {  // (a & b) | c
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#ifndef Q_INFLIGHT
#define Q_INFLIGHT 8  // posting words a lane has in flight (16 measured the same: 49.2 against 48.7 us)
#endif
template <int D, class ST>
__device__ __forceinline__ u64 q_ladder(const ST& S, const u64* __restrict__ posting, uint32_t W, uint32_t iw, uint32_t n, u64& gathered) {
    uint32_t lo[D], hi[D];
#pragma unroll
    for (int x = 0; x < D; x++) lo[x] = hi[x] = 0;
#define Q_STEP(m_)                                             \
    {                                                          \
        const uint32_t ml_ = (uint32_t)(m_), mh_ = (uint32_t)((m_) >> 32); \
        _Pragma("unroll") for (int x = D - 1; x >= 1; x--) {   \
            lo[x] = q_and_or(lo[x - 1], ml_, lo[x]);           \
            hi[x] = q_and_or(hi[x - 1], mh_, hi[x]);           \
        }                                                      \
        lo[0] |= ml_;                                          \
        hi[0] |= mh_;                                          \
    }
    // Q_INFLIGHT posting words in flight per lane: a set whose window ends before this word contributes 0, which leaves the ladder as
    // it is (the order of the words does not matter for the 4- and 8-ladders)
    uint32_t j = 0;
    for (; j + Q_INFLIGHT <= n; j += Q_INFLIGHT) {
        u64 m[Q_INFLIGHT];
#pragma unroll
        for (int u = 0; u < Q_INFLIGHT; u++) {
            const bool a = S.lens[j + u] > iw;
            m[u] = a ? posting[(uint64_t)S.setid[j + u] * W + iw] : 0ull;
            gathered += (u64)a;
        }
#pragma unroll
        for (int u = 0; u < Q_INFLIGHT; u++) Q_STEP(m[u])
    }
    for (; j + 8 <= n; j += 8) {
        u64 m[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool a = S.lens[j + u] > iw;
            m[u] = a ? posting[(uint64_t)S.setid[j + u] * W + iw] : 0ull;
            gathered += (u64)a;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) Q_STEP(m[u])
    }
    for (; j + 4 <= n; j += 4) {
        u64 m[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool a = S.lens[j + u] > iw;
            m[u] = a ? posting[(uint64_t)S.setid[j + u] * W + iw] : 0ull;
            gathered += (u64)a;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) Q_STEP(m[u])
    }
    for (; j < n; j++) {
        if (S.lens[j] <= iw) continue;
        const u64 m = posting[(uint64_t)S.setid[j] * W + iw];
        gathered++;
        Q_STEP(m)
    }
#undef Q_STEP
    return ((u64)hi[D - 1] << 32) | lo[D - 1];
}

// qmeta per query: {n_sets, minCount, status}; status bit0 = too many sets
// One WORKGROUP per query: wave 0 prepares the set list (Matches' filter, the early-return cut, the 16-ladder's gather
// order), then the Q_WAVES waves share the query's word range, 64 words per wave step, so that a dense index (W ~ 3 k words,
// k = 10) is streamed by thousands of waves instead of one per query.
// Two variants: the light one handles the queries of the 4- / 8-ladder regimes (minCount <= 12: every query of the overlap command
// at its defaults) and writes every query's qmeta / seed bitset; the heavy one only the 16-ladder and exact-count regimes
// (minCount >= 13: queries of fifty seeds and more), whose 17-level ladder and bit-sliced counter need twice the registers.  The
// light kernel alone then runs at eight waves per SIMD instead of four; the heavy one is launched only when a query of the batch
// can reach minCount 13 at all.
template <bool HEAVY, bool BIG = false>
struct query_kernel {
    enum { THREADS = 64 * Q_WAVES };
    typedef typename std::conditional<BIG, QBig, QWave>::type ST;
    static __device__ void run(const int32_t* __restrict__ qsegs, const u64* __restrict__ qoff, uint32_t nq, const u64* __restrict__ posting,
                               const uint32_t* __restrict__ pmeta, uint32_t n_seqs, uint32_t W, const int32_t* __restrict__ mc, uint32_t mc_n,
                               u64* __restrict__ cand, uint32_t* __restrict__ qmeta, u64* __restrict__ words_read, uint32_t* __restrict__ qcnt,
                               uint32_t word_base, const uint32_t* __restrict__ n_seqs_dev, u64* __restrict__ qsets, uint32_t SW,
                               uint32_t dbg_flags, uint32_t split, u64* __restrict__ own_zero, uint32_t* __restrict__ big_ws,
                               uint32_t big_stride) {
        // (the pair-offset scan as the last act of this launch's last workgroup - round 5's DP_QUERY_SCAN - cost a release fence per
        // workgroup, more than the launch it saved: profiles/r05/ab12_query_scan.txt; removed in round 6)
        if (blockIdx.x < nq * split)
            body(qsegs, qoff, nq, posting, pmeta, n_seqs, W, mc, mc_n, cand, qmeta, words_read, qcnt, word_base, n_seqs_dev, qsets, SW, dbg_flags, split, own_zero,
                 big_ws, big_stride);
    }
    static __device__ void body(const int32_t* __restrict__ qsegs, const u64* __restrict__ qoff,
                                                             uint32_t nq, const u64* __restrict__ posting,
                                                             const uint32_t* __restrict__ pmeta, uint32_t n_seqs, uint32_t W,
                                                             const int32_t* __restrict__ mc, uint32_t mc_n,
                                                             u64* __restrict__ cand, uint32_t* __restrict__ qmeta,
                                                             u64* __restrict__ words_read, uint32_t* __restrict__ qcnt,
                                                             uint32_t word_base, const uint32_t* __restrict__ n_seqs_dev,
                                                             u64* __restrict__ qsets, uint32_t SW, uint32_t dbg_flags, uint32_t split,
                                                             u64* __restrict__ own_zero, uint32_t* __restrict__ big_ws, uint32_t big_stride) {
    if (n_seqs_dev) n_seqs = *n_seqs_dev;  // (chunks made on the device: the host only knows an upper bound)
    // word_base: a shard of a larger index (dp_index_set_global) holds the words [word_base, word_base + W) of every set; pmeta
    // and n_seqs then describe the WHOLE sets (global windows, counts), so the filter, the early-return cut and the gather
    // order are those of the unsharded query, and this launch fills in the candidate words of its own range.
    __shared__ ST S;
    __shared__ uint32_t sh_u[8];
    __shared__ int64_t sh_ilast;
    __shared__ unsigned long long sh_gathered;
    const int lane = dp_lane();
    const int wave = threadIdx.x >> 6;
    // `split` workgroups per query can share its word range (64-word pieces dealt round robin; DP_QUERY_SPLIT).  Measured on the
    // dense regime (W ~ 1.5 k words, 25 sets per query, 668 queries): 1 -> 62.8 us, 2 -> 63.3, 3 -> 64.6, 4 -> 66.0, 8 -> 106: the
    // kernel is not held back by the 668 workgroups' fit on 256 CUs, so one workgroup per query stays the default
    const uint32_t q = blockIdx.x / split, part = blockIdx.x % split;
    if (q >= nq) return;
    const uint32_t set_cap = BIG ? big_stride : (uint32_t)Q_MAXSETS;
    if (threadIdx.x == 0) {
        sh_gathered = 0;
        sh_u[5] = 0;
        if constexpr (BIG) {  // the query's slice of the workspace: five 4-byte arrays of big_stride + 2 entries, then the 16-byte ones
            uint32_t* base = big_ws + (size_t)blockIdx.x * ((size_t)big_stride + 2) * Q_BIG_WORDS_PER_SET;
            S.setid = base;
            S.lens = base + (big_stride + 2);
            S.tmp_id = base + 2 * (size_t)(big_stride + 2);
            S.tmp_len = base + 3 * (size_t)(big_stride + 2);
            S.ev_word = base + 4 * (size_t)(big_stride + 2);
            S.ev_first8 = (uint16_t(*)[8])(base + 5 * (size_t)(big_stride + 2));
            S.n_ev = 0;
        }
    }
    if constexpr (BIG) __syncthreads();
    // own_zero (light variant, one workgroup per query): the query's rows - candidate words, seed bitset, the per-query words - are
    // cleared here, by the workgroup that owns them, instead of by a launch over all of them before this one; own_zero itself is
    // the chaining stage's cursor block (C_CURSOR_BYTES), cleared by query 0
    if (!HEAVY && own_zero) {
        for (uint32_t w = threadIdx.x; w < W; w += THREADS) cand[(uint64_t)q * W + w] = 0ull;
        if (qsets)
            for (uint32_t w = threadIdx.x; w < SW; w += THREADS) qsets[(uint64_t)q * SW + w] = 0ull;
        if (threadIdx.x < 4) qmeta[4 * q + threadIdx.x] = 0u;
        if (threadIdx.x == 4) words_read[q] = 0ull;
        if (threadIdx.x == 5) qcnt[q] = 0u;
        if (q == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + C_CURSOR_BYTES / 8) own_zero[threadIdx.x - 64] = 0ull;
        __syncthreads();
    }
    const int32_t* seg = qsegs + qoff[q];
    const uint32_t ns = (uint32_t)((qoff[q + 1] - qoff[q]) / 2);
    // the query's own seed set (what the chaining stage's prefilter intersects with the targets' sets); its row was zeroed with
    // the other per-round buffers.  Waves 1.. do it while wave 0 prepares the set list.
    if (!HEAVY && qsets && part == 0 && (wave != 0 || Q_WAVES == 1))
        for (uint32_t i = threadIdx.x - (Q_WAVES == 1 ? 0 : 64); i < ns; i += 64 * (Q_WAVES == 1 ? 1 : Q_WAVES - 1)) {
            const uint32_t seed = (uint32_t)seg[2 * i + 1];
            atomicOr(&qsets[(uint64_t)q * SW + (seed >> 6)], 1ull << (seed & 63));
        }
    if (wave == 0) {

    // --- Matches(): filtered list of posting sets (seeds/seeds.go:336-347).  The reference walks the query's seeds in order,
    // accepts a seed that differs from the last ACCEPTED one and whose set does not hold every sequence.  A seed that passes the
    // second test is rejected only when it equals the last accepted seed, so the last accepted seed always equals the previous
    // seed that passed the second test: "differs from the previous passing seed" decides, which 64 lanes test at once (with 150+
    // seeds per query in the dense regime the one-lane walk and its dependent loads were most of this kernel's time).
    uint32_t n = 0, start = 0xffffffffu, end = 0, status = 0;
    {
        const u64 lanesBelow = (1ull << lane) - 1ull;
        int32_t carrySeed = -1;
        for (uint32_t base = 0; base < ns; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool valid = i < ns;
            const int32_t seed = valid ? seg[2 * i + 1] : -1;
            uint4 pm = make_uint4(0xffffffffu, 0, 0, 0);
            if (valid) pm = *(const uint4*)(pmeta + 4 * (size_t)(uint32_t)seed);
            const bool f = valid && pm.x < n_seqs;
            const u64 fmask = __ballot(f);
            const u64 below = fmask & lanesBelow;
            const int src = below ? 63 - __builtin_clzll(below) : 0;
            const int32_t fromLane = __shfl(seed, src, 64);
            const int32_t prevPassing = below ? fromLane : carrySeed;
            const bool acc = f && seed != prevPassing;
            const u64 amask = __ballot(acc);
            const uint32_t pos = n + (uint32_t)__popcll(amask & lanesBelow);
            if (acc && pos < set_cap) {
                S.setid[pos] = (uint32_t)seed;
                S.lens[pos] = pm.w;
                start = min(start, pm.y);
                end = max(end, pm.z);
            }
            const uint32_t cnt = (uint32_t)__popcll(amask);
            if (n + cnt > set_cap) status |= 1;  // (ordinary variants: the BIG ones take this query; BIG: more than 65 535 sets)
            n = min(set_cap, n + cnt);
            if (fmask) carrySeed = __shfl(seed, 63 - __builtin_clzll(fmask), 64);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            start = min(start, (uint32_t)__shfl_xor((int)start, d, 64));
            end = max(end, (uint32_t)__shfl_xor((int)end, d, 64));
        }
        __builtin_amdgcn_wave_barrier();
    }
    int minCount = 0;
    if (n >= 5 && n < mc_n) minCount = mc[n];
    if constexpr (BIG) {
        if (n <= (uint32_t)Q_MAXSETS) status |= 4u;  // (the ordinary variants' query: nothing to do here)
        __threadfence();                             // (the lists are global memory: the other waves read them behind the barrier)
    }
    if (!HEAVY && lane == 0 && part == 0 && !(BIG && (status & 4u))) {
        qmeta[4 * q + 0] = n;
        qmeta[4 * q + 1] = (uint32_t)minCount;
        qmeta[4 * q + 2] = status | (n >= mc_n ? 2u : 0u);
        qmeta[4 * q + 3] = ns;
    }
    if (lane == 0) {
        sh_u[0] = n;
        sh_u[1] = start;
        sh_u[2] = end;
        sh_u[3] = status;
        sh_u[4] = (uint32_t)minCount;
        sh_ilast = end;
    }
    if (!(n < 5 || status || n >= mc_n)) {
    __builtin_amdgcn_wave_barrier();

    // --- early return of GetSharedIDs (util/bitset.go:335-342): the (n-minCount+1)-th drop ends the scan
    //     before its word is gathered.  d-th drop happens at word max(start, d-th smallest lens).
    int64_t i_last = end;  // last word index that is processed
    {
        const int dstar = (int)n - minCount + 1;  // >= 1
        // d-th smallest of lens: rank selection with all lanes
        uint32_t cutLens = 0xffffffffu;
        for (uint32_t j = lane; j < n; j += 64) {
            uint32_t lj = S.lens[j];
            int less = 0, eq_before = 0;
            for (uint32_t t = 0; t < n; t++) {
                uint32_t lt = S.lens[t];
                less += lt < lj;
                eq_before += (lt == lj) && (t < j);
            }
            if (less + eq_before == dstar - 1) cutLens = lj;  // this element has rank dstar (1-based)
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cutLens = min(cutLens, (uint32_t)__shfl_xor((int)cutLens, d, 64));
        if (cutLens != 0xffffffffu) {
            int64_t icut = max((int64_t)start, (int64_t)cutLens);
            if (icut - 1 < i_last) i_last = icut - 1;
        }
    }

    // --- for the 16-ladder the order of the gathered words matters (8th word never reaches v1): simulate the
    //     swap-removals of util/bitset.go:333-353 and record the first eight sets after every event.
    const bool ladder16 = minCount >= 13;
    if (HEAVY && ladder16) {
        if (lane == 0) {
            uint32_t cn = n;
            for (uint32_t j = 0; j < n; j++) {
                S.tmp_id[j] = j;
                S.tmp_len[j] = S.lens[j];
            }
            uint32_t shortest = 0xffffffffu;
            for (uint32_t j = 0; j < n; j++) shortest = min(shortest, S.tmp_len[j]);
            uint32_t nev = 0;
            // entry 0: order valid from `start` (after the drops performed at i == start, if any)
            int64_t i = start;
            bool first = true;
            while (i <= i_last) {
                if (shortest <= (uint32_t)i) {
                    uint32_t nextShortest = end;
                    for (uint32_t j = 0; j < cn; j++) {
                        if (S.tmp_len[j] <= (uint32_t)i) {
                            uint32_t last = cn - 1;
                            S.tmp_id[j] = S.tmp_id[last];
                            S.tmp_len[j] = S.tmp_len[last];
                            cn = last;
                            j--;
                        } else if (S.tmp_len[j] < nextShortest) {
                            nextShortest = S.tmp_len[j];
                        }
                    }
                    shortest = nextShortest;
                    first = true;
                }
                if (first) {
                    S.ev_word[nev] = (uint32_t)i;
                    for (int t = 0; t < 8; t++) S.ev_first8[nev][t] = (uint16_t)S.tmp_id[t];
                    nev++;
                    first = false;
                }
                // jump to the next word at which something is dropped
                int64_t nxt = (int64_t)shortest;
                if (nxt <= i) nxt = i + 1;
                i = nxt;
            }
            S.n_ev = nev;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) sh_ilast = i_last;
    }  // usable query
    }  // wave 0
    __syncthreads();
    const uint32_t n = sh_u[0], start = sh_u[1], status = sh_u[3];
    const int minCount = (int)sh_u[4];
    const int64_t i_last = sh_ilast;
    if (n < 5 || status || n >= mc_n) return;  // cand row stays zero
    if (dbg_flags & 1u) return;  // DP_QUERY_DEBUG=1 (timing experiments): the set-up alone, no posting word is read
    const bool ladder16 = minCount >= 13;
    if (ladder16 != HEAVY) return;  // (the other variant's query)
    const uint32_t n_ev = ladder16 ? S.n_ev : 0;
    const bool exact = minCount > 24;  // fast=false (util/bitset.go:309-311)

    u64 gathered = 0;
    int nCand = 0;  // Matches() result size of this query (bits set in its cand row)
    const int64_t w_lo = max((int64_t)start, (int64_t)word_base), w_hi = min(i_last, (int64_t)word_base + (int64_t)W - 1);
    posting -= word_base;  // (indexed by the global word below)
    for (int64_t ib = w_lo + 64 * (int64_t)(wave * split + part); ib <= w_hi; ib += 64 * (int64_t)(Q_WAVES * split)) {
        const int64_t i = ib + lane;
        if (i > w_hi) continue;
        const uint32_t iw = (uint32_t)i;
        u64 v = 0;
        if constexpr (!HEAVY) {
            // 4- and 8-ladders (asm:121-314) are order independent: v_t = bits present in >= t of the live sets; only the levels up
            // to the query's minCount are kept (level t depends on the levels below it alone)
            switch (minCount) {
                case 0:
                case 1: v = q_ladder<1>(S, posting, W, iw, n, gathered); break;
                case 2: v = q_ladder<2>(S, posting, W, iw, n, gathered); break;
                case 3: v = q_ladder<3>(S, posting, W, iw, n, gathered); break;
                case 4: v = q_ladder<4>(S, posting, W, iw, n, gathered); break;
                case 5: v = q_ladder<5>(S, posting, W, iw, n, gathered); break;
                case 6: v = q_ladder<6>(S, posting, W, iw, n, gathered); break;
                case 7: v = q_ladder<7>(S, posting, W, iw, n, gathered); break;
                default: v = q_ladder<8>(S, posting, W, iw, n, gathered); break;  // 8, and 9..12 saturate at 8 (bitset.go:369-372)
            }
        } else {
            // 16-ladder (asm:317-509) in the exact gather order
            uint32_t t = 0;
            while (t + 1 < n_ev && S.ev_word[t + 1] <= iw) t++;
            u64 L[17];
#pragma unroll
            for (int x = 0; x < 17; x++) L[x] = 0;
            u64 inFirst8[BIG ? 1 : Q_MAXSETS / 64];  // (BIG: the eight ids themselves are compared below)
            uint32_t f8[8];
#pragma unroll
            for (int x = 0; x < (BIG ? 1 : Q_MAXSETS / 64); x++) inFirst8[x] = 0;
#pragma unroll
            for (int p = 0; p < 8; p++) {
                uint32_t j = S.ev_first8[t][p];
                f8[p] = j;
                if constexpr (!BIG) inFirst8[j >> 6] |= 1ull << (j & 63);
                u64 m = posting[(uint64_t)S.setid[j] * W + iw];
                gathered++;
#pragma unroll
                for (int x = 16; x >= 2; x--) L[x] |= L[x - 1] & m;
                if (p != 7) L[1] |= m;  // step 8 omits "ORQ DX, R8" (asm:407-428)
            }
            // (a count over at most Q_MAXSETS = 512 live sets needs ten bits: with eight - until round 5 - a sequence that holds more than
            // 255 of a long query's seeds, i.e. its best candidates, wrapped around and was dropped; found by the flag matrix with
            // overlap_size 2000, num_seeds 30, min_hits 0.4: 315 sets, minCount 126)
            enum { Q_PLANES = BIG ? 16 : 10 };
            static_assert((1 << Q_PLANES) > (BIG ? (int)Q_BIG_MAXSETS : Q_MAXSETS), "the exact count of addSoftUnionIDs must hold the set list");
            u64 planes[Q_PLANES];
#pragma unroll
            for (int x = 0; x < Q_PLANES; x++) planes[x] = 0;
            for (uint32_t j = 0; j < n; j++) {
                if (S.lens[j] <= iw) continue;
                if constexpr (BIG) {
                    if (j == f8[0] || j == f8[1] || j == f8[2] || j == f8[3] || j == f8[4] || j == f8[5] || j == f8[6] || j == f8[7]) continue;
                } else {
                    if ((inFirst8[j >> 6] >> (j & 63)) & 1) continue;
                }
                u64 m = posting[(uint64_t)S.setid[j] * W + iw];
                gathered++;
#pragma unroll
                for (int x = 16; x >= 2; x--) L[x] |= L[x - 1] & m;
                L[1] |= m;
            }
            v = minCount >= 16 ? L[16] : minCount == 15 ? L[15] : minCount == 14 ? L[14] : L[13];
            if (exact && v) {
                // addSoftUnionIDs (bitset.go:509-538): keep a bit only if its true count over the live sets
                // reaches minCount.  Bit-sliced counter, Q_PLANES planes.
                for (uint32_t j = 0; j < n; j++) {
                    if (S.lens[j] <= iw) continue;
                    u64 c = posting[(uint64_t)S.setid[j] * W + iw];
#pragma unroll
                    for (int x = 0; x < Q_PLANES; x++) {
                        u64 nc = planes[x] & c;
                        planes[x] ^= c;
                        c = nc;
                    }
                }
                // ge = (count >= minCount), most significant plane first
                u64 gt = 0, eq = ~0ull;
#pragma unroll
                for (int x = Q_PLANES - 1; x >= 0; x--) {
                    u64 bit = ((minCount >> x) & 1) ? ~0ull : 0ull;
                    gt |= eq & planes[x] & ~bit;
                    eq &= ~(planes[x] ^ bit);
                }
                v &= (gt | eq);
            }
        }
        cand[(uint64_t)q * W + (iw - word_base)] = v;
        nCand += __popcll(v);
    }
    gathered = (u64)wave_sum((int)gathered);
    nCand = wave_sum(nCand);
    if (lane == 0) {
        atomicAdd(&sh_gathered, (unsigned long long)gathered);
        atomicAdd(&sh_u[5], (uint32_t)nCand);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (both were zeroed with the other per-round buffers; the query's workgroups add their shares)
        if (sh_gathered) atomicAdd((unsigned long long*)&words_read[q], sh_gathered);
        if (sh_u[5]) atomicAdd(&qcnt[q], sh_u[5]);
    }
}
};

// ---------------------------------------------------------------------------------------------------------------
// A6 + A7 + A8: matchWorker body

// Profiling build (make PROF=1): per-wave phase timers of the chaining kernels (DP_CHAIN_PROF=1 prints them).  In the normal
// build the timers are compiled out - their sums live in a per-thread array that ends up in scratch memory.
#ifdef DP_PROF_BUILD
#define DP_PROFILING 1
#else
#define DP_PROFILING 0
#endif
struct ChainProf {  // ticks of 10 ns, summed per wave (profiling build)
    unsigned long long pairs, chained, rec, stagea, pre, pair;   // chain_spec_kernel's phases
    unsigned long long stageb, initial, walk, out, last;          // chain_pair's
    unsigned long long bev, nev, nperfect, why;  // why: 5 x 12-bit counts of why a pair was not a perfect chain          // wave_chain_reg's
};
#define C_WAVES 4
#define S_WAVES 4  // waves per workgroup of the kernels on the slim layout
#define C_OPEN 500       // len(align.open)          seeds/alignment.go:299
#define C_RESULTS 500    // len(align.results)
#define C_POOLSTATES 10000
#define C_NODES (1u << 16)

// CNodes of a wave's slice of the pool: its chain links, and - for an aligner whose reduced buffer (maxLength ints + the index map,
// maxLength / 2) is larger than the LDS copy - room for both behind them
static inline uint32_t chain_pool_stride(uint32_t max_query_len) {
    return C_NODES + (max_query_len > 510 ? (uint32_t)(((size_t)max_query_len + max_query_len / 2 + 8) * 4 + 7) / 8 : 0u);
}

struct CNode {  // one link of a chain (pairState.prev history), written once
    uint16_t a, b;
    int32_t prev;
};

#define C_ACAP 512       // query segments staged in LDS (ints)
#define C_BCAP 1056      // target segments staged in LDS (ints); larger targets take the one-lane path
#define C_LNODES 512     // chain links kept in LDS before spilling to the global pool
#define C_REV 256        // b events of the reg tier
#define C_QSW 256        // query bitset words staged in LDS

struct CWave {
    typedef uint32_t col_t;
    enum { COLN = 64, EVN = C_REV, ACAP = C_ACAP, RSEEDS = 255, SLIM = 0, ROWS = 64, BCAP = C_BCAP };
    int32_t aRed[512];
    int32_t aMap[256];
    int32_t aSegL[C_ACAP];
    int32_t bSegL[C_BCAP];
    u64 aFlag[C_ACAP / 128];      // bit i: seed i of a is in bSet
    u64 bFlag[C_BCAP / 128 + 1];  // bit j: seed j of b is in aSet
    u64 qsetL[C_QSW];             // the query's seed bitset when it fits
    union {
        struct {  // lds / one-lane tiers
            int32_t o_aPos[C_OPEN], o_bPos[C_OPEN], o_aGap[C_OPEN], o_bGap[C_OPEN], o_aGapIndex[C_OPEN], o_len[C_OPEN],
                o_node[C_OPEN];
            int32_t evOff[C_BCAP / 2];   // b seeds that reach the chain walk: accumulated bOffset ...
            uint16_t evIdx[C_BCAP / 2];  // ... and bIndex
            CNode lnodes[C_LNODES];
        };
        struct {  // reg tier
            col_t col[64][COLN];  // matched pairs of open chain i: reducedA | bSeedIndex << 6
            int4 ev[C_REV];
            int32_t ps[64];
            uint32_t rescol[64];
        };
    };
};

// The reg tier's working set alone, a quarter of CWave: the proposal pass (chain_spec_kernel) runs one wave per PAIR and is
// bound by how many waves a CU holds.  Pairs that do not fit (query > 128 seeds, reduced a > 64, > 128 b events, more than
// 32 open chains, target > C_BCAP ints) get no proposal here: the later passes and the final walk have the full CWave.
struct CSlim {
    typedef uint16_t col_t;
    // (round 6: 32 open chains of up to 64 links instead of 64 of up to 32 - the same 4 KB.  In the dense regime (k = 10) a query
    // window holds 40-60 seeds and its real overlaps are chains of 33-60 links with two or three chains open: ~190 pairs a round did
    // not fit, their queries stopped at them in every pass and the final walk chained those queries' tails one pair after another -
    // 400 of a round's 670 us of chaining, profiles/r06/k10_chain_phases_slots1.txt)
    enum { COLN = 64, EVN = 128, ACAP = 256, RSEEDS = 64, SLIM = 1, ROWS = 32, BCAP = C_BCAP };
    int32_t aRed[2 * 64 + 2];
    int32_t aMap[64];
    int32_t aSegL[256];
    u64 aFlag[2];
    u64 bFlag[C_BCAP / 128 + 1];
    union {
        int32_t bSegL[C_BCAP];  // dead once the b events are built
        col_t col[ROWS][COLN];
    };
    int4 ev[EVN];
    int32_t ps[64];
    uint32_t rescol[64];
};

__device__ __forceinline__ void node_put(CWave& L, CNode* __restrict__ nodes, int idx, CNode nd) {
    if (idx < C_LNODES)
        L.lnodes[idx] = nd;
    else
        nodes[idx] = nd;
}
__device__ __forceinline__ CNode node_get(const CWave& L, const CNode* __restrict__ nodes, int idx) {
    return idx < C_LNODES ? L.lnodes[idx] : nodes[idx];
}

__device__ __forceinline__ bool bs_contains(const u64* __restrict__ set, int32_t x) { return (set[x >> 6] >> (x & 63)) & 1ull; }

// seeds/alignment.go:411-424; written with selects only (reference parameters made the compiler keep minGap/maxGap
// in scratch memory inside the chain loops)
#define gap_range(gap_, k_, mn_, mx_)                                        \
    {                                                                        \
        const int g__ = (gap_);                                              \
        const int m0__ = (g__ * 2) / 3 - (k_);                               \
        const int x0__ = (g__ * 3) / 2 + (k_) + 1;                           \
        const bool neg__ = m0__ < 0;                                         \
        const bool small__ = !neg__ && x0__ < 20;                            \
        mx_ = neg__ ? (x0__ < 0 ? 0 : x0__) : (small__ ? 20 : x0__);         \
        mn_ = neg__ ? -(k_) : (small__ ? 0 : m0__);                          \
    }

// seedAligner.PairwiseAlignments (seeds/alignment.go:426-616), executed by ONE lane.  Returns the length of
// results[0] (the chain matchWorker keeps, overlap/overlap.go:368-375) or 0; *resNode = its last node.
// err: 1 reduced buffer overflow, 2 state pool, 4 results overflow, 8 node pool
// aFlag/bFlag (LDS, may be null): membership bits precomputed by the whole wave, replacing the bitset probes.
__device__ int pairwise_align(const int32_t* aSeg, int aN, const int32_t* bSeg, int bN, const u64* __restrict__ aSet,
                              const u64* __restrict__ bSet, const u64* aFlag, const u64* bFlag, int minMatches, int k,
                              int maxLength, CWave& L, CNode* __restrict__ nodes, int* resNode, uint32_t* err, int32_t* spill = nullptr) {
    if (minMatches == 0) minMatches = 1;
    // reduced a and its index map: in LDS while the reference's own capacity (maxLength ints = overlap_size / 2, seeds/alignment.go:
    // 298-302) fits there; a command run with -overlap_size beyond 1024 gets the wave's spill area behind its node pool instead, so
    // that the only "reduced buffer" limit is the reference's
    int32_t* const aRed = spill ? spill : L.aRed;
    int32_t* const aMap = spill ? spill + maxLength + 2 : L.aMap;
    const int redCap = spill ? maxLength + 2 : 512;
    int nNodes = 0;
    int live = 0;
    // prepareInitial :341-388
    int maxAIndex = aN - minMatches * 2 + 1;
    int aLen = 0, offset = -k, startSize = 0, prevSeedA = -1;
    for (int i = 1; i < aN; i += 2) {
        int aSeed = aSeg[i];
        const bool inB = aFlag ? ((aFlag[i >> 7] >> ((i >> 1) & 63)) & 1ull) : bs_contains(bSet, aSeed);
        if (!inB) {
            offset += aSeg[i - 1] + k;
            maxAIndex--;
            continue;
        }
        if (aSeed == prevSeedA && (i >= aN - 2 || aSeg[i + 2] == prevSeedA)) {
            offset += aSeg[i - 1] + k;
            maxAIndex--;
            continue;
        }
        prevSeedA = aSeed;
        offset += aSeg[i - 1] + k;
        if (aLen * 2 + 1 >= maxLength || aLen >= maxLength / 2 || aLen * 2 + 2 >= redCap) {
            *err |= 1;
            return 0;
        }
        aRed[aLen * 2] = offset;
        aRed[aLen * 2 + 1] = aSeed;
        aMap[aLen] = i / 2;
        offset = -k;
        if (aLen <= maxAIndex) {
            startSize++;
            live++;
        }
        aLen++;
    }
    if (aLen * 2 >= maxLength) {
        *err |= 1;
        return 0;
    }
    aRed[aLen * 2] = 0;
    while (startSize > 0 && (2 * (startSize - 1) + 1) > maxAIndex) {
        startSize--;
        live--;
    }
    const int initialSize = startSize;
    const int aRedLen = aLen * 2 + 1;
    int openSize = 0, resultsSize = 0, firstLen = 0, firstNode = -1;

#define REMOVE_OPEN(i_)                                                            \
    {                                                                              \
        int sl_ = L.o_len[i_], sn_ = L.o_node[i_];                                 \
        int last_ = openSize - 1;                                                  \
        L.o_aPos[i_] = L.o_aPos[last_];                                            \
        L.o_bPos[i_] = L.o_bPos[last_];                                            \
        L.o_aGap[i_] = L.o_aGap[last_];                                            \
        L.o_bGap[i_] = L.o_bGap[last_];                                            \
        L.o_aGapIndex[i_] = L.o_aGapIndex[last_];                                  \
        L.o_len[i_] = L.o_len[last_];                                              \
        L.o_node[i_] = L.o_node[last_];                                            \
        openSize--;                                                                \
        if (sl_ >= minMatches) {                                                   \
            if ((sl_ * 2) / 3 > minMatches) minMatches = (sl_ * 2) / 3;            \
            if (resultsSize >= C_RESULTS) {                                        \
                *err |= 4;                                                         \
                return 0;                                                          \
            }                                                                      \
            if (resultsSize == 0) {                                                \
                firstLen = sl_;                                                    \
                firstNode = sn_;                                                   \
            }                                                                      \
            resultsSize++;                                                         \
        } else {                                                                   \
            live -= sl_;                                                           \
        }                                                                          \
    }

    int maxBIndex = bN - minMatches * 2 + 1;
    int bOffset = 0, prevSeed = -1;
    for (int bIndex = 1; bIndex < bN; bIndex += 2) {
        const int bSeed = bSeg[bIndex];
        const bool inA = bFlag ? ((bFlag[bIndex >> 7] >> ((bIndex >> 1) & 63)) & 1ull) : bs_contains(aSet, bSeed);
        if (!inA) {
            bOffset += bSeg[bIndex + 1] + k;
            continue;
        }
        if (bSeed == prevSeed && (bIndex >= bN - 2 || bSeg[bIndex + 2] == prevSeed)) {
            bOffset += bSeg[bIndex + 1] + k;
            continue;
        }
        prevSeed = bSeed;
        int found = -1;
        for (int i = openSize - 1; i >= 0; i--) {  // searchMatch :465-547
            int bGap = L.o_bGap[i] + bOffset;
            L.o_bGap[i] = bGap;
            int minGap, maxGap;
            gap_range(bGap, k, minGap, maxGap);
            int aGap = L.o_aGap[i], aGapIndex = L.o_aGapIndex[i];
            bool ended = false;
            while (aGap < minGap) {
                if (aGapIndex >= aRedLen) {
                    ended = true;
                    break;
                }
                aGap += aRed[aGapIndex + 1] + k;
                aGapIndex += 2;
            }
            L.o_aGap[i] = aGap;
            L.o_aGapIndex[i] = aGapIndex;
            if (ended) {
                REMOVE_OPEN(i);
                break;
            }
            bool extended = false;
            if (aGap <= maxGap) {
                int g = aGap;
                for (int j = aGapIndex; j < aRedLen && g <= maxGap; j += 2) {
                    if (aRed[j] == bSeed) {
                        found = j;
                        if (nNodes >= (int)C_NODES) {
                            *err |= 8;
                            return 0;
                        }
                        if (++live > C_POOLSTATES) {
                            *err |= 2;
                            return 0;
                        }
                        CNode nd;
                        nd.a = (uint16_t)aMap[j / 2];
                        nd.b = (uint16_t)(bIndex / 2);
                        nd.prev = L.o_node[i];
                        node_put(L, nodes, nNodes, nd);
                        const int nl = L.o_len[i] + 1;
                        L.o_aPos[i] = j;
                        L.o_bPos[i] = bIndex;
                        L.o_aGapIndex[i] = j + 2;
                        L.o_aGap[i] = aRed[j + 1];
                        L.o_bGap[i] = bSeg[bIndex + 1];
                        L.o_len[i] = nl;
                        L.o_node[i] = nNodes++;
                        if ((nl * 2) / 3 > minMatches) {
                            minMatches = (nl * 2) / 3;
                            maxBIndex = bN - minMatches * 2 + 1;
                        }
                        extended = true;
                        break;
                    }
                    g += aRed[j + 1] + k;
                }
            }
            if (extended) break;
            if (L.o_len[i] + (bN - bIndex) < minMatches) {
                REMOVE_OPEN(i);
            } else {
                L.o_bGap[i] += bSeg[bIndex + 1] + k;
            }
        }
        bOffset = 0;
        if (bIndex <= maxBIndex) {  // :550-587
            for (int i = 0; i < initialSize; i++) {
                const int aPos = 2 * i + 1;
                if (aPos != found && aRed[aPos] == bSeed) {
                    if (found != -1) {
                        for (int j = 0; j < openSize; j++) {
                            if (L.o_bPos[j] == bIndex && L.o_aPos[j] == aPos) {
                                found = aPos;
                                break;
                            }
                        }
                    }
                    if (found == aPos || openSize >= C_OPEN) continue;
                    if (nNodes >= (int)C_NODES) {
                        *err |= 8;
                        return 0;
                    }
                    if (++live > C_POOLSTATES) {
                        *err |= 2;
                        return 0;
                    }
                    CNode nd;
                    nd.a = (uint16_t)aMap[i];
                    nd.b = (uint16_t)(bIndex / 2);
                    nd.prev = -1;
                    node_put(L, nodes, nNodes, nd);
                    L.o_aPos[openSize] = aPos;
                    L.o_bPos[openSize] = bIndex;
                    L.o_aGapIndex[openSize] = aPos + 2;
                    L.o_aGap[openSize] = aRed[aPos + 1];
                    L.o_bGap[openSize] = bSeg[bIndex + 1];
                    L.o_len[openSize] = 1;
                    L.o_node[openSize] = nNodes++;
                    openSize++;
                }
            }
        }
    }
    for (int i = 0; i < openSize; i++) {  // :597-604
        if (L.o_len[i] >= minMatches) {
            if (resultsSize >= C_RESULTS) {
                *err |= 4;
                return 0;
            }
            if (resultsSize == 0) {
                firstLen = L.o_len[i];
                firstNode = L.o_node[i];
            }
            resultsSize++;
        }
    }
#undef REMOVE_OPEN
    *resNode = firstNode;
    return resultsSize ? firstLen : 0;
}

// ---- wave-cooperative PairwiseAlignments -----------------------------------------------------------------------
// a and b are staged in L.aSegL / L.bSegL with their membership bits; control flow is wave-uniform.  Three tiers share
// the parallel prepareInitial:
//   reg  (aLen <= 64, <= 256 b events, <= 64 open chains): chain i lives in the registers of lane i and keeps its
//        matched pairs in an LDS column, so an event costs a ballot plus register arithmetic;
//   lds  (anything staged): open chains in the LDS arrays, one lane per chain, history as linked nodes;
//   lane (operands too large to stage): the scalar transcription above.
// In both parallel tiers the reference's last-to-first walk over the open chains with its `break`s is restored after
// the lanes have evaluated their chains independently: the highest chain that ended or extended stops the walk,
// chains above it apply their updates/removals (swap-with-last, descending), chains below it stay untouched (stale
// bGap).  Chains evaluated in one step cannot influence each other: removals above the break never reach minMatches
// (length + remaining < minMatches), so minMatches is constant until the break itself.

// prepareInitial :341-388.  Returns 0; 1 when the reference would overrun `reduced` (err bit 1); 2 when the layout's staging arrays
// are too small for a reduced a the reference can hold (CSlim: the pair is left to the full-size path; CWave: to the one-lane
// transcription, whose reduced a may live in the wave's spill area).
template <class LW>
__device__ int wave_prepare_initial(int aN, int minMatches, int k, int maxLength, LW& L, int* aLenOut, int* startOut) {
    const int lane = dp_lane();
    const u64 lanesBelow = (1ull << lane) - 1ull;
    const int nA = aN >> 1;
    int aLen = 0, startSize = 0;
    const int C0 = aN - minMatches * 2 + 1;
    int prevSeed = -1, P = 0, PatKept = 0;
    bool bad = false, small = false;
    for (int base = 0; base < nA; base += 64) {
        const int s = base + lane;
        const bool valid = s < nA;
        const int seed = valid ? L.aSegL[2 * s + 1] : -2;
        const int gap = valid ? L.aSegL[2 * s] : 0;
        const bool inB = valid && ((L.aFlag[base >> 6] >> lane) & 1ull);
        const u64 inBmask = __ballot(inB);
        const u64 below = inBmask & lanesBelow;
        const int pl = below ? 63 - __builtin_clzll(below) : 0;
        int pseed = __shfl(seed, pl, 64);
        if (!below) pseed = prevSeed;
        const bool last = s >= nA - 1;
        const int nextRaw = (valid && !last) ? L.aSegL[2 * s + 3] : 0;
        const bool keep = inB && !(seed == pseed && (last || nextRaw == pseed));
        const u64 keepMask = __ballot(keep);
        const int Pin = P + wave_incl_sum(valid ? gap + k : 0);
        const u64 kb = keepMask & lanesBelow;
        const int kl = kb ? 63 - __builtin_clzll(kb) : 0;
        int Pk = __shfl(Pin, kl, 64);
        if (!kb) Pk = PatKept;
        const int rank = aLen + __popcll(kb);
        bool isStart = false;
        if (keep) {
            if (rank >= (int)LW::RSEEDS) {
                small = true;
            } else if (rank * 2 + 1 >= maxLength || rank >= maxLength / 2) {
                bad = true;
            } else {
                L.aRed[2 * rank] = Pin - Pk - k;
                L.aRed[2 * rank + 1] = seed;
                L.aMap[rank] = s;
                isStart = rank <= C0 - (s - rank);
            }
        }
        if (__ballot(small)) return 2;  // (the fallback finds the reference's own limit, if the pair hits it, by itself)
        if (__ballot(bad)) return 1;
        startSize += __popcll(__ballot(isStart));
        if (inBmask) prevSeed = __shfl(seed, 63 - __builtin_clzll(inBmask), 64);
        if (keepMask) PatKept = __shfl(Pin, 63 - __builtin_clzll(keepMask), 64);
        P = __shfl(Pin, 63, 64);
        aLen += __popcll(keepMask);
    }
    if (aLen * 2 >= maxLength) return 1;
    if (lane == 0) L.aRed[aLen * 2] = 0;
    const int maxAIndex = C0 - (nA - aLen);
    while (startSize > 0 && (2 * (startSize - 1) + 1) > maxAIndex) startSize--;
    *aLenOut = aLen;
    *startOut = startSize;
    return 0;
}

// b seeds that reach searchMatch (:449-457) with the bOffset accumulated since the previous one.  WIDE: 16-byte
// records {bIndex, bOffset, seed, gap after} for the reg tier, else evIdx/evOff.  Returns the number of events
// (records beyond `cap` are not stored).
template <bool WIDE, class LW>
__device__ int wave_b_events(int bN, int k, LW& L, int cap) {
    const int lane = dp_lane();
    const u64 lanesBelow = (1ull << lane) - 1ull;
    const int nB = bN >> 1;
    int nE = 0, prevSeed = -1, Q = 0, QatEvent = 0;
    for (int base = 0; base < nB; base += 64) {
        const int s = base + lane;
        const bool valid = s < nB;
        const int seed = valid ? L.bSegL[2 * s + 1] : -2;
        const int gapAfter = valid ? L.bSegL[2 * s + 2] : 0;
        const bool inA = valid && ((L.bFlag[base >> 6] >> lane) & 1ull);
        const u64 inAmask = __ballot(inA);
        const u64 below = inAmask & lanesBelow;
        const int pl = below ? 63 - __builtin_clzll(below) : 0;
        int pseed = __shfl(seed, pl, 64);
        if (!below) pseed = prevSeed;
        const bool last = s >= nB - 1;
        const int nextRaw = (valid && !last) ? L.bSegL[2 * s + 3] : 0;
        const bool ev = inA && !(seed == pseed && (last || nextRaw == pseed));
        const u64 evMask = __ballot(ev);
        const int Qin = Q + wave_incl_sum((valid && !ev) ? gapAfter + k : 0);
        const u64 eb = evMask & lanesBelow;
        const int el = eb ? 63 - __builtin_clzll(eb) : 0;
        int Qe = __shfl(Qin, el, 64);
        if (!eb) Qe = QatEvent;
        if (ev) {
            const int idx = nE + __popcll(eb);
            if (idx < cap) {
                if constexpr (WIDE) {
                    L.ev[idx] = make_int4(2 * s + 1, Qin - Qe, seed, gapAfter);
                } else {
                    L.evIdx[idx] = (uint16_t)(2 * s + 1);
                    L.evOff[idx] = Qin - Qe;
                }
            }
        }
        if (inAmask) prevSeed = __shfl(seed, 63 - __builtin_clzll(inAmask), 64);
        if (evMask) QatEvent = __shfl(Qin, 63 - __builtin_clzll(evMask), 64);
        Q = __shfl(Qin, 63, 64);
        nE += __popcll(evMask);
    }
    return nE;
}

#define RL(v_, l_) __builtin_amdgcn_readlane((v_), (l_))
#define RFL(v_) __builtin_amdgcn_readfirstlane(v_)

// reg tier.  Returns the length of results[0] (its pairs are left in L.rescol as reducedA | bSeed<<6), 0, or -1 when
// the pair needs the lds tier.
template <class LW>
__device__ int wave_chain_reg(int aLen, int startSize, int bN, int minMatches, int k, LW& L, uint32_t* err, ChainProf* cp,
                              bool walkAlways) {
    const bool prof = DP_PROFILING && cp != nullptr;
    const int lane = dp_lane();
    if (startSize == 0) return 0;  // no initial position: no chain can ever start
    int live = startSize;
    const int initialSize = startSize;
    const int aRedLen = aLen * 2 + 1;
    const unsigned long long tb0 = prof ? wall_clock64() : 0ull;
    const int nE = wave_b_events<true>(bN, k, L, (int)LW::EVN);
    if (nE > (int)LW::EVN) return -1;
    __builtin_amdgcn_wave_barrier();  // (CSlim: the columns below reuse b's staging area)
    if (prof) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        cp->bev += wall_clock64() - tb0;
        cp->nev += (u64)nE;
    }
    const int myA = lane < aLen ? L.aRed[2 * lane + 1] : -1;
    const int myOff = lane < aLen ? L.aRed[2 * lane] : 0;                // aRed[2 * lane]: the gap in front of kept seed `lane`
    L.ps[lane] = wave_incl_sum(lane < aLen ? myOff + k : 0);             // g(r) - g(r0) = ps[r] - ps[r0]
    int st_aPos = 0, st_bPos = 0, st_aGap = 0, st_bGap = 0, st_aGapIndex = 0, st_len = 0;
    int openSize = 0, resultsSize = 0, firstLen = 0;
    int maxBIndex = bN - minMatches * 2 + 1;
    const u64 initMask = initialSize >= 64 ? ~0ull : ((1ull << initialSize) - 1ull);

    // The perfect chain, decided for all events at once (lane e = event e).  An overlap is, nearly always, one chain that starts at
    // the first event and that every following event extends by the very next kept seed of a.  The walk below finds that in
    // nE dependent steps of a few hundred instructions each - and a CU's instruction issue, shared by its sixteen waves, is what
    // bounds this kernel.  Here: event 0 starts exactly one chain (one initial position r0 holds its seed); every event e >= 1 shows
    // the seed of position r0 + e at a distance inside the gap window (searchMatch's cursor then does not move: aGap >= minGap, the
    // hit is the cursor's own position, no cumulative-gap test) and matches no OTHER initial position (no second chain is ever
    // opened; tested for every event whether or not maxBIndex would still allow a start: a sufficient condition).  Then the walk is
    // known without walking: nE extensions of chain 0, no removal, results[0] = that chain iff nE >= minMatches (the ratchet
    // only raises minMatches to 2 nE / 3 <= nE).  Anything else: the walk.
    int why = 0;  // (profiling: 0 size limits, 1 event 0 starts no or several chains, 2 start not allowed any more, 3 seed / gap, 4 second start)
    if (!walkAlways && nE >= 2 && nE <= 64 && startSize + nE <= C_POOLSTATES) {  // (rescol holds 64 pairs in every layout)
        why = 1;
        const int4 myEv = lane < nE ? L.ev[lane] : make_int4(0, 0, -2, 0);  // {bIndex, bOffset, seed, gap after}
        const u64 m0 = __ballot(myA == RL(myEv.z, 0)) & initMask;
        if (__popcll(m0) == 1 && RL(myEv.x, 0) > maxBIndex) why = 2;
        if (__popcll(m0) == 1 && RL(myEv.x, 0) <= maxBIndex) {
            why = 3;
            const int r0 = __builtin_ctzll(m0);
            const int r = r0 + lane;
            const int aSeedAt = __shfl(myA, r & 63, 64), aGapAt = __shfl(myOff, r & 63, 64);
            const int prevGapAfter = __shfl_up(myEv.w, 1, 64);
            bool bad = false;
            if (lane >= 1 && lane < nE) {
                int minGap, maxGap;
                gap_range(prevGapAfter + myEv.y, k, minGap, maxGap);
                bad = r >= aLen || aSeedAt != myEv.z || aGapAt < minGap || aGapAt > maxGap;
            }
            if (!__ballot(bad)) {
                why = 4;
                for (int i = 0; i < initialSize; i++) {
                    const int ai = RL(myA, i);
                    if (lane >= 1 && lane < nE && myEv.z == ai && i != r) bad = true;
                }
            }
            if (!__ballot(bad)) {
                if (prof) cp->nperfect++;
                if (nE < minMatches) return 0;
                if (lane < nE) L.rescol[lane] = (uint32_t)r | ((uint32_t)(myEv.x >> 1) << 6);
                return nE;
            }
        }
    }
    if (prof) cp->why += 1ull << (12 * why);

    // removeOpenState :390-409 for chain i_ (uniform): results[0] keeps a copy of its column
#define REMOVE_OPEN_R(i_)                                                          \
    {                                                                              \
        const int ri_ = (i_);                                                      \
        const int last_ = openSize - 1;                                            \
        const int sl_ = RL(st_len, ri_);                                           \
        if (sl_ >= minMatches) {                                                   \
            if ((sl_ * 2) / 3 > minMatches) minMatches = (sl_ * 2) / 3;            \
            if (resultsSize >= C_RESULTS) {                                        \
                *err |= 4;                                                         \
                return 0;                                                          \
            }                                                                      \
            if (resultsSize == 0) {                                                \
                firstLen = sl_;                                                    \
                if (lane < sl_) L.rescol[lane] = L.col[ri_][lane];                 \
            }                                                                      \
            resultsSize++;                                                         \
        } else {                                                                   \
            live -= sl_;                                                           \
        }                                                                          \
        if (ri_ != last_) {                                                        \
            const int ll_ = RL(st_len, last_);                                     \
            if (lane < ll_) L.col[ri_][lane] = L.col[last_][lane];                 \
            const int m0_ = RL(st_aPos, last_), m1_ = RL(st_bPos, last_), m2_ = RL(st_aGap, last_),        \
                      m3_ = RL(st_bGap, last_), m4_ = RL(st_aGapIndex, last_);     \
            if (lane == ri_) {                                                     \
                st_aPos = m0_;                                                     \
                st_bPos = m1_;                                                     \
                st_aGap = m2_;                                                     \
                st_bGap = m3_;                                                     \
                st_aGapIndex = m4_;                                                \
                st_len = ll_;                                                      \
            }                                                                      \
        }                                                                          \
        openSize--;                                                                \
    }

    int4 evNext = L.ev[0];
    for (int e = 0; e < nE; e++) {
        const int4 evc = evNext;
        if (e + 1 < nE) evNext = L.ev[e + 1];
        const int bIndex = RFL(evc.x), bOffset = RFL(evc.y), bSeed = RFL(evc.z), gapAfter = RFL(evc.w);
        const u64 m = __ballot(myA == bSeed);  // reduced a positions holding this seed
        int found = -1;
        if (openSize > 0) {  // searchMatch :465-547
            int outcome = 0;  // 1 keep, 2 too short, 3 ran off the end of a, 4 extends
            int bGap = 0, aGap = 0, aGapIndex = 0, xj = -1;
            if (lane < openSize) {
                bGap = st_bGap + bOffset;
                int minGap, maxGap;
                gap_range(bGap, k, minGap, maxGap);
                aGap = st_aGap;
                aGapIndex = st_aGapIndex;
                bool ended = false;
                while (aGap < minGap) {
                    if (aGapIndex >= aRedLen) {
                        ended = true;
                        break;
                    }
                    aGap += L.aRed[aGapIndex + 1] + k;
                    aGapIndex += 2;
                }
                if (ended) {
                    outcome = 3;
                } else {
                    if (aGap <= maxGap) {
                        // first reduced position >= the cursor with this seed; gaps grow strictly, so it is the
                        // window hit iff its cumulative gap is still <= maxGap
                        const int r0 = aGapIndex >> 1;
                        const u64 mm = r0 < 64 ? (m >> r0) : 0ull;
                        if (mm) {
                            const int r = r0 + __builtin_ctzll(mm);
                            if (r == r0 || aGap + L.ps[r] - L.ps[r0] <= maxGap) xj = 2 * r + 1;
                        }
                    }
                    if (xj >= 0)
                        outcome = 4;
                    else
                        outcome = (st_len + (bN - bIndex) < minMatches) ? 2 : 1;
                }
            }
            const u64 brk = __ballot(outcome >= 3);
            const int ibLane = brk ? 63 - __builtin_clzll(brk) : -1;
            if (outcome == 1 && lane > ibLane) {
                st_bGap = bGap + gapAfter + k;
                st_aGap = aGap;
                st_aGapIndex = aGapIndex;
            }
            u64 shortMask = __ballot(outcome == 2 && lane > ibLane);
            while (shortMask) {
                const int ln = 63 - __builtin_clzll(shortMask);
                shortMask &= ~(1ull << ln);
                REMOVE_OPEN_R(ln);
            }
            if (ibLane >= 0) {
                if (RL(outcome, ibLane) == 3) {
                    REMOVE_OPEN_R(ibLane);
                } else {
                    const int j = RL(xj, ibLane);
                    found = j;
                    if (++live > C_POOLSTATES) {
                        *err |= 2;
                        return 0;
                    }
                    const int nl = RL(st_len, ibLane) + 1;
                    if (nl > (int)LW::COLN) return -1;  // (CSlim) longer than a column
                    const int nextGap = L.aRed[j + 1];
                    if (lane == ibLane) {
                        L.col[ibLane][nl - 1] = (typename LW::col_t)((uint32_t)(j >> 1) | ((uint32_t)(bIndex >> 1) << 6));
                        st_aPos = j;
                        st_bPos = bIndex;
                        st_aGapIndex = j + 2;
                        st_aGap = nextGap;
                        st_bGap = gapAfter;
                        st_len = nl;
                    }
                    if ((nl * 2) / 3 > minMatches) {
                        minMatches = (nl * 2) / 3;
                        maxBIndex = bN - minMatches * 2 + 1;
                    }
                }
            }
        }
        if (bIndex <= maxBIndex) {  // new chains :550-587
            u64 mi = m & initMask;
            while (mi) {
                const int i = __builtin_ctzll(mi);
                mi &= mi - 1;
                const int aPos = 2 * i + 1;
                if (aPos == found) continue;
                if (found != -1) {
                    if (__ballot(lane < openSize && st_bPos == bIndex && st_aPos == aPos)) found = aPos;
                }
                if (found == aPos) continue;
                if (openSize >= (int)LW::ROWS) return -1;  // the reference keeps up to 500 open chains: lds tier
                if (++live > C_POOLSTATES) {
                    *err |= 2;
                    return 0;
                }
                const int nextGap = L.aRed[aPos + 1];
                if (lane == openSize) {
                    L.col[openSize][0] = (typename LW::col_t)((uint32_t)i | ((uint32_t)(bIndex >> 1) << 6));
                    st_aPos = aPos;
                    st_bPos = bIndex;
                    st_aGapIndex = aPos + 2;
                    st_aGap = nextGap;
                    st_bGap = gapAfter;
                    st_len = 1;
                }
                openSize++;
            }
        }
    }
#undef REMOVE_OPEN_R
    // :597-604
    const u64 okMask = __ballot(lane < openSize && st_len >= minMatches);
    if (okMask) {
        const int cnt = __popcll(okMask);
        if (resultsSize + cnt > C_RESULTS) {
            *err |= 4;
            return 0;
        }
        if (resultsSize == 0) {
            const int f = __builtin_ctzll(okMask);
            firstLen = RL(st_len, f);
            if (lane < firstLen) L.rescol[lane] = L.col[f][lane];
        }
        resultsSize += cnt;
    }
    return resultsSize ? firstLen : 0;
}

// lds tier.  Returns the length of results[0] or 0; *resNode = its last node.
__device__ int wave_chain_lds(int aLen, int startSize, int bN, int minMatches, int k, CWave& L, CNode* __restrict__ nodes,
                              int* resNode, uint32_t* err) {
    const int lane = dp_lane();
    int nNodes = 0;
    int live = startSize;
    const int initialSize = startSize;
    const int aRedLen = aLen * 2 + 1;
    const int nE = wave_b_events<false>(bN, k, L, C_BCAP / 2);

    // reduced a seeds, one per lane and 64-position block, for the per-event equality masks
    int myA[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int j = c * 64 + lane;
        myA[c] = j < aLen ? L.aRed[2 * j + 1] : -1;
    }

    int openSize = 0, resultsSize = 0, firstLen = 0, firstNode = -1;
    int maxBIndex = bN - minMatches * 2 + 1;

    // removeOpenState :390-409 for chain i_ (wave-uniform)
#define REMOVE_OPEN_W(i_)                                                          \
    {                                                                              \
        const int ri_ = (i_);                                                      \
        const int sl_ = L.o_len[ri_], sn_ = L.o_node[ri_];                         \
        const int last_ = openSize - 1;                                            \
        if (lane == 0) {                                                           \
            L.o_aPos[ri_] = L.o_aPos[last_];                                       \
            L.o_bPos[ri_] = L.o_bPos[last_];                                       \
            L.o_aGap[ri_] = L.o_aGap[last_];                                       \
            L.o_bGap[ri_] = L.o_bGap[last_];                                       \
            L.o_aGapIndex[ri_] = L.o_aGapIndex[last_];                             \
            L.o_len[ri_] = L.o_len[last_];                                         \
            L.o_node[ri_] = L.o_node[last_];                                       \
        }                                                                          \
        openSize--;                                                                \
        if (sl_ >= minMatches) {                                                   \
            if ((sl_ * 2) / 3 > minMatches) minMatches = (sl_ * 2) / 3;            \
            if (resultsSize >= C_RESULTS) {                                        \
                *err |= 4;                                                         \
                return 0;                                                          \
            }                                                                      \
            if (resultsSize == 0) {                                                \
                firstLen = sl_;                                                    \
                firstNode = sn_;                                                   \
            }                                                                      \
            resultsSize++;                                                         \
        } else {                                                                   \
            live -= sl_;                                                           \
        }                                                                          \
    }

    for (int e = 0; e < nE; e++) {
        const int bIndex = L.evIdx[e];
        const int bOffset = L.evOff[e];
        const int bSeed = L.bSegL[bIndex];
        const int gapAfter = L.bSegL[bIndex + 1];
        int found = -1;
        // searchMatch :465-547
        int hi = openSize;
        while (hi > 0) {
            const int lo = hi > 64 ? hi - 64 : 0;
            const int i = lo + lane;
            const bool act = i < hi;
            int outcome = 0;  // 1 keep, 2 too short, 3 ran off the end of a, 4 extends
            int bGap = 0, aGap = 0, aGapIndex = 0, xj = -1;
            if (act) {
                bGap = L.o_bGap[i] + bOffset;
                int minGap, maxGap;
                gap_range(bGap, k, minGap, maxGap);
                aGap = L.o_aGap[i];
                aGapIndex = L.o_aGapIndex[i];
                bool ended = false;
                while (aGap < minGap) {
                    if (aGapIndex >= aRedLen) {
                        ended = true;
                        break;
                    }
                    aGap += L.aRed[aGapIndex + 1] + k;
                    aGapIndex += 2;
                }
                if (ended) {
                    outcome = 3;
                } else {
                    if (aGap <= maxGap) {
                        int g = aGap;
                        for (int j = aGapIndex; j < aRedLen && g <= maxGap; j += 2) {
                            if (L.aRed[j] == bSeed) {
                                xj = j;
                                break;
                            }
                            g += L.aRed[j + 1] + k;
                        }
                    }
                    if (xj >= 0)
                        outcome = 4;
                    else
                        outcome = (L.o_len[i] + (bN - bIndex) < minMatches) ? 2 : 1;
                }
            }
            const u64 brk = __ballot(outcome >= 3);
            const int ibLane = brk ? 63 - __builtin_clzll(brk) : -1;
            if (outcome == 1 && lane > ibLane) {
                L.o_bGap[i] = bGap + gapAfter + k;
                L.o_aGap[i] = aGap;
                L.o_aGapIndex[i] = aGapIndex;
            }
            u64 shortMask = __ballot(outcome == 2 && lane > ibLane);
            while (shortMask) {
                const int ln = 63 - __builtin_clzll(shortMask);
                shortMask &= ~(1ull << ln);
                REMOVE_OPEN_W(lo + ln);
            }
            if (ibLane < 0) {
                hi = lo;
                continue;
            }
            const int bi = lo + ibLane;
            if (__shfl(outcome, ibLane, 64) == 3) {
                REMOVE_OPEN_W(bi);
            } else {
                const int j = __shfl(xj, ibLane, 64);
                found = j;
                if (nNodes >= (int)C_NODES) {
                    *err |= 8;
                    return 0;
                }
                if (++live > C_POOLSTATES) {
                    *err |= 2;
                    return 0;
                }
                const int nl = L.o_len[bi] + 1;
                if (lane == 0) {
                    CNode nd;
                    nd.a = (uint16_t)L.aMap[j / 2];
                    nd.b = (uint16_t)(bIndex / 2);
                    nd.prev = L.o_node[bi];
                    node_put(L, nodes, nNodes, nd);
                    L.o_aPos[bi] = j;
                    L.o_bPos[bi] = bIndex;
                    L.o_aGapIndex[bi] = j + 2;
                    L.o_aGap[bi] = L.aRed[j + 1];
                    L.o_bGap[bi] = gapAfter;
                    L.o_len[bi] = nl;
                    L.o_node[bi] = nNodes;
                }
                nNodes++;
                if ((nl * 2) / 3 > minMatches) {
                    minMatches = (nl * 2) / 3;
                    maxBIndex = bN - minMatches * 2 + 1;
                }
            }
            break;
        }
        // new chains :550-587
        if (bIndex <= maxBIndex) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c * 64 >= initialSize) break;
                u64 m = __ballot(myA[c] == bSeed && c * 64 + lane < initialSize);
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int i = c * 64 + bit;
                    const int aPos = 2 * i + 1;
                    if (aPos == found) continue;
                    if (found != -1) {
                        bool dup = false;
                        for (int base = 0; base < openSize; base += 64) {
                            const int j = base + lane;
                            if (__ballot(j < openSize && L.o_bPos[j] == bIndex && L.o_aPos[j] == aPos)) {
                                dup = true;
                                break;
                            }
                        }
                        if (dup) found = aPos;
                    }
                    if (found == aPos || openSize >= C_OPEN) continue;
                    if (nNodes >= (int)C_NODES) {
                        *err |= 8;
                        return 0;
                    }
                    if (++live > C_POOLSTATES) {
                        *err |= 2;
                        return 0;
                    }
                    if (lane == 0) {
                        CNode nd;
                        nd.a = (uint16_t)L.aMap[i];
                        nd.b = (uint16_t)(bIndex / 2);
                        nd.prev = -1;
                        node_put(L, nodes, nNodes, nd);
                        L.o_aPos[openSize] = aPos;
                        L.o_bPos[openSize] = bIndex;
                        L.o_aGapIndex[openSize] = aPos + 2;
                        L.o_aGap[openSize] = L.aRed[aPos + 1];
                        L.o_bGap[openSize] = gapAfter;
                        L.o_len[openSize] = 1;
                        L.o_node[openSize] = nNodes;
                    }
                    nNodes++;
                    openSize++;
                }
            }
        }
    }
#undef REMOVE_OPEN_W
    // :597-604
    for (int base = 0; base < openSize; base += 64) {
        const int i = base + lane;
        const u64 okMask = __ballot(i < openSize && L.o_len[i] >= minMatches);
        if (!okMask) continue;
        const int cnt = __popcll(okMask);
        if (resultsSize + cnt > C_RESULTS) {
            *err |= 4;
            return 0;
        }
        if (resultsSize == 0) {
            const int f = base + __builtin_ctzll(okMask);
            firstLen = L.o_len[f];
            firstNode = L.o_node[f];
        }
        resultsSize += cnt;
    }
    *resNode = firstNode;
    return resultsSize ? firstLen : 0;
}


struct MRec {
    uint32_t q, t;
    uint32_t off, len;
};

// Anchors of a chain on its target: base offset of the first matched target seed from the target's start
// (SeedSequence.GetSeedOffset, seeds/sequence.go) and of the last one from its end (GetSeedOffsetFromEnd).  Trimmed()
// (overlap/combine.go:171-181) needs both for every match; summing the gaps on the host means streaming the whole chunk
// (~6 KB of pinned memory) per match, here it is one wave per match over segments that are resident anyway.
// anchors[2*slot] = seg[0] + sum_{t=1..first}(seg[2t]+k), anchors[2*slot+1] = seg[n-1] + sum_{t=last+1..ns-1}(seg[2t]+k);
// DP_NO_ANCHOR (INT32_MIN) when the chain's indices are not inside the target (the host then sums itself).  An anchor itself may
// be negative: a chunk is a SubSequence of its read's seed sequence, its first gap is the gap in front of its first seed - negative
// where two seeds overlap, which in the dense-seed regime (k = 10) they mostly do - and so may its last gap be.  (Until round 4
// the sentinel was -1 and the consensus kernel read every negative anchor as "unknown": five windows per k = 10 round went to
// the host path for it, each time with the round's whole scan output - 90 MB - fetched behind them.)
struct AnchorFetch {  // 8-byte words copied by the launch (either direction: pinned host blocks on one side)
    unsigned long long* dst[3];
    const unsigned long long* src[3];
    unsigned long long n8[3];
};
struct match_anchor_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const MRec* __restrict__ recs, const uint32_t* __restrict__ n_pairs,
                                                           uint32_t pair_cap, const int32_t* __restrict__ mb,
                                                           const dp_seq_ref* __restrict__ refs, const int32_t* __restrict__ segs, int k,
                                                           int32_t* __restrict__ anchors, const AnchorFetch F, uint32_t* __restrict__ zero_word,
                                                           const int32_t* __restrict__ ma, const int32_t* __restrict__ qsegs,
                                                           const u64* __restrict__ qoff, int32_t* __restrict__ cover) {
    const int lane = threadIdx.x & 63;
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 2) zero_word[threadIdx.x] = 0;  // (the consensus stage's two list counters: next launches)
    // (the consensus kernel's input block - pinned host memory - is brought over by this launch, which runs just before it; a
    // pending chaining stage's cursor block and per-query words go the other way)
#pragma unroll
    for (int r = 0; r < 3; r++)
        for (unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; j < F.n8[r]; j += (unsigned long long)gridDim.x * blockDim.x)
            F.dst[r][j] = __builtin_nontemporal_load(&F.src[r][j]);
    const uint32_t nslots = min(*n_pairs, pair_cap);
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); slot < nslots; slot += waves) {
        const MRec r = recs[slot];
        if (r.len == 0) continue;
        const int first = mb[r.off], last = mb[r.off + r.len - 1];
        const dp_seq_ref ref = refs[r.t];
        const int ns = (int)ref.n_seeds;
        const int32_t* s = segs + ref.seg_off;
        if (first < 0 || last < 0 || first >= ns || last >= ns) {
            if (lane == 0) {
                anchors[2 * slot] = anchors[2 * slot + 1] = DP_NO_ANCHOR;
                if (cover) cover[2 * slot] = cover[2 * slot + 1] = DP_NO_ANCHOR;
            }
            continue;
        }
        int a = 0, b = 0;
        for (int t = 1 + lane; t <= first; t += 64) a += s[2 * t] + k;
        for (int t = last + 1 + lane; t <= ns - 1; t += 64) b += s[2 * t] + k;
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        if (lane == 0) {
            anchors[2 * slot] = s[0] + a;
            anchors[2 * slot + 1] = s[2 * ns] + b;
        }
        if (!cover) continue;
        // GetBasesCovered on both sides of the match (seeds/sequence.go:830; the consensus stage's filter, commands/overlap.go:205):
        // len * k plus the negative gaps between consecutive matched seeds.  One lane per matched pair here; the consensus kernel
        // used to walk a match's pairs on one lane, two dependent loads per pair (4.5 of a mean window's 41 us).  The query side is
        // summed on the FORWARD query with flipped indices for a match of the reverse-complement query, exactly as that walk did.
        {
            const uint32_t qf = r.q & ~1u;
            const bool isRc = (r.q & 1u) != 0;
            const int32_t* aSeg = qsegs + qoff[qf];
            const int sA = (int)(qoff[qf + 1] - qoff[qf]) >> 1;
            const int a0 = ma[r.off], b0 = first;
            const bool bad0 = a0 < 0 || a0 >= sA || b0 < 0 || b0 >= ns;
            bool badI = false;
            int ca = 0, cb = 0;
            for (uint32_t i = 1 + (uint32_t)lane; i < r.len && !bad0; i += 64) {
                const int pa = ma[r.off + i - 1], a1 = ma[r.off + i], pb = mb[r.off + i - 1], b1 = mb[r.off + i];
                if (a1 >= sA || a1 < 0 || b1 >= ns || b1 < 0) {
                    badI = true;
                    continue;
                }
                if (pa >= sA || pa < 0 || pb >= ns || pb < 0) continue;  // (flagged by the lane that owns that pair)
                const int lo = isRc ? sA - 1 - a1 : pa, hi = isRc ? sA - 1 - pa : a1;
                int dA = -k;
                for (int j = lo + 1; j <= hi; j++) dA += aSeg[2 * j] + k;
                for (int j = hi + 1; j <= lo; j++) dA -= aSeg[2 * j] + k;
                int dB = -k;
                for (int j = pb + 1; j <= b1; j++) dB += s[2 * j] + k;
                if (dA < 0) ca += dA;
                if (dB < 0) cb += dB;
            }
            for (int o = 32; o > 0; o >>= 1) {
                ca += __shfl_xor(ca, o);
                cb += __shfl_xor(cb, o);
            }
            const bool anyBad = __ballot(badI) != 0;
            if (lane == 0) {
                cover[2 * slot] = bad0 ? DP_NO_ANCHOR : anyBad ? DP_NO_ANCHOR + 1 : ca + (int)r.len * k;
                cover[2 * slot + 1] = cb + (int)r.len * k;
            }
        }
    }
}
};

// ---- the chaining stage: matchWorker's candidate loop (overlap/overlap.go:357-383) without its serial latency -------------
// A query's candidates are chained in ascending order because of the ratchet: a chain longer than 3/2 minMatches raises
// minMatches for the candidates after it (:380-382), and minMatches goes into PairwiseAlignments.  One wave walking a
// query's ~10 candidates one after the other (round 1) leaves the GPU idle: 668 waves, ~15 us of dependent LDS/ballot steps
// per pair.  The ratchet, however, almost always moves exactly once - at the query's first hit.  So:
//   every (query, candidate) PAIR gets a fixed slot: pair p = pbase[q] + rank of the candidate (pbase = exclusive scan of the
//   per-query candidate counts the query kernel leaves behind), a record MRec[p] and a scratch column for its chain;
//   walk(0)  one wave per query: candidates in order until the first pair that is actually chained (one chain per query);
//   spec     one wave per remaining PAIR, all in parallel, chained with the minMatches the query has reached so far; the
//            result is only a proposal: it is stored with the minMatches it was computed for;
//   walk(1)  one wave per query again: replays the ratchet over the proposals in candidate order - a proposal is taken iff
//            it was computed with the minMatches in force at its turn - and stops at the first pair that needs another value;
//   spec, walk(2)  the same once more for the queries whose ratchet moved again; walk(2) chains what is still open itself.
// Every pair ends up with exactly the chain the serial loop produces (bit-identical: tests force 0 passes as well).
struct PSpec {
    int32_t c;    // CountIntersectionTo result of the pair (-1: not computed yet)
    int32_t mm;   // minMatches the proposal below was chained with (-1: none)
    int32_t len;  // its chain length (pairs in the pair's scratch column)
    uint32_t pad;
};
struct QState {
    int32_t mm;      // minMatches in force for the query's next candidate
    uint32_t next;   // rank of the first candidate that is not final yet
};

struct ChainArgs {
    const int32_t* qsegs;
    const u64* qoff;
    uint32_t nq;
    const u64* qsets;
    const uint32_t* qmeta;
    const uint32_t* qcnt;
    const u64* cand;
    const dp_seq_ref* refs;
    const int32_t* segs;
    const u64* seedsets;
    uint32_t W, SW;
    const int32_t* mc;
    uint32_t mc_n;
    int k, maxLength, tier;
    CNode* pool;
    uint32_t pool_stride;  // CNodes per wave: C_NODES + room for a reduced query beyond the LDS copy (maxLength > 510)
    uint32_t* pbase;     // [nq + 1] first pair of each query
    u64* ibase;          // [nq + 1] first scratch int of each query (a pair's column holds nSeeds(q) ints)
    uint32_t* clist;     // [pairs] candidate (indexed-sequence index) of each pair, ascending within a query
    uint32_t* pq;        // [pairs] query of each pair
    int pass;            // proposal pass this launch belongs to (cursor[8 + pass] != 0: a query was still open when it started)
    PSpec* pspec;        // [pairs]
    QState* qstate;      // [nq]
    MRec* recs;          // [pairs] final record of each pair; len 0 = no match
    int32_t *sa, *sb;    // scratch columns
    int32_t *ma, *mb;    // packed chains of the final records
    uint32_t pair_cap;
    u64 sint_cap;
    uint32_t int_cap;
    uint32_t* cursor;    // [0] packed ints used, [2] error bits, [3] overflow flag, [8 + pass] "a query is open" flags, [16..19] totals, [24 + 2 * pass + (q & 1)], pass 0 / 1: pairs left open by the pass's resolve step, [32 ..] 64 shards of the algorithmic bytes (u64)
    int walk_always;      // DP_CHAIN_PERFECT=0 (tests): no pair takes the perfect-chain shortcut of wave_chain_reg
    int pack;             // 1: final chains are copied into ma/mb, densely (what a host fetch wants); 0: they stay where they were
                          // chained - the pair's scratch column - and the record's offset points there (ma = sa, mb = sb for the
                          // device consumers; dp_fetch_overlaps packs them then, should a host consumer turn up)
    uint32_t walk_blocks; // grid of chain_walk_kernel for this round (its node pool is sized for it)
    uint32_t walk0_blocks; // grid of its slim form (mode 0)
    uint32_t n_refs;      // entries of refs[] (indexed sequences, or their upper bound)
    uint32_t prof_walk_slot;   // first per-wave slot of walk(0) in prof[] (behind the passes')
    uint32_t prof_stride;      // per-wave slots of one pass
    unsigned long long* prof;  // DP_CHAIN_PROF=1: [0] pairs looked at, [1] chained, [2..6] wall-clock ticks (100 MHz) per phase of chain_spec_kernel
};

// per-query candidate counts -> pair / scratch offsets (one workgroup; a round has a few hundred to a few ten thousand queries)
struct pair_scan_kernel {
    enum { THREADS = 1024 };
    static __device__ void run(const uint32_t* __restrict__ qcnt, const u64* __restrict__ qoff, uint32_t nq,
                                                          uint32_t* __restrict__ pbase, u64* __restrict__ ibase, u64* __restrict__ totals) {
    __shared__ u64 shp[1024], shi[1024];
    pair_scan_body<1024>(qcnt, qoff, nq, pbase, ibase, totals, shp, shi);
}
};

// Load that goes to L2: for words another lane of this wave has just stored (a plain load may be served from a stale line of
// the CU's vector L1).
__device__ __forceinline__ int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a (the query) into LDS: segments and, when it fits, its seed bitset.  Returns the bitset to probe (LDS or global).
template <class LW>
__device__ __forceinline__ const u64* chain_stage_a(LW& L, const int32_t* __restrict__ aSeg, int aN, const u64* __restrict__ qset,
                                                    uint32_t SW, bool aStaged) {
    const int lane = dp_lane();
    if (aStaged) {
        for (int i = lane; i < aN; i += 64) L.aSegL[i] = aSeg[i];
        if constexpr (!LW::SLIM) {
            if (SW <= C_QSW) {
                for (uint32_t i = lane; i < SW; i += 64) L.qsetL[i] = qset[i];
            }
        }
    }
    if constexpr (!LW::SLIM) {
        if (aStaged && SW <= C_QSW) return (const u64*)L.qsetL;
    }
    return qset;
}

// CountIntersectionTo(seedSet, minMatches) (overlap.go:359): the asm's early exit only ever returns a value >= maxCount,
// so comparing the full popcount with minMatches is the same test
__device__ __forceinline__ int chain_prefilter(const u64* __restrict__ tset, const u64* qs, uint32_t SW) {
    int c = 0;
    for (uint32_t w = dp_lane(); w < SW; w += 64) c += __popcll(tset[w] & qs[w]);
    return RFL(wave_sum(c));
}
// The same count - CountIntersectionTo(seedSet, matchSet), overlap.go:362: distinct seeds of the query that the target holds -
// from the query's side: its seeds are staged (L.aSegL), a few dozen of them, so one probe of the target's set per seed (one
// round of loads, 8 B each) replaces a pass over both sets' rows (157 words each at 10 k seeds: three rounds of two loads, the
// largest single share of a pair's time in chain_spec_kernel: DP_CHAIN_PROF).  Lanes 0 .. nSeeds-1 (nSeeds <= 64) own one seed
// each; a seed counts at its first occurrence in the query.  *inMask = membership of every seed (what chain_pair's aFlag holds).
template <class LW>
__device__ __forceinline__ int chain_prefilter_seeds(const LW& L, const u64* __restrict__ tset, int nSeeds, u64* inMask) {
    const int i = dp_lane();
    bool in = false, dup = false;
    if (i < nSeeds) {
        const int32_t seed = L.aSegL[2 * i + 1];
        in = bs_contains(tset, seed);
        for (int j = 0; j < i; j++) dup |= L.aSegL[2 * j + 1] == seed;
    }
    *inMask = __ballot(in);
    return __popcll(__ballot(in && !dup));
}

// PairwiseAlignments(a, b = candidate t, minMatches) by the whole wave; the chain (results[0], the one matchWorker keeps,
// overlap.go:368-375) goes to the pair's scratch column.  Returns its length (0: none); CSlim only: -1 = does not fit here.
template <class LW>
__device__ int chain_pair(LW& L, CNode* __restrict__ nodes, const ChainArgs& A, const int32_t* __restrict__ aSeg, int aN, bool aStaged,
                          const u64* qs, const u64* __restrict__ qset, uint32_t t, int minMatches, int32_t* __restrict__ ca,
                          int32_t* __restrict__ cb, bool haveAMask = false, u64 aMask = 0, const dp_seq_ref* rPre = nullptr,
                          ChainProf* cp = nullptr) {
    const int lane = dp_lane();
    // (profiling build: ticks of: b staged + flags, initial positions, b events + chain walk, result out)
#define CP_TICK(f_)                                                  \
    if (DP_PROFILING && cp) {                                        \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        const unsigned long long n_ = wall_clock64();                \
        cp->f_ += n_ - cp->last;                                     \
        cp->last = n_;                                               \
    }
    if (DP_PROFILING && cp) cp->last = wall_clock64();
    const u64* tset = A.seedsets + (uint64_t)t * A.SW;
    const dp_seq_ref r = rPre ? *rPre : A.refs[t];
    const int32_t* bSeg = A.segs + r.seg_off;
    const int bN = RFL((int)(2 * r.n_seeds + 1));
    const bool staged = aStaged && bN <= (int)LW::BCAP;
    const int nSeeds = aN >> 1, nBSeeds = bN >> 1;
    const int k = A.k;
    if (LW::SLIM && (!staged || A.tier != 0)) return -1;  // (a forced tier is the full-size path's business)
    if (staged) {  // stage b and both membership bit vectors with the whole wave
        for (int i = lane; i < bN; i += 64) L.bSegL[i] = bSeg[i];
        if (haveAMask && nSeeds <= 64) {  // (the prefilter probed the target's set with these very seeds)
            if (lane == 0) L.aFlag[0] = aMask;
        } else {
            for (int base = 0; base < nSeeds; base += 64) {
                const int s = base + lane;
                bool f = false;
                if (s < nSeeds) f = bs_contains(tset, L.aSegL[2 * s + 1]);
                const u64 m = __ballot(f);
                if (lane == 0) L.aFlag[base >> 6] = m;
            }
        }
        for (int base = 0; base < nBSeeds; base += 64) {
            const int s = base + lane;
            bool f = false;
            if (s < nBSeeds) f = bs_contains(qs, L.bSegL[2 * s + 1]);
            const u64 m = __ballot(f);
            if (lane == 0) L.bFlag[base >> 6] = m;
        }
    }
    int resLen = 0, resNode = -1, usedTier = 3;
    uint32_t err = 0;
    CP_TICK(stageb)
    if (staged) {
        const int mm = minMatches == 0 ? 1 : minMatches;
        int aLen = 0, startSize = 0;
        const int prep = wave_prepare_initial(aN, mm, k, A.maxLength, L, &aLen, &startSize);
        if (prep) {
            if (LW::SLIM) return -1;
            if constexpr (!LW::SLIM) {
                if (prep == 2) {  // more kept seeds than the layout stages, fewer than the reference's buffer holds: one lane, from global memory
                    if (lane == 0)
                        resLen = pairwise_align(aSeg, aN, bSeg, bN, qset, tset, nullptr, nullptr, minMatches, k, A.maxLength, L, nodes, &resNode, &err,
                                                A.maxLength > 510 ? (int32_t*)(nodes + C_NODES) : (int32_t*)nullptr);
                    resLen = __shfl(resLen, 0, 64);
                    resNode = __shfl(resNode, 0, 64);
                } else {
                    err |= 1;
                }
            }
            usedTier = 0;
        } else {
            CP_TICK(initial)
            resLen = -1;
            if (aLen <= 64 && (A.tier == 0 || LW::SLIM)) {
                resLen = wave_chain_reg(aLen, startSize, bN, mm, k, L, &err, cp, A.walk_always != 0);
                usedTier = 1;
            }
            if constexpr (LW::SLIM) {
                if (resLen < 0 || err) return -1;  // (capacity errors are the full-size path's to report)
            } else {
                if (resLen < 0) {
                    resLen = wave_chain_lds(aLen, startSize, bN, mm, k, L, nodes, &resNode, &err);
                    usedTier = 2;
                }
            }
        }
    } else {
        if constexpr (!LW::SLIM) {
            if (lane == 0)
                resLen = pairwise_align(aSeg, aN, bSeg, bN, qset, tset, nullptr, nullptr, minMatches, k, A.maxLength, L, nodes, &resNode, &err,
                                        A.maxLength > 510 ? (int32_t*)(nodes + C_NODES) : (int32_t*)nullptr);
            resLen = __shfl(resLen, 0, 64);
            resNode = __shfl(resNode, 0, 64);
        }
    }
    CP_TICK(walk)
    if (lane == 0 && err) atomicOr(&A.cursor[2], err);
    if (err) resLen = 0;
    if (resLen > 0) {
        if (usedTier == 1) {  // pairs sit in the result column
            if (lane < resLen) {
                const uint32_t v = L.rescol[lane];
                ca[lane] = L.aMap[v & 63u];
                cb[lane] = (int32_t)(v >> 6);
            }
        } else if (lane == 0) {
            if constexpr (!LW::SLIM) {
                int node = resNode;
                for (int x = resLen - 1; x >= 0 && node >= 0; x--) {  // extractMatch :326-335
                    CNode nd = node_get(L, nodes, node);
                    ca[x] = nd.a;
                    cb[x] = nd.b;
                    node = nd.prev;
                }
            }
        }
    }
    CP_TICK(out)
#undef CP_TICK
    return resLen > 0 ? resLen : 0;
}

// mode 0: from the query's first candidate up to and including the first pair that is chained; 2 (last kernel of the stage):
// whatever chain_resolve_kernel left open - proposals are taken where they fit, the rest is chained here, serially.
// (mode 1 = mode 2 that stops instead of chaining: kept for experiments.)
// SLIM (mode 0 only): the speculative kernel's 8.6 KB layout, one wave per query with sixteen of them on a CU instead of four
// (a round's ~1 300 queries are then all resident at once: 45 -> 2x us); a pair that needs the full layout stops its query there
// - it stays open, the proposal passes skip it and the final walk (full layout) chains it.
// LY: 0 = the full layout (CWave), 1 = CSlim (mode 0 only)
template <int LY>
struct chain_walk_kernel {
    typedef typename std::conditional<LY == 0, CWave, CSlim>::type LW;
    enum { SLIM = LY != 0, WAVES = SLIM ? S_WAVES : C_WAVES, THREADS = 64 * WAVES };
    static __device__ void run(const ChainArgs A, const int mode) {
    __shared__ LW sh[WAVES];
    LW& L = sh[threadIdx.x >> 6];
    const int lane = dp_lane();
    // (every wave owns a slice of the node pool, sized for THIS round's grid: in a launch shared with other rounds - the largest
    // round's grid - the blocks beyond it have nothing to do)
    const uint32_t nblocks = SLIM ? A.walk0_blocks : A.walk_blocks;
    if (blockIdx.x >= nblocks) return;
    const uint32_t waves = nblocks * WAVES;
    const uint32_t gw = blockIdx.x * WAVES + (threadIdx.x >> 6);
    CNode* nodes = SLIM ? (CNode*)nullptr : A.pool + (uint64_t)gw * A.pool_stride;  // (C_NODES links + the spill area of pairwise_align)
    if (mode != 0 && (A.cursor[3] != 0 || (A.pass > 0 && A.cursor[8 + A.pass] == 0))) return;  // overflow / every query closed already
    // (profiling build, mode 0: per-wave sums like chain_spec_kernel's, in the slots behind the passes')
    const bool wprof = DP_PROFILING && A.prof && mode == 0;
    ChainProf pf = {};
    unsigned long long tprev = 0;
#define WK_TICK(f_)                                                  \
    if (wprof) {                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        const unsigned long long now_ = wall_clock64();              \
        pf.f_ += now_ - tprev;                                       \
        tprev = now_;                                                \
    }
    const unsigned long long tkernel0 = wprof ? wall_clock64() : 0ull;
    for (uint32_t q = gw; q < A.nq; q += waves) {
        if (wprof) tprev = wall_clock64();
        const uint32_t cnt = A.qcnt[q];
        const int aN = RFL((int)(A.qoff[q + 1] - A.qoff[q]));
        const uint32_t nSeeds = (uint32_t)aN / 2;
        if (cnt == 0) continue;
        const uint32_t pb = A.pbase[q];
        const u64 ib = A.ibase[q];
        if ((u64)pb + cnt > (u64)A.pair_cap || ib + (u64)cnt * nSeeds > A.sint_cap) {
            if (lane == 0) A.cursor[3] = 1;
            continue;
        }
        int mm;
        uint32_t next;
        if (mode == 0) {
            // Matches() result as a list: the set bits of the query's cand row in ascending order
            uint32_t running = 0;
            for (uint32_t wb = 0; wb < A.W; wb += 64) {
                const uint32_t w = wb + lane;
                u64 m = w < A.W ? A.cand[(uint64_t)q * A.W + w] : 0ull;
                const int pc = __popcll(m);
                const int incl = wave_incl_sum(pc);
                uint32_t at = running + (uint32_t)(incl - pc);
                while (m) {
                    A.pq[pb + at] = q;
                    A.clist[pb + at++] = w * 64 + (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                }
                running += (uint32_t)__shfl(incl, 63, 64);
            }
            for (uint32_t i = lane; i < cnt; i += 64) {
                PSpec e = {-1, -1, 0, 0};
                A.pspec[pb + i] = e;
            }
            __threadfence_block();
            mm = RFL(nSeeds < A.mc_n ? A.mc[nSeeds] : 0x7fffffff);  // int(hitFraction*numSeeds+0.5), overlap.go:356
            next = 0;
        } else {
            const QState st = A.qstate[q];
            mm = RFL(st.mm);
            next = (uint32_t)RFL((int)st.next);
            if (next >= cnt) continue;
        }
        WK_TICK(rec)  // the query's records and (mode 0) its candidate list
        const int32_t* aSeg = A.qsegs + A.qoff[q];
        const u64* qset = A.qsets + (uint64_t)q * A.SW;
        const bool aStaged = aN <= (int)LW::ACAP && A.tier != 3;
        const u64* qs = chain_stage_a(L, aSeg, aN, qset, A.SW, aStaged);
        WK_TICK(stagea)
        // algorithmic bytes of this query's share: two bitset rows per candidate (the exact-intersection
        // prefilter), both segment arrays per chained pair (4-byte ints here), the chain written out - counted once per
        // pair, when it becomes final
        u64 algBytes = 0;
        uint32_t i = next;
        for (; i < cnt; i++) {
            const uint32_t p = pb + i;
            const uint32_t t = mode == 0 ? ld_agent(&A.clist[p]) : A.clist[p];
            int32_t* ca = A.sa + ib + (u64)i * nSeeds;
            int32_t* cb = A.sb + ib + (u64)i * nSeeds;
            PSpec sp = {-1, -1, 0, 0};
            if (mode != 0) sp = A.pspec[p];
            int c = RFL(sp.c);
            const int spmm = RFL(sp.mm);
            int len;
            bool chained = false;
            bool haveMask = false;
            u64 aMask = 0;
            if (c < 0) {
                if (aStaged && nSeeds <= 64) {
                    c = chain_prefilter_seeds(L, A.seedsets + (uint64_t)t * A.SW, (int)nSeeds, &aMask);
                    haveMask = true;
                } else {
                    c = chain_prefilter(A.seedsets + (uint64_t)t * A.SW, qs, A.SW);
                }
            }
            pf.pairs++;
            WK_TICK(pre)
            if (c < mm) {
                len = 0;
            } else if (spmm == mm) {
                len = RFL(sp.len);
            } else if (mode == 1) {
                break;  // needs chaining with a minMatches nobody proposed for: the next spec pass does it
            } else {
                if (DP_PROFILING && A.prof && mode == 2 && lane == 0) {  // (what the final walk still chains, and how much of it because of the slim layout)
                    atomicAdd(&A.cursor[20], 1u);
                    if (spmm == -2) atomicAdd(&A.cursor[21], 1u);
                }
                len = chain_pair(L, nodes, A, aSeg, aN, aStaged, qs, qset, t, mm, ca, cb, haveMask, aMask, nullptr, wprof ? &pf : (ChainProf*)nullptr);
                if (SLIM && len < 0) {  // needs the full layout: the query stays open at this pair (marked: slim passes skip it)
                    if (lane == 0) {
                        PSpec o = {c, -2, 0, 0};
                        A.pspec[p] = o;
                    }
                    break;
                }
                chained = true;
                pf.chained++;
                WK_TICK(pair)
            }
            algBytes += 16ull * A.SW;
            if (c >= mm) algBytes += 4ull * (u64)(aN + (int)(2 * A.refs[t].n_seeds + 1));
            uint32_t off = 0;
            if (len > 0 && !A.pack) {
                off = (uint32_t)(ib + (u64)i * nSeeds);
                algBytes += 8ull * (u64)len;
            } else if (len > 0) {
                if (lane == 0) off = atomicAdd(&A.cursor[0], (uint32_t)len);
                off = (uint32_t)__shfl((int)off, 0, 64);
                if ((u64)off + (u64)len <= (u64)A.int_cap) {
                    __threadfence_block();
                    for (int x = lane; x < len; x += 64) {
                        A.ma[off + x] = ld_agent(&ca[x]);
                        A.mb[off + x] = ld_agent(&cb[x]);
                    }
                } else {
                    if (lane == 0) A.cursor[3] = 1;
                    len = 0;
                }
                algBytes += 8ull * (u64)len;
            }
            if (lane == 0) {
                MRec rec = {q, t, off, (uint32_t)len};
                A.recs[p] = rec;
            }
            if (len > 0 && len * 2 > mm * 3) mm = (len * 2) / 3;  // ratchet, overlap.go:380-382
            if (mode == 0 && chained) {
                i++;
                break;
            }
        }
        if (lane == 0) {
            QState st = {mm, i};
            A.qstate[q] = st;
            if (algBytes) atomicAdd((unsigned long long*)(A.cursor + 32) + (q & 63u), (unsigned long long)algBytes);
            if (mode == 0 && i < cnt) A.cursor[8] = 1u;  // a query is open ahead of proposal pass 0 (a flag: every writer stores 1)
        }
        WK_TICK(out)
    }
    if (wprof && lane == 0) {
        unsigned long long* slot = A.prof + 16 * ((size_t)A.prof_walk_slot + gw);
        slot[0] = pf.pairs, slot[1] = pf.chained, slot[2] = pf.rec, slot[3] = pf.stagea, slot[4] = pf.pre, slot[5] = pf.pair;
        slot[6] = pf.stageb, slot[7] = pf.initial, slot[8] = pf.walk, slot[9] = pf.out;
        slot[10] = tkernel0;
        slot[11] = wall_clock64();
        slot[12] = pf.bev, slot[13] = pf.nev, slot[14] = pf.nperfect, slot[15] = pf.why;
    }
#undef WK_TICK
}
};

__device__ void chain_resolve_query(const ChainArgs& A, uint32_t q, int lane);

// one wave per open pair: prefilter + chain with the minMatches its query has reached; stored as a proposal.
// (Round 6: the later passes on the full layout - four waves a CU, every pair the slim layout cannot hold chained in parallel instead of
// by the final walk - were built and measured SLOWER at k = 13 and at k = 10, profiles/r06/ab_launch_diet_k13.txt, ab_full_from_k10.txt:
// workgroups that want 131 KB of a CU's LDS are placed late.  What made the difference was the slim layout's geometry, see CSlim.)
struct chain_spec_kernel {
    typedef CSlim LW;
    enum { WAVES = S_WAVES, THREADS = 64 * WAVES };
    static __device__ void run(const ChainArgs A, const u64* __restrict__ totals) {
    __shared__ LW sh[WAVES];
    LW& L = sh[threadIdx.x >> 6];
    const int lane = dp_lane();
    const uint32_t waves = gridDim.x * WAVES;
    const uint32_t gw = blockIdx.x * WAVES + (threadIdx.x >> 6);
    // (a query that did not fit the buffers left its pairs' pq / clist unwritten: the host repeats the stage with larger ones)
    if (A.cursor[3] != 0 || A.cursor[8 + A.pass] == 0) return;  // ... or no query is open any more
    const uint32_t total = (uint32_t)min(totals[0], (u64)A.pair_cap);
    // DP_CHAIN_PROF: every wave keeps its own sums (ticks of 10 ns) and stores them into its own slot of the pass - no shared
    // counters (ten thousand atomics on one word take longer than the kernel)
    ChainProf pf = {};
    unsigned long long tprev = 0;
#define SP_TICK(f_)                                                  \
    if ((DP_PROFILING && A.prof)) {                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        const unsigned long long now_ = wall_clock64();              \
        pf.f_ += now_ - tprev;                                       \
        tprev = now_;                                                \
    }
    const unsigned long long tkernel0 = (DP_PROFILING && A.prof) ? wall_clock64() : 0ull;
    for (uint32_t p = gw; p < total; p += waves) {
        if ((DP_PROFILING && A.prof)) tprev = wall_clock64();
        // A pair is a chain of dependent loads (pair -> query -> its records -> its segments -> probes of the target's set ...) at
        // 1-2 us per level under this kernel's own load: everything whose address is known is asked for at once, ahead of the
        // tests that may drop the pair (a load behind a branch waits for the branch), the target's reference included.
        const uint32_t q = A.pq[p];
        const uint32_t t = A.clist[p];
        PSpec sp = A.pspec[p];
        const uint32_t pb_ = A.pbase[q], cnt_ = A.qcnt[q];
        const QState st = A.qstate[q];
        const u64 qo0 = A.qoff[q], qo1 = A.qoff[q + 1];
        const u64 ib = A.ibase[q];
        const dp_seq_ref rT = A.refs[min(t, A.n_refs - 1)];  // (t is only meaningful for a pair that passes the tests below)
        do {  // (a `continue` below leaves this block, not the loop: every pair reports to its query - see the end of the loop body)
        const uint32_t i = p - pb_;
        if (i >= cnt_) continue;  // (beyond a capped query)
        if (i < st.next) continue;  // final already
        const int mm = RFL(st.mm);
        if (RFL(sp.mm) == mm) continue;
        if (RFL(sp.mm) == -2) continue;  // (an earlier pass found that this pair needs the full layout: the final walk's)
        const int aN = RFL((int)(qo1 - qo0));
        const uint32_t nSeeds = (uint32_t)aN / 2;
        if (ib + (u64)cnt_ * nSeeds > A.sint_cap) continue;  // (flagged by the walk)
        const int32_t* aSeg = A.qsegs + qo0;
        const u64* qset = A.qsets + (uint64_t)q * A.SW;
        const bool aStaged = aN <= (int)LW::ACAP;
        pf.pairs++;
        SP_TICK(rec)  // the pair's own records (query state, proposal, candidate)
        const u64* qs = chain_stage_a(L, aSeg, aN, qset, A.SW, aStaged);
        SP_TICK(stagea)  // query side staged
        int c = RFL(sp.c);
        bool haveMask = false;
        u64 aMask = 0;
        if (c < 0) {
            if (aStaged && nSeeds <= 64) {
                c = chain_prefilter_seeds(L, A.seedsets + (uint64_t)t * A.SW, (int)nSeeds, &aMask);
                haveMask = true;
            } else {
                c = chain_prefilter(A.seedsets + (uint64_t)t * A.SW, qs, A.SW);
            }
        }
        SP_TICK(pre)  // prefilter
        int len = 0, pmm = mm;
        if (c >= mm) {
            pf.chained++;
            len = chain_pair(L, (CNode*)nullptr, A, aSeg, aN, aStaged, qs, qset, t, mm, A.sa + ib + (u64)i * nSeeds, A.sb + ib + (u64)i * nSeeds,
                             haveMask, aMask, &rT, (DP_PROFILING && A.prof) ? &pf : (ChainProf*)nullptr);
            if (len < 0) {  // does not fit the slim layout: no proposal (-2: no slim pass tries again), the final walk chains it
                len = 0;
                pmm = -2;
            }
        }
        SP_TICK(pair)  // chain_pair
        if (lane == 0) {
            PSpec o = {c, pmm, len, 0};
            A.pspec[p] = o;
        }
        __builtin_amdgcn_wave_barrier();
        } while (0);
    }
    if ((DP_PROFILING && A.prof) && lane == 0) {
        unsigned long long* slot = A.prof + 16 * ((size_t)A.pass * A.prof_stride + gw);
        slot[0] = pf.pairs, slot[1] = pf.chained, slot[2] = pf.rec, slot[3] = pf.stagea, slot[4] = pf.pre, slot[5] = pf.pair;
        slot[6] = pf.stageb, slot[7] = pf.initial, slot[8] = pf.walk, slot[9] = pf.out;
        slot[10] = tkernel0;
        slot[11] = wall_clock64();
        slot[12] = pf.bev, slot[13] = pf.nev, slot[14] = pf.nperfect, slot[15] = pf.why;
    }
#undef SP_TICK
}
};

// One wave per query: replays the ratchet over the proposals of its open pairs - 64 pairs at a time, one per lane, the
// serial part is a register loop - and makes every pair up to the first one that needs another minMatches final: packed
// chain, record, algorithmic bytes.  Everything that touches memory is lane-parallel.
// One query of the resolve step: replays the ratchet over the proposals of its open pairs - 64 pairs at a time, one per lane, the
// serial part is a register loop - and makes every pair up to the first one that needs another minMatches final: packed
// chain, record, algorithmic bytes.  Everything that touches memory is lane-parallel.
__device__ void chain_resolve_query(const ChainArgs& A, uint32_t q, int lane) {
    const uint32_t cnt = A.qcnt[q];
    if (cnt == 0) return;
    const QState st = A.qstate[q];
    int mm = RFL(st.mm);
    uint32_t next = (uint32_t)RFL((int)st.next);
    if (next >= cnt) return;
    const uint32_t pb = A.pbase[q];
    const int aN = RFL((int)(A.qoff[q + 1] - A.qoff[q]));
    const uint32_t nSeeds = (uint32_t)aN / 2;
    const u64 ib = A.ibase[q];
    if ((u64)pb + cnt > (u64)A.pair_cap || ib + (u64)cnt * nSeeds > A.sint_cap) return;  // (flagged by walk 0)
    u64 algBytes = 0;
    bool stopped = false;
    while (next < cnt && !stopped) {
        const uint32_t i = next + lane;
        const bool valid = i < cnt;
        PSpec sp = {-1, -1, 0, 0};
        uint32_t t = 0;
        if (valid) {
            sp = A.pspec[pb + i];
            t = A.clist[pb + i];
        }
        const int nHere = (int)min(64u, cnt - next);
        int stop = nHere;  // lanes [0, stop) become final
        u64 chainedMask = 0, hitMask = 0;
        for (int l = 0; l < nHere; l++) {
            const int c = RL(sp.c, l);
            if (c < 0) {
                stop = l;
                break;
            }
            if (c < mm) continue;
            if (RL(sp.mm, l) != mm) {
                stop = l;
                break;
            }
            chainedMask |= 1ull << l;
            const int len = RL(sp.len, l);
            if (len > 0) {
                hitMask |= 1ull << l;
                if (len * 2 > mm * 3) mm = (len * 2) / 3;  // ratchet, overlap.go:380-382
            }
        }
        const bool fin = lane < stop;
        const bool hit = fin && ((hitMask >> lane) & 1ull);
        const int myLen = hit ? sp.len : 0;
        uint32_t myOff = (uint32_t)(ib + (u64)i * nSeeds);  // (pack == 0: the chain stays in the pair's scratch column)
        bool room = true;
        if (A.pack) {
            const int incl = wave_incl_sum(myLen);
            const int totalLen = __shfl(incl, 63, 64);
            uint32_t off0 = 0;
            if (totalLen > 0) {
                if (lane == 0) off0 = atomicAdd(&A.cursor[0], (uint32_t)totalLen);
                off0 = (uint32_t)__shfl((int)off0, 0, 64);
            }
            myOff = off0 + (uint32_t)(incl - myLen);
            room = (u64)off0 + (u64)totalLen <= (u64)A.int_cap;
            if (!room && lane == 0) A.cursor[3] = 1;
        }
        if (fin) {
            int wlen = 0;
            if (hit && !A.pack) {
                wlen = myLen;
            } else if (hit && room) {
                const int32_t* ca = A.sa + ib + (u64)i * nSeeds;
                const int32_t* cb = A.sb + ib + (u64)i * nSeeds;
                for (int x = 0; x < myLen; x++) {
                    A.ma[myOff + x] = ca[x];
                    A.mb[myOff + x] = cb[x];
                }
                wlen = myLen;
            }
            MRec rec = {q, t, hit && room ? myOff : 0u, (uint32_t)wlen};
            A.recs[pb + i] = rec;
            algBytes += 16ull * A.SW + 8ull * (u64)wlen;
            if ((chainedMask >> lane) & 1ull) algBytes += 4ull * (u64)(aN + (int)(2 * A.refs[t].n_seeds + 1));
        }
        next += (uint32_t)stop;
        stopped = stop < nHere;
    }
    // algBytes is per lane here: reduce
    unsigned long long ab = algBytes;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ab += __shfl_xor(ab, d, 64);
    if (lane == 0) {
        QState o = {mm, next};
        A.qstate[q] = o;
        if (ab) atomicAdd((unsigned long long*)(A.cursor + 32) + (q & 63u), ab);
        if (next < cnt) {
            A.cursor[9 + A.pass] = 1u;  // still open: the next pass has work (a flag)
            if (A.pass < 2) atomicAdd(&A.cursor[24 + 2 * A.pass + (q & 1u)], cnt - next);  // (what the host sizes the next round's passes by)
        }
    }
}

struct chain_resolve_kernel {
    enum { THREADS = 256 };
    static __device__ void run(const ChainArgs A) {
    const int lane = dp_lane();
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    if (A.cursor[3] != 0 || A.cursor[8 + A.pass] == 0) return;  // buffers overflowed (stage is repeated) / no query was open before this pass
    for (uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); q < A.nq; q += waves) chain_resolve_query(A, q, lane);
}
};

static uint32_t query_dbg_flags() {
    static const uint32_t f = (uint32_t)dp_tune("query_debug", 0);
    return f;
}

// What the host sends for the query stage - query offsets, query segments, the minCount table - as one block in the context's
// pinned staging area, laid out as it will lie on the device.  Returns the block's size (0: nothing to send).
static size_t query_block_stage(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t nq, double hf, uint32_t* mc_n_out,
                                uint32_t* max_seeds_out, int32_t* mc_last_out, size_t* up_off_out, size_t* up_segs_out) {
    const uint64_t nseg = nq ? q_off[nq] : 0;
    // int(hitFraction*float64(n)+0.5) for every n that can occur (seeds/seeds.go:351, overlap/overlap.go:356);
    // evaluated on the host in IEEE double (this file is built with -ffp-contract=off)
    uint32_t maxSeeds = 0;
    for (uint32_t q = 0; q < nq; q++) maxSeeds = std::max<uint32_t>(maxSeeds, (uint32_t)((q_off[q + 1] - q_off[q]) / 2));
    const uint32_t mc_n = std::max<uint32_t>(maxSeeds + 1, 8);
    std::vector<int32_t> mc(mc_n);
    for (uint32_t n = 0; n < mc_n; n++) {
        volatile double prod = hf * (double)n;
        volatile double sum = prod + 0.5;
        mc[n] = (int32_t)sum;
    }
    const size_t up_segs = nseg * 4, up_off = ((size_t)nq + 1) * 8, up_mc = (size_t)mc_n * 4;
    if (dev_reserve(ctx, ctx->d_qsegs, up_off + up_segs + up_mc + 64)) return 0;
    // stage through pinned memory: copies from pageable buffers stall the stream
    if (pin_reserve(ctx, ctx->h_qup, up_segs + up_off + up_mc + 64)) return 0;
    uint8_t* up = (uint8_t*)ctx->h_qup.p;
    memcpy(up, q_off, up_off);
    memcpy(up + up_off, q_segs, up_segs);
    memcpy(up + up_off + up_segs, mc.data(), up_mc);
    *mc_n_out = mc_n;
    *max_seeds_out = maxSeeds;
    *mc_last_out = mc[maxSeeds];
    *up_off_out = up_off;
    *up_segs_out = up_segs;
    return up_off + up_segs + up_mc;
}

// The queries of the round's coming dp_find_overlaps, announced before the index is built: they are staged now and the next
// launch with room for it - dp_index_build_chunked's first kernel - brings them to the device, so that the query stage starts
// with its kernel (no upload launch, no staging between the index build and the query kernel).  dp_find_overlaps recognises the
// block by comparing; anything else in between simply makes it upload as before.
extern "C" int dp_query_prestage(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t n_queries, double hit_fraction) {
    if (!ctx || (n_queries && (!q_segs || !q_off))) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_query_prestage: bad arguments") : DP_ERR_ARG;
    ctx->q_pre_bytes = 0;
    ctx->q_pre_fetched = false;
    if (!n_queries) return DP_OK;
    hipSetDevice(ctx->device);
    uint32_t mc_n = 0, maxSeeds = 0;
    int32_t mcLast = 0;
    size_t up_off = 0, up_segs = 0;
    const size_t bytes = query_block_stage(ctx, q_segs, q_off, n_queries, hit_fraction, &mc_n, &maxSeeds, &mcLast, &up_off, &up_segs);
    if (!bytes) return DP_ERR_HIP;
    ctx->q_pre_bytes = bytes;
    ctx->q_pre_nq = n_queries;
    ctx->q_pre_hf = hit_fraction;
    return DP_OK;
}

// Uploads the queries, builds their seed bitsets and runs the index query (Matches -> GetSharedIDs) for all of them.
// Leaves d_qsegs/d_qoff/d_qsets/d_cand/d_qmeta on the device.  Events ev[4]/ev[5] bracket the query kernel.
int dp_query_stage(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t nq, double hf, uint32_t** d_qmeta_out,
                   uint64_t** d_words_out, int32_t** d_mc_out, uint32_t* mc_n_out, uint32_t** d_qcnt_out) {
    const uint32_t W = ctx->W, SW = ctx->SW, M = ctx->n_seqs;
    // what the host sends - query offsets, query segments, the minCount table - is one block on both sides: one copy.  Announced
    // and on the device already (dp_query_prestage + the index build's first launch)?  Then it is only compared.
    uint32_t mc_n = 0, maxSeeds = 0;
    int32_t mcLast = 0;
    size_t up_off = ((size_t)nq + 1) * 8, up_segs = (size_t)(nq ? q_off[nq] : 0) * 4;
    bool on_device = false;
    if (ctx->q_pre_bytes && ctx->q_pre_fetched && ctx->q_pre_nq == nq && ctx->q_pre_hf == hf && ctx->q_pre_bytes >= up_off + up_segs &&
        ctx->h_qup.p && memcmp(ctx->h_qup.p, q_off, up_off) == 0 && memcmp((const uint8_t*)ctx->h_qup.p + up_off, q_segs, up_segs) == 0) {
        for (uint32_t q = 0; q < nq; q++) maxSeeds = std::max<uint32_t>(maxSeeds, (uint32_t)((q_off[q + 1] - q_off[q]) / 2));
        mc_n = std::max<uint32_t>(maxSeeds + 1, 8);
        if (ctx->q_pre_bytes == up_off + up_segs + (size_t)mc_n * 4) {
            mcLast = ((const int32_t*)((const uint8_t*)ctx->h_qup.p + up_off + up_segs))[maxSeeds];
            on_device = true;
        }
    }
    if (!on_device && ctx->q_pre_fetched) DP_HIP(dp_stream_sync(ctx));  // (a launch may still be copying the announced block)
    ctx->q_pre_bytes = 0;
    ctx->q_pre_fetched = false;
    if (!on_device) {
        const size_t bytes = query_block_stage(ctx, q_segs, q_off, nq, hf, &mc_n, &maxSeeds, &mcLast, &up_off, &up_segs);
        if (!bytes) return DP_ERR_HIP;
    }
    const size_t up_mc = (size_t)mc_n * 4;
    uint8_t* up = (uint8_t*)ctx->h_qup.p;
    ctx->qoff_dev = (const u64*)ctx->d_qsegs.p;
    ctx->qsegs_dev = (const int32_t*)((const uint8_t*)ctx->d_qsegs.p + up_off);
    if (dev_reserve(ctx, ctx->d_qsets, (size_t)nq * SW * 8 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_qmeta, (size_t)nq * 16 + (size_t)nq * 8 + (size_t)nq * 4 + (size_t)mc_n * 4 + 64)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_cand, (size_t)nq * W * 8 + 64)) return DP_ERR_HIP;
    uint32_t* d_qmeta = (uint32_t*)ctx->d_qmeta.p;
    u64* d_words = (u64*)((uint8_t*)ctx->d_qmeta.p + (size_t)nq * 16);
    uint32_t* d_qcnt = (uint32_t*)((uint8_t*)ctx->d_qmeta.p + (size_t)nq * 24);
    const int32_t* d_mc = (const int32_t*)((const uint8_t*)ctx->d_qsegs.p + up_off + up_segs);
    if (dev_reserve(ctx, ctx->d_cursor, C_CURSOR_BYTES)) return DP_ERR_HIP;
    // workgroups per query (DP_QUERY_SPLIT, experiments: see query_kernel)
    static const int split_env = (int)dp_tune("query_split", 0);
    const uint32_t q_split = split_env > 0 ? (uint32_t)std::min(split_env, 16) : 1u;
    static const bool own_rows_env = true;
    // (short rows only - the sparse regime: a few hundred words per query; the dense regime's rows - W ~ 3 k words - are cleared
    // faster by the launch over all of them, and its index query is the kernel the bandwidth figure is quoted on)
    const bool own_rows = own_rows_env && q_split == 1 && (size_t)W + SW <= 1024;
    {   // (the chaining stage's cursor block rides along: it is zero when the first attempt starts; and the same launch fetches
        // the upload block from its pinned staging - no copy is handed to the runtime)
        // (with one workgroup per query - the default - the light query kernel clears its query's rows itself and this launch only
        // brings the upload block over)
        const dp_zero_region z[4] = {{ctx->d_qsets.p, (size_t)nq * SW * 8}, {ctx->d_cand.p, (size_t)nq * W * 8}, {d_qmeta, (size_t)nq * 28},
                                     {ctx->d_cursor.p, C_CURSOR_BYTES}};
        const dp_fetch_region f = {ctx->d_qsegs.p, up, up_off + up_segs + up_mc};
        if (int rc = dp_zero_fetch_regions(ctx, z, own_rows ? 0 : 4, &f, on_device ? 0 : 1)) return rc;  // (nothing at all: no launch)
    }
    DP_HIP(dp_mark(ctx, 4));
    dp_launch<query_kernel<false>>(ctx, dim3(nq * q_split), dim3(64 * Q_WAVES),
                       ctx->qsegs_dev, ctx->qoff_dev, nq, (const u64*)ctx->d_posting.p,
                       (const uint32_t*)ctx->d_pmeta.p, ctx->global_n_seqs ? ctx->global_n_seqs : M, W, (const int32_t*)d_mc, mc_n,
                       (u64*)ctx->d_cand.p, d_qmeta, d_words, d_qcnt, ctx->word_base,
                       ctx->chunks_on_device ? (const uint32_t*)ctx->d_nseqs.p : (const uint32_t*)nullptr, (u64*)ctx->d_qsets.p, SW,
                       query_dbg_flags(), q_split, own_rows ? (u64*)ctx->d_cursor.p : (u64*)nullptr, (uint32_t*)nullptr, 0u);
    // the 16-ladder / exact-count regimes start at minCount 13: only a batch with a query of that many seeds needs the heavy variant
    if (mcLast >= 13) dp_launch<query_kernel<true>>(ctx, dim3(nq * q_split), dim3(64 * Q_WAVES),
                       ctx->qsegs_dev, ctx->qoff_dev, nq, (const u64*)ctx->d_posting.p,
                       (const uint32_t*)ctx->d_pmeta.p, ctx->global_n_seqs ? ctx->global_n_seqs : M, W, (const int32_t*)d_mc, mc_n,
                       (u64*)ctx->d_cand.p, d_qmeta, d_words, d_qcnt, ctx->word_base,
                       ctx->chunks_on_device ? (const uint32_t*)ctx->d_nseqs.p : (const uint32_t*)nullptr, (u64*)ctx->d_qsets.p, SW,
                       query_dbg_flags(), q_split, (u64*)nullptr, (uint32_t*)nullptr, 0u);
    if (maxSeeds > Q_MAXSETS) {
        // a query may hold more sets than the kernel's LDS lists (round 6): the BIG variants, lists in global memory, for those queries
        const uint32_t stride = std::min<uint32_t>(maxSeeds, Q_BIG_MAXSETS);
        if (dev_reserve(ctx, ctx->d_qbig, (size_t)nq * q_split * ((size_t)stride + 2) * Q_BIG_WORDS_PER_SET * 4 + 64)) return DP_ERR_HIP;
        dp_launch<query_kernel<false, true>>(ctx, dim3(nq * q_split), dim3(64 * Q_WAVES),
                       ctx->qsegs_dev, ctx->qoff_dev, nq, (const u64*)ctx->d_posting.p,
                       (const uint32_t*)ctx->d_pmeta.p, ctx->global_n_seqs ? ctx->global_n_seqs : M, W, (const int32_t*)d_mc, mc_n,
                       (u64*)ctx->d_cand.p, d_qmeta, d_words, d_qcnt, ctx->word_base,
                       ctx->chunks_on_device ? (const uint32_t*)ctx->d_nseqs.p : (const uint32_t*)nullptr, (u64*)nullptr, SW,
                       query_dbg_flags(), q_split, (u64*)nullptr, (uint32_t*)ctx->d_qbig.p, stride);
        if (mcLast >= 13) dp_launch<query_kernel<true, true>>(ctx, dim3(nq * q_split), dim3(64 * Q_WAVES),
                       ctx->qsegs_dev, ctx->qoff_dev, nq, (const u64*)ctx->d_posting.p,
                       (const uint32_t*)ctx->d_pmeta.p, ctx->global_n_seqs ? ctx->global_n_seqs : M, W, (const int32_t*)d_mc, mc_n,
                       (u64*)ctx->d_cand.p, d_qmeta, d_words, d_qcnt, ctx->word_base,
                       ctx->chunks_on_device ? (const uint32_t*)ctx->d_nseqs.p : (const uint32_t*)nullptr, (u64*)nullptr, SW,
                       query_dbg_flags(), q_split, (u64*)nullptr, (uint32_t*)ctx->d_qbig.p, stride);
    }
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 5));

    *d_qmeta_out = d_qmeta;
    *d_words_out = d_words;
    *d_mc_out = (int32_t*)d_mc;
    *mc_n_out = mc_n;
    if (d_qcnt_out) *d_qcnt_out = d_qcnt;
    return DP_OK;
}

int dp_fetch_overlaps_impl(dp_ctx* ctx, int want_candidates, dp_match_batch* out);

// State of a chaining stage between its launch and the evaluation of what it reported (the stage may be left pending:
// dp_find_overlaps with want_candidates bit 2, finished by dp_consensus_paf in the wait it needs anyway).
struct FindState {
    uint32_t nq = 0;
    int k = 0;
    uint32_t max_query_len = 0;
    int chain_tier = 0, passes = 3;
    uint32_t walk_blocks = 0, spec_blocks = 1024;
    uint32_t* d_qmeta = nullptr;
    uint32_t* d_qcnt = nullptr;
    const int32_t* d_mc = nullptr;
    uint32_t mc_n = 0;
    uint64_t want_pairs = 0, want_sints = 0, want_ints = 0;
    uint32_t pair_cap = 0, int_cap = 0;
    uint64_t sint_cap = 0;
    int attempt = 0;
    float chain_ms = 0;
    bool pending = false;
    bool defer_fetch = false;  // the first attempt's read-back rides in the anchors launch of the consensus call (a pending stage)
    bool fetch_owed = false;
    uint32_t cur[32];
    double query_ms = 0;
    uint64_t query_bytes = 0, chain_bytes = 0, alg_bytes = 0;
};
void dp_find_state_free(dp_ctx* ctx) {
    delete ctx->find_state;
    ctx->find_state = nullptr;
}
bool dp_find_pending(const dp_ctx* ctx) { return ctx->find_state && ctx->find_state->pending; }
uint32_t dp_find_pair_cap(const dp_ctx* ctx) { return ctx->find_state ? ctx->find_state->pair_cap : 0; }

// one attempt of the chaining stage with the current capacities: launches and the read-back of what it reports; no wait
static int chain_enqueue(dp_ctx* ctx, FindState& st) {
    if (st.attempt > 8) return dp_fail(ctx, DP_ERR_STATE, "dp_find_overlaps: output buffers keep overflowing");
    const uint32_t nq = st.nq;
    const uint32_t W = ctx->W, SW = ctx->SW;
    uint32_t* d_cur = (uint32_t*)ctx->d_cursor.p;
    u64* d_totals = (u64*)((uint8_t*)ctx->d_cursor.p + 64);
    u64* d_ibase = (u64*)ctx->d_pbase.p;  // (8-byte aligned first)
    uint32_t* d_pbase = (uint32_t*)(d_ibase + nq + 1);
    QState* d_qstate = (QState*)(d_pbase + nq + 1 + ((nq + 1) & 1));
    {
        // (a stage left pending is read by the consensus kernel before anybody knows whether it fitted its buffers: records the
        // stage did not write must at least be harmless - all-zero when the buffer is new, those of an earlier round otherwise)
        const void* before = ctx->d_mrec.p;
        if (dev_reserve(ctx, ctx->d_mrec, (size_t)st.want_pairs * sizeof(MRec))) return DP_ERR_HIP;
        if (ctx->d_mrec.p != before) DP_HIP(hipMemsetAsync(ctx->d_mrec.p, 0, ctx->d_mrec.cap, ctx->stream));
    }
    if (dev_reserve(ctx, ctx->d_pspec, (size_t)st.want_pairs * sizeof(PSpec))) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_clist, (size_t)st.want_pairs * 8)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_sa, (size_t)st.want_sints * 4)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_sb, (size_t)st.want_sints * 4)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_ma, (size_t)st.want_ints * 4)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_mb, (size_t)st.want_ints * 4)) return DP_ERR_HIP;
    st.pair_cap = (uint32_t)std::min<uint64_t>(0xfffffff0ull, st.want_pairs);
    st.sint_cap = st.want_sints;
    st.int_cap = (uint32_t)std::min<uint64_t>(0xfffffff0ull, st.want_ints);
    ChainArgs A;
    A.qsegs = ctx->qsegs_dev;
    A.qoff = ctx->qoff_dev;
    A.nq = nq;
    A.qsets = (const u64*)ctx->d_qsets.p;
    A.qmeta = st.d_qmeta;
    A.qcnt = st.d_qcnt;
    A.cand = (const u64*)ctx->d_cand.p;
    A.refs = (const dp_seq_ref*)ctx->d_seqrefs.p;
    A.segs = (const int32_t*)ctx->d_segs.p;
    A.seedsets = (const u64*)ctx->d_seedsets.p;
    A.W = W;
    A.SW = SW;
    A.mc = st.d_mc;
    A.mc_n = st.mc_n;
    A.k = st.k;
    A.maxLength = (int)st.max_query_len;
    A.tier = st.chain_tier;
    A.pool = (CNode*)ctx->d_pool.p;
    A.pool_stride = chain_pool_stride(st.max_query_len);
    A.pbase = d_pbase;
    A.ibase = d_ibase;
    A.clist = (uint32_t*)ctx->d_clist.p;
    A.pq = A.clist + st.pair_cap;
    A.pass = 0;
    A.pspec = (PSpec*)ctx->d_pspec.p;
    A.qstate = d_qstate;
    A.recs = (MRec*)ctx->d_mrec.p;
    A.sa = (int32_t*)ctx->d_sa.p;
    A.sb = (int32_t*)ctx->d_sb.p;
    A.ma = (int32_t*)ctx->d_ma.p;
    A.mb = (int32_t*)ctx->d_mb.p;
    {
        // a stage whose consumers are on the device (it stays pending for dp_consensus_paf) leaves the chains in their scratch
        // columns: no cursor atomics, no copies in the walk and resolve kernels (DP_CHAIN_PACK=1: always pack)
        const char* pk = getenv("DP_CHAIN_PACK");  // (read per call: tests switch it between jobs of one process)
        const bool always_pack = pk && pk[0] == '1';
        A.pack = (st.defer_fetch && !always_pack && st.sint_cap < 0xfffffff0ull) ? 0 : 1;
        const char* pe = getenv("DP_CHAIN_PERFECT");  // (read per call: tests switch it between jobs of one process)
        A.walk_always = pe && pe[0] == '0' ? 1 : 0;
        ctx->chains_packed = A.pack != 0;
    }
    A.pair_cap = st.pair_cap;
    A.sint_cap = st.sint_cap;
    A.int_cap = st.int_cap;
    A.cursor = d_cur;
    A.walk_blocks = st.walk_blocks;
    A.walk0_blocks = std::max<uint32_t>(1, std::min<uint32_t>(1024, (nq + S_WAVES - 1) / S_WAVES));
    A.n_refs = std::max<uint32_t>(1, ctx->n_seqs);
    A.prof = nullptr;
    A.prof_walk_slot = 0;
    A.prof_stride = 0;
    static const bool chain_prof = dp_debug("chain_prof");
    if (chain_prof) {
        const size_t pb = ((size_t)st.passes * st.spec_blocks * S_WAVES + (size_t)A.walk0_blocks * S_WAVES) * 128;
        if (dev_reserve(ctx, ctx->d_sched, pb + 128)) return DP_ERR_HIP;
        A.prof = (unsigned long long*)ctx->d_sched.p;
        A.prof_walk_slot = (uint32_t)(st.passes * st.spec_blocks * S_WAVES);
        A.prof_stride = st.spec_blocks * S_WAVES;
        DP_HIP(hipMemsetAsync(A.prof, 0, pb, ctx->stream));
    }
    if (st.attempt > 0) DP_HIP(hipMemsetAsync(ctx->d_cursor.p, 0, C_CURSOR_BYTES, ctx->stream));  // (attempt 0: zeroed with the query stage's buffers)
    DP_HIP(dp_mark(ctx, 6));
    dp_launch<pair_scan_kernel>(ctx, dim3(1), dim3(1024), (const uint32_t*)st.d_qcnt, ctx->qoff_dev, nq, d_pbase, d_ibase, d_totals);
    // mode 0 on the slim layout (a forced tier is the full layout's business)
    const uint32_t spec_blocks = st.spec_blocks;
    if (A.tier == 0 && st.passes > 0) {
        dp_launch<chain_walk_kernel<1>>(ctx, dim3(A.walk0_blocks), dim3(64 * S_WAVES), A, 0);
    } else {
        dp_launch<chain_walk_kernel<0>>(ctx, dim3(st.walk_blocks), dim3(64 * C_WAVES), A, 0);
    }
    for (int ps = 0; ps < st.passes; ps++) {
        A.pass = ps;
        dp_launch<chain_spec_kernel>(ctx, dim3(spec_blocks), dim3(64 * S_WAVES), A, (const u64*)d_totals);
        dp_launch<chain_resolve_kernel>(ctx, dim3(std::min<uint32_t>(1024, (nq + 3) / 4)), dim3(256), A);
    }
    A.pass = st.passes;
    // the final walk rarely has a query to do after the passes: a quarter of the workgroups (each wants 128 KB of a CU's LDS before it
    // can look whether there is anything to do) - a query that is still open finds a wave among 192
    if (st.passes > 0) A.walk_blocks = std::min<uint32_t>(st.walk_blocks, 48);
    dp_launch<chain_walk_kernel<0>>(ctx, dim3(A.walk_blocks), dim3(64 * C_WAVES), A, 2);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 7));
    {
        // the cursor block and the status words, per-query posting-word counts and candidate counts (a few KB) come back in any
        // case, in the same wait: stored into their pinned blocks by one small launch of this stream (dp_zero_fetch_regions works
        // in either direction) instead of two copies handed to the runtime
        // (a stage that stays pending: nobody reads them before the consensus call's wait - its anchors launch, the next one of
        // this stream, stores them instead: dp_match_anchors_launch)
        if (st.defer_fetch && st.attempt == 0) {
            st.fetch_owed = true;
        } else {
            const dp_fetch_region f[2] = {{ctx->h_cursor.p, ctx->d_cursor.p, C_CURSOR_BYTES}, {ctx->h_qm.p, st.d_qmeta, (size_t)nq * 28}};
            if (int rc = dp_zero_fetch_regions(ctx, nullptr, 0, f, 2)) return rc;
        }
    }
    st.attempt++;
    return DP_OK;
}

// after a wait: what did the attempt report?  *grow: a buffer was too small - capacities raised, run it again
static int chain_check(dp_ctx* ctx, FindState& st, bool* grow) {
    memcpy(st.cur, ctx->h_cursor.p, 128);
    st.alg_bytes = 0;
    for (int i = 0; i < 64; i++) st.alg_bytes += ((const uint64_t*)((const uint8_t*)ctx->h_cursor.p + 128))[i];
    st.chain_ms += dp_elapsed(ctx, 6, 7);
    uint64_t tot_pairs, tot_sints;
    memcpy(&tot_pairs, &st.cur[16], 8);
    memcpy(&tot_sints, &st.cur[18], 8);
    if (tot_pairs > 0xfffffff0ull) return dp_fail(ctx, DP_ERR_CAPACITY, "more than 2^32 (query, candidate) pairs in one round");
    *grow = false;
    if (tot_pairs > st.pair_cap) {
        st.want_pairs = tot_pairs + tot_pairs / 2 + 1024;
        *grow = true;
    }
    if (tot_sints > st.sint_cap) {
        st.want_sints = tot_sints + tot_sints / 2 + 1024;
        *grow = true;
    }
    if ((uint64_t)st.cur[0] > st.int_cap) {
        st.want_ints = std::max<uint64_t>(st.want_ints * 2, (uint64_t)st.cur[0] + 1024);
        *grow = true;
    }
    if (!*grow && st.cur[3]) return dp_fail(ctx, DP_ERR_STATE, "dp_find_overlaps: overflow flag without a total that exceeds a buffer");
    return DP_OK;
}

// a checked attempt without overflow: the stage's errors, totals and statistics
static int chain_finish(dp_ctx* ctx, FindState& st) {
    st.pending = false;
    static const bool chain_prof = dp_debug("chain_prof");
    if (chain_prof && ctx->d_sched.p) {
        const size_t waves = (size_t)st.spec_blocks * S_WAVES;
        const size_t wwaves = (size_t)std::max<uint32_t>(1, std::min<uint32_t>(1024, (st.nq + S_WAVES - 1) / S_WAVES)) * S_WAVES;
        std::vector<unsigned long long> h(((size_t)st.passes * waves + wwaves) * 16);
        if (hipMemcpy(h.data(), ctx->d_sched.p, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            for (int ps = -1; ps < st.passes; ps++) {
                unsigned long long sum[16] = {0}, mx[16] = {0}, t0 = ~0ull, t1 = 0, busiest = 0, mostPairs = 0;
                size_t active = 0;
                for (size_t w = 0; w < (ps < 0 ? wwaves : waves); w++) {
                    const unsigned long long* r = &h[16 * (ps < 0 ? (size_t)st.passes * waves + w : (size_t)ps * waves + w)];
                    if (!r[10]) continue;
                    t0 = std::min(t0, r[10]);
                    t1 = std::max(t1, r[11]);
                    if (!r[0]) continue;
                    active++;
                    for (int i = 0; i < 10; i++) sum[i] += r[i];
                    for (int i = 2; i < 10; i++) mx[i] = std::max(mx[i], r[i]);
                    for (int i = 12; i < 16; i++) sum[i] += r[i];
                    busiest = std::max(busiest, r[11] - r[10]);
                    mostPairs = std::max(mostPairs, r[0]);
                }
                if (!sum[0]) continue;
                const double n = (double)sum[0], nc = (double)std::max<unsigned long long>(1, sum[1]);
                fprintf(stderr, "[chain prof] pass %d (-1 = walk 0; its 'records' include the candidate list): %llu pairs (%llu chained) on %zu waves (at most %llu per wave), kernel span %.1f us, busiest wave %.1f us | us per pair: "
                                "records %.2f stage a %.2f prefilter %.2f chain_pair %.2f | per chained pair: stage b + flags %.2f initial %.2f events + walk %.2f (b events %.2f us; %.1f events; %.1f %% perfect chains) out %.2f\n",
                        ps, sum[0], sum[1], active, mostPairs, (t1 - t0) / 100.0, busiest / 100.0, sum[2] / 100.0 / n, sum[3] / 100.0 / n, sum[4] / 100.0 / n,
                        sum[5] / 100.0 / n, sum[6] / 100.0 / nc, sum[7] / 100.0 / nc, sum[8] / 100.0 / nc, sum[12] / 100.0 / nc, sum[13] / nc, 100.0 * sum[14] / nc, sum[9] / 100.0 / nc);
                {
                    unsigned long long w5[5] = {0, 0, 0, 0, 0};
                    for (size_t w = 0; w < (ps < 0 ? wwaves : waves); w++) {
                        const unsigned long long v = h[16 * (ps < 0 ? (size_t)st.passes * waves + w : (size_t)ps * waves + w) + 15];
                        for (int i = 0; i < 5; i++) w5[i] += (v >> (12 * i)) & 0xfff;
                    }
                    fprintf(stderr, "[chain prof]   walked, not perfect: %llu size limits, %llu event 0 starts no / several chains, %llu start beyond maxBIndex, %llu seed or gap, %llu second start\n",
                            w5[0], w5[1], w5[2], w5[3], w5[4]);
                }
                fprintf(stderr, "[chain prof]   largest per-wave sums, us: records %.1f stage a %.1f prefilter %.1f chain_pair %.1f (stage b %.1f initial %.1f walk %.1f) out %.1f\n",
                        mx[2] / 100.0, mx[3] / 100.0, mx[4] / 100.0, mx[5] / 100.0, mx[6] / 100.0, mx[7] / 100.0, mx[8] / 100.0, mx[9] / 100.0);
            }
        }
    }
    if (chain_prof) fprintf(stderr, "[chain prof] final walk: %u pairs chained, %u of them marked for the full layout; open ahead of pass 1 / 2: %u / %u, passes %d\n",
                            st.cur[20], st.cur[21], st.cur[24] + st.cur[25], st.cur[26] + st.cur[27], st.passes);
    st.query_ms = dp_elapsed(ctx, 4, 5);
    st.chain_bytes = st.alg_bytes;
    if (st.cur[2]) {
        char msg[160];
        snprintf(msg, sizeof msg, "overlap chaining hit a reference capacity limit (bits %u: 1 reduced buffer, 2 state pool, 4 results, 8 nodes)", st.cur[2]);
        return dp_fail(ctx, DP_ERR_CAPACITY, msg);
    }
    st.query_bytes = 0;
    {
        const uint32_t* qm = (const uint32_t*)ctx->h_qm.p;
        const u64* words = (const u64*)((const uint8_t*)ctx->h_qm.p + (size_t)st.nq * 16);
        for (uint32_t q = 0; q < st.nq; q++) {
            if (qm[4 * q + 2] & 1) return dp_fail(ctx, DP_ERR_CAPACITY, "query with more than 65535 usable seeds");
            st.query_bytes += words[q] * 8;
        }
    }
    ctx->chain_open_ahead[0] = st.passes >= 1 ? st.cur[24] + st.cur[25] : ~0u;
    ctx->chain_open_ahead[1] = st.passes >= 2 ? st.cur[26] + st.cur[27] : ctx->chain_open_ahead[0];
    uint64_t tp = 0;
    memcpy(&tp, &st.cur[16], 8);
    ctx->n_pairs = (uint32_t)tp;
    ctx->last_nq = st.nq;
    ctx->last_ni = st.cur[0];
    memcpy(&ctx->last_sints, &st.cur[18], 8);
    ctx->last_k = st.k;
    ctx->find_valid = true;
    return DP_OK;
}

// A pending chaining stage after the caller's own wait on the context's stream: evaluates the attempt; when a buffer was too
// small the stage is run again (with waits) until it fits.  *reran: whatever the caller queued behind the first attempt read
// incomplete records and has to be queued again.
int dp_find_complete(dp_ctx* ctx, bool* reran) {
    *reran = false;
    FindState* st = ctx->find_state;
    if (!st || !st->pending) return DP_OK;
    for (;;) {
        bool grow = false;
        if (int rc = chain_check(ctx, *st, &grow)) return rc;
        if (!grow) break;
        *reran = true;
        if (int rc = chain_enqueue(ctx, *st)) return rc;
        DP_HIP(dp_stream_sync(ctx));
    }
    return chain_finish(ctx, *st);
}
void dp_find_stats(const dp_ctx* ctx, double* query_ms, double* chain_ms, uint64_t* query_bytes, uint64_t* chain_bytes) {
    const FindState* st = ctx->find_state;
    *query_ms = st ? st->query_ms : 0;
    *chain_ms = st ? (double)st->chain_ms : 0;
    *query_bytes = st ? st->query_bytes : 0;
    *chain_bytes = st ? st->chain_bytes : 0;
}

int dp_find_overlaps_impl(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t nq, double hf, int k,
                          uint32_t max_query_len, int want_candidates, dp_match_batch* out) {
    if (!ctx->round_open) return dp_fail(ctx, DP_ERR_STATE, "dp_find_overlaps before dp_round_begin");
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    out->n_queries = nq;
    ctx->find_valid = false;
    ctx->n_pairs = 0;
    ctx->last_nq = nq;
    ctx->last_k = k;
    if (ctx->find_state) ctx->find_state->pending = false;
    const uint32_t M = ctx->n_seqs;
    if (pin_reserve(ctx, ctx->h_cursor, C_CURSOR_BYTES)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_cand_off, ((size_t)nq + 1) * 8)) return DP_ERR_HIP;
    ((uint64_t*)ctx->h_cand_off.p)[0] = 0;
    out->cand_off = (const uint64_t*)ctx->h_cand_off.p;
    if (pin_reserve(ctx, ctx->h_moff, 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_qm, (size_t)nq * 28 + 16)) return DP_ERR_HIP;
    ((uint64_t*)ctx->h_moff.p)[0] = 0;
    out->off = (const uint64_t*)ctx->h_moff.p;
    if (nq == 0 || M == 0) {
        for (uint32_t q = 0; q <= nq; q++) ((uint64_t*)ctx->h_cand_off.p)[q] = 0;
        return DP_OK;
    }
    if (!ctx->find_state) ctx->find_state = new FindState();
    FindState& st = *ctx->find_state;
    st = FindState();
    st.nq = nq;
    st.k = k;
    st.max_query_len = max_query_len;
    u64* d_words = nullptr;
    int32_t* d_mc = nullptr;
    if (dev_reserve(ctx, ctx->d_cursor, C_CURSOR_BYTES)) return DP_ERR_HIP;  // [0..63] cursor words, [64..] totals (u64 pairs, u64 scratch ints)
    if (dev_reserve(ctx, ctx->d_pbase, ((size_t)nq + 1) * 4 + ((size_t)nq + 1) * 8 + (size_t)nq * sizeof(QState) + (size_t)nq * 4 + 128)) return DP_ERR_HIP;
    if (int rc = dp_query_stage(ctx, q_segs, q_off, nq, hf, &st.d_qmeta, &d_words, &d_mc, &st.mc_n, &st.d_qcnt)) return rc;
    st.d_mc = d_mc;
    const char* tier_env = getenv("DP_CHAIN_TIER");  // tests: 2 = lds tier, 3 = one-lane tier for every pair
    st.chain_tier = tier_env ? atoi(tier_env) : 0;
    const char* pass_env = getenv("DP_CHAIN_PASSES");  // proposal passes (0 = the serial walk alone: the round-1 behaviour)
    st.passes = pass_env ? std::max(0, std::min(6, atoi(pass_env))) : 3;
    // (round 6) the third pass is left out only when this context's previous stage left it next to nothing.  A threshold of 24 pairs - "at
    // k = 13 it sees ~18 short pairs, which the final walk chains as well" - saved two launches a round and cost more than it saved: the
    // final walk, one wave per QUERY, went from 13 to 53 us a launch under five slots (profiles/r06/k13_against_round5.txt).
    if (!pass_env && ctx->chain_open_ahead[1] < 2u) st.passes = 2;
    st.walk_blocks = std::min<uint32_t>(256, (nq + C_WAVES - 1) / C_WAVES);
    st.spec_blocks = 1024;  // 4096 persistent waves, 16 per CU: what CSlim's 8.5 KB per wave lets a CU hold
    st.spec_blocks = (uint32_t)std::max(1L, dp_tune("spec_blocks", st.spec_blocks));
    if (dev_reserve(ctx, ctx->d_pool, (size_t)st.walk_blocks * C_WAVES * chain_pool_stride(st.max_query_len) * sizeof(CNode))) return DP_ERR_HIP;
    // capacities: what the buffers hold now (at least a floor); a run that needs more reports its totals and is repeated
    // with larger buffers (deterministic: same results)
    st.want_pairs = std::max<uint64_t>(1u << 14, ctx->d_mrec.cap / sizeof(MRec));
    st.want_sints = std::max<uint64_t>(1u << 18, ctx->d_sa.cap / 4);
    st.want_ints = std::max<uint64_t>(1u << 18, ctx->d_ma.cap / 4);
    st.defer_fetch = (want_candidates & 6) == 6;
    if (int rc = chain_enqueue(ctx, st)) return rc;
    if ((want_candidates & 6) == 6) {
        // bit 2: nothing of this call is read before dp_consensus_paf - the stage stays pending and is evaluated in that call's
        // wait (one wait per round less)
        st.pending = true;
        for (uint32_t q = 0; q <= nq; q++) ((uint64_t*)ctx->h_cand_off.p)[q] = 0;
        return DP_OK;
    }
    for (;;) {
        DP_HIP(dp_stream_sync(ctx));
        bool grow = false;
        if (int rc = chain_check(ctx, st, &grow)) return rc;
        if (!grow) break;
        if (int rc = chain_enqueue(ctx, st)) return rc;
    }
    if (int rc = chain_finish(ctx, st)) return rc;
    out->query_kernel_ms = st.query_ms;
    out->chain_kernel_ms = st.chain_ms;
    out->query_bytes = st.query_bytes;
    out->chain_bytes = st.chain_bytes;
    if (want_candidates & 2) {
        for (uint32_t q = 0; q <= nq; q++) ((uint64_t*)ctx->h_cand_off.p)[q] = 0;
        return DP_OK;
    }
    const double qk = out->query_kernel_ms, ck = out->chain_kernel_ms;
    const uint64_t qb = out->query_bytes, cb = out->chain_bytes;
    int rc = dp_fetch_overlaps_impl(ctx, want_candidates & 1, out);
    out->query_kernel_ms = qk;
    out->chain_kernel_ms = ck;
    out->query_bytes = qb;
    out->chain_bytes = cb;
    return rc;
}

// Anchors of every final record of the chaining stage (device resident): d_manchor[2 * pair]
int dp_match_anchors_launch(dp_ctx* ctx, const dp_fetch_region* fetch, uint32_t* zero_word) {
    // (a pending chaining stage: the pair count is on the device only - the launch covers the stage's capacity)
    const uint32_t nslots = dp_find_pending(ctx) ? dp_find_pair_cap(ctx) : ctx->n_pairs;
    if (dev_reserve(ctx, ctx->d_manchor, (size_t)nslots * 16 + 32)) return DP_ERR_HIP;  // anchors [2 * nslots] | covered bases [2 * nslots]
    // what a pending chaining stage still has to report (chain_enqueue): its cursor block and per-query words, into their pinned blocks
    FindState* fs = ctx->find_state;
    dp_fetch_region owed[2] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
    if (fs && fs->fetch_owed) {
        owed[0] = {ctx->h_cursor.p, ctx->d_cursor.p, C_CURSOR_BYTES};
        owed[1] = {ctx->h_qm.p, fs->d_qmeta, (size_t)fs->nq * 28};
        fs->fetch_owed = false;
    }
    if (!nslots) {
        if (owed[0].dst)
            if (int rc = dp_zero_fetch_regions(ctx, nullptr, 0, owed, 2)) return rc;
        const dp_zero_region z = {zero_word, 8};
        return (fetch || zero_word) ? dp_zero_fetch_regions(ctx, &z, zero_word ? 1 : 0, fetch, fetch ? 1 : 0) : DP_OK;
    }
    AnchorFetch F;
    const dp_fetch_region* fr[3] = {fetch, owed[0].dst ? &owed[0] : nullptr, owed[1].dst ? &owed[1] : nullptr};
    for (int i = 0; i < 3; i++) {
        F.dst[i] = fr[i] ? (unsigned long long*)fr[i]->dst : nullptr;
        F.src[i] = fr[i] ? (const unsigned long long*)fr[i]->src : nullptr;
        F.n8[i] = fr[i] ? (unsigned long long)((fr[i]->bytes + 7) / 8) : 0ull;
    }
    const u64* d_totals = (const u64*)((const uint8_t*)ctx->d_cursor.p + 64);
    dp_launch<match_anchor_kernel>(ctx, dim3(std::min<uint32_t>(8192, (nslots + 3) / 4)), dim3(256),
                       (const MRec*)ctx->d_mrec.p, (const uint32_t*)d_totals, nslots, (const int32_t*)dp_chain_b(ctx),
                       (const dp_seq_ref*)ctx->d_seqrefs.p, (const int32_t*)ctx->d_segs.p, ctx->last_k, (int32_t*)ctx->d_manchor.p,
                       F, zero_word, (const int32_t*)dp_chain_a(ctx), (const int32_t*)ctx->qsegs_dev, (const u64*)ctx->qoff_dev,
                       (int32_t*)ctx->d_manchor.p + 2 * (size_t)nslots);
    ctx->mcover_off = 2 * (size_t)nslots;
    DP_HIP(hipGetLastError());
    return DP_OK;
}

// Download of the chaining stage's records.  Pair slots are in canonical order already: queries ascending, candidates
// ascending within a query.
// Chains left in their scratch columns (ChainArgs.pack == 0), for a host consumer after all: copied into ma/mb densely, the
// records' offsets rewritten.  One wave per record; the order of the chains in ma/mb is whatever the atomics make it (the
// records say where each one is).
struct chain_pack_kernel {
    enum { THREADS = 256 };
    static __device__ void run(MRec* __restrict__ recs, uint32_t nslots, const int32_t* __restrict__ sa, const int32_t* __restrict__ sb,
                               int32_t* __restrict__ ma, int32_t* __restrict__ mb, uint32_t* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); slot < nslots; slot += waves) {
        const MRec r = recs[slot];
        if (r.len == 0) continue;
        uint32_t off = 0;
        if (lane == 0) off = atomicAdd(total, r.len);
        off = (uint32_t)__shfl((int)off, 0, 64);
        for (uint32_t x = lane; x < r.len; x += 64) {
            ma[off + x] = sa[r.off + x];
            mb[off + x] = sb[r.off + x];
        }
        if (lane == 0) recs[slot].off = off;
    }
}
};

int dp_fetch_overlaps_impl(dp_ctx* ctx, int want_candidates, dp_match_batch* out) {
    if (!ctx->find_valid) return dp_fail(ctx, DP_ERR_STATE, "dp_fetch_overlaps: no chaining stage output on this context");
    hipSetDevice(ctx->device);
    memset(out, 0, sizeof(*out));
    if (!ctx->chains_packed && ctx->n_pairs) {
        // the stage left its chains in the scratch columns (its consumers were on the device): pack them now
        const uint64_t bound = std::max<uint64_t>(1, ctx->find_state ? std::min<uint64_t>(ctx->find_state->sint_cap, ctx->last_sints) : 0);
        if (bound > 0xfffffff0ull) return dp_fail(ctx, DP_ERR_CAPACITY, "dp_fetch_overlaps: more than 2^32 chain ints");
        if (dev_reserve(ctx, ctx->d_ma, (size_t)bound * 4)) return DP_ERR_HIP;
        if (dev_reserve(ctx, ctx->d_mb, (size_t)bound * 4)) return DP_ERR_HIP;
        uint32_t* d_total = (uint32_t*)ctx->d_cursor.p;  // (the stage's cursor block is free again: word 0 counted packed ints anyway)
        DP_HIP(hipMemsetAsync(d_total, 0, 4, ctx->stream));
        dp_launch<chain_pack_kernel>(ctx, dim3(std::min<uint32_t>(2048, (ctx->n_pairs + 3) / 4)), dim3(256), (MRec*)ctx->d_mrec.p, ctx->n_pairs,
                                     (const int32_t*)ctx->d_sa.p, (const int32_t*)ctx->d_sb.p, (int32_t*)ctx->d_ma.p, (int32_t*)ctx->d_mb.p, d_total);
        DP_HIP(hipGetLastError());
        if (pin_reserve(ctx, ctx->h_cursor, C_CURSOR_BYTES)) return DP_ERR_HIP;
        DP_HIP(hipMemcpyAsync(ctx->h_cursor.p, d_total, 4, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(dp_stream_sync(ctx));
        ctx->last_ni = *(const uint32_t*)ctx->h_cursor.p;
        ctx->chains_packed = true;
    }
    const uint32_t nq = ctx->last_nq, nslots = ctx->n_pairs, ni = ctx->last_ni;
    const int k = ctx->last_k;
    out->n_queries = nq;
    const u64* d_totals = (const u64*)((const uint8_t*)ctx->d_cursor.p + 64);
    if (pin_reserve(ctx, ctx->h_cand_off, ((size_t)nq + 1) * 8)) return DP_ERR_HIP;
    out->cand_off = (const uint64_t*)ctx->h_cand_off.p;
    if (pin_reserve(ctx, ctx->h_mrec, (size_t)nslots * sizeof(MRec) + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_ta, (size_t)ni * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_tb, (size_t)ni * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_manchor, (size_t)nslots * 8 + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_manchor, (size_t)nslots * 8 + 16)) return DP_ERR_HIP;
    const uint32_t* h_qcnt = (const uint32_t*)((const uint8_t*)ctx->h_qm.p + (size_t)nq * 24);
    const int32_t* ta = (const int32_t*)ctx->h_ta.p;
    const int32_t* tb = (const int32_t*)ctx->h_tb.p;
    if (nslots) {
        dp_launch<match_anchor_kernel>(ctx, dim3(std::min<uint32_t>(1024, (nslots + 3) / 4)), dim3(256),
                           (const MRec*)ctx->d_mrec.p, (const uint32_t*)d_totals, nslots, (const int32_t*)ctx->d_mb.p,
                           (const dp_seq_ref*)ctx->d_seqrefs.p, (const int32_t*)ctx->d_segs.p, k, (int32_t*)ctx->d_manchor.p,
                           AnchorFetch{{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}}, (uint32_t*)nullptr,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, (const u64*)nullptr, (int32_t*)nullptr);
        DP_HIP(hipGetLastError());
        DP_HIP(hipMemcpyAsync(ctx->h_manchor.p, ctx->d_manchor.p, (size_t)nslots * 8, hipMemcpyDeviceToHost, ctx->stream));
        DP_HIP(hipMemcpyAsync(ctx->h_mrec.p, ctx->d_mrec.p, (size_t)nslots * sizeof(MRec), hipMemcpyDeviceToHost, ctx->stream));
        if (ni) {
            DP_HIP(hipMemcpyAsync(ctx->h_ta.p, ctx->d_ma.p, (size_t)ni * 4, hipMemcpyDeviceToHost, ctx->stream));
            DP_HIP(hipMemcpyAsync(ctx->h_tb.p, ctx->d_mb.p, (size_t)ni * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (want_candidates) {
        if (pin_reserve(ctx, ctx->h_cand_list, (size_t)nslots * 4 + 16)) return DP_ERR_HIP;
        if (nslots) DP_HIP(hipMemcpyAsync(ctx->h_cand_list.p, ctx->d_clist.p, (size_t)nslots * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    DP_HIP(dp_stream_sync(ctx));
    MRec* recs = (MRec*)ctx->h_mrec.p;
    uint32_t nm = 0;
    uint64_t total_len = 0;
    for (uint32_t i = 0; i < nslots; i++)
        if (recs[i].len) {
            nm++;
            total_len += recs[i].len;
        }
    if (pin_reserve(ctx, ctx->h_ma, (size_t)total_len * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mb, (size_t)total_len * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mq, (size_t)nm * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_mt, (size_t)nm * 4 + 16)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_moff, ((size_t)nm + 1) * 8)) return DP_ERR_HIP;
    if (pin_reserve(ctx, ctx->h_manout, (size_t)nm * 8 + 16)) return DP_ERR_HIP;
    uint32_t* mq = (uint32_t*)ctx->h_mq.p;
    uint32_t* mt = (uint32_t*)ctx->h_mt.p;
    uint64_t* moff = (uint64_t*)ctx->h_moff.p;
    int32_t* fa = (int32_t*)ctx->h_ma.p;
    int32_t* fb = (int32_t*)ctx->h_mb.p;
    const int32_t* anch = (const int32_t*)ctx->h_manchor.p;
    int32_t* anchOut = (int32_t*)ctx->h_manout.p;
    uint64_t pos = 0;
    uint32_t j = 0;
    for (uint32_t i = 0; i < nslots; i++) {
        const MRec& r = recs[i];
        if (!r.len) continue;
        mq[j] = r.q;
        mt[j] = r.t;
        anchOut[2 * j] = anch[2 * (size_t)i];
        anchOut[2 * j + 1] = anch[2 * (size_t)i + 1];
        moff[j] = pos;
        memcpy(fa + pos, ta + r.off, (size_t)r.len * 4);
        memcpy(fb + pos, tb + r.off, (size_t)r.len * 4);
        pos += r.len;
        j++;
    }
    moff[nm] = pos;
    out->n_matches = nm;
    out->query = mq;
    out->target = mt;
    out->off = moff;
    out->match_a = fa;
    out->match_b = fb;
    out->target_anchor = anchOut;
    uint64_t* co = (uint64_t*)ctx->h_cand_off.p;
    if (want_candidates) {
        uint64_t p2 = 0;
        for (uint32_t q = 0; q < nq; q++) {
            co[q] = p2;
            p2 += h_qcnt[q];
        }
        co[nq] = p2;
        out->cand = (const uint32_t*)ctx->h_cand_list.p;
    } else {
        for (uint32_t q = 0; q <= nq; q++) co[q] = 0;
    }
    return DP_OK;
}

extern "C" int dp_fetch_overlaps(dp_ctx* ctx, dp_match_batch* out) {
    if (!ctx || !out) return DP_ERR_ARG;
    return dp_fetch_overlaps_impl(ctx, 0, out);
}

extern "C" int dp_find_overlaps(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t n_queries,
                                double hit_fraction, int k, uint32_t max_query_len, int want_candidates, dp_match_batch* out) {
    if (!ctx || !out || (n_queries && (!q_segs || !q_off))) return DP_ERR_ARG;
    return dp_find_overlaps_impl(ctx, q_segs, q_off, n_queries, hit_fraction, k, max_query_len, want_candidates, out);
}

// A19/A20 map flavour: implemented in dp_map.hip
extern "C" int dp_map_windows(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len,
                              uint32_t n_windows, int k, dp_chain_batch* out) {
    if (!ctx || !out || (n_windows && (!w_segs || !w_off || !w_len))) return DP_ERR_ARG;
    return dp_map_windows_impl(ctx, w_segs, w_off, w_len, n_windows, k, out);
}

extern "C" int dp_map_windows_shard(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len, uint32_t n_windows,
                                    int k, int phase, int32_t* thr_io, dp_chain_batch* out) {
    if (!ctx || !out || !thr_io || (phase != 0 && phase != 1) || (n_windows && (!w_segs || !w_off || !w_len))) return DP_ERR_ARG;
    return dp_map_windows_impl(ctx, w_segs, w_off, w_len, n_windows, k, out, phase, thr_io);
}
