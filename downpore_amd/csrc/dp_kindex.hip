// libdownpore_hip.so — resident k-mer position index: an alternative to re-scanning every read in every round.
// Built once per read set and k (counting sort of all k-mer start positions by k-mer value, 8 B per base resident in
// HBM — what the 288 GB are for), it turns the round's scan (A2/A10) into: look the round's seed k-mers up, sort their
// few hundred thousand occurrences by position, and cut that list by item ranges.  Output is byte-identical to the scan
// kernels' (counts, survivor compaction and segments go through the same code after the counting step).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "dp_common.h"
#include "dp_launch.h"

struct dp_kindex {
    std::mutex mu;
    int k = 0;
    bool built = false;
    bool unavailable = false;  // k too large for a direct-addressed table, or not enough free HBM: callers scan instead
    uint64_t n_pos = 0;
    DevBuf off;  // uint64 [4^k + 1]
    float build_ms = 0;  // device time of the sorted build
    DevBuf pos;  // [n_pos] k-mer starts grouped by k-mer value: (read << 32 | position in the read) in 64 bits (pos_fmt 8), or
                 // (read << pbits | position) in 32 bits (4) / 40 bits with the top byte in pos_hi (5): KxPos, dp_common.h
    DevBuf pos_hi;
    int pos_fmt = 8, pbits = 32;
    KxPos view() const { return KxPos{pos.p, (const uint8_t*)pos_hi.p, (uint32_t)pos_fmt, (uint32_t)pbits}; }
};

// every k-mer start of every read: slot = off[kmer] + (arrival rank inside its bucket); order inside a bucket is arbitrary
__global__ void kidx_scatter_kernel(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                    const uint32_t* __restrict__ len, uint32_t n_reads, int k, const uint64_t* __restrict__ off,
                                    uint32_t* __restrict__ cursor, uint64_t* __restrict__ pos) {
    const int lane = dp_lane();
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int sh = 32 - 2 * k;
    for (uint32_t r = gw; r < n_reads; r += waves) {
        const uint32_t L = len[r];
        if (L < (uint32_t)k) continue;
        const uint64_t a0 = boff[r] * 4, a1 = a0 + (L - k + 1);
        const uint64_t g0 = a0 >> 5, g1 = (a1 - 1) >> 5;
        for (uint64_t gb = g0; gb <= g1; gb += 64) {
            const uint64_t g = gb + lane;
            if (g > g1) continue;
            const uint32_t* p = (const uint32_t*)(packed + g * 8);
            const uint32_t w0 = __builtin_bswap32(p[0]), w1 = __builtin_bswap32(p[1]), w2 = __builtin_bswap32(p[2]);
            for (int j = 0; j < 32; j++) {
                const uint64_t a = g * 32 + (uint64_t)j;
                if (a < a0 || a >= a1) continue;
                const int q = j >> 4, rr = j & 15;
                const uint32_t hi = q ? w1 : w0, lo = q ? w2 : w1;
                const uint32_t win = rr ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * rr) : hi;
                const uint32_t kmer = win >> sh;
                const uint32_t slot = atomicAdd(&cursor[kmer], 1u);
                pos[off[kmer] + slot] = ((uint64_t)r << 32) | (uint64_t)(a - a0);
            }
        }
    }
}

__global__ void kidx_count_kernel(const uint8_t* __restrict__ packed, const uint64_t* __restrict__ boff,
                                  const uint32_t* __restrict__ len, uint32_t n_reads, int k, uint32_t* __restrict__ counts) {
    const int lane = dp_lane();
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int sh = 32 - 2 * k;
    for (uint32_t r = gw; r < n_reads; r += waves) {
        const uint32_t L = len[r];
        if (L < (uint32_t)k) continue;
        const uint64_t a0 = boff[r] * 4, a1 = a0 + (L - k + 1);
        const uint64_t g0 = a0 >> 5, g1 = (a1 - 1) >> 5;
        for (uint64_t gb = g0; gb <= g1; gb += 64) {
            const uint64_t g = gb + lane;
            if (g > g1) continue;
            const uint32_t* p = (const uint32_t*)(packed + g * 8);
            const uint32_t w0 = __builtin_bswap32(p[0]), w1 = __builtin_bswap32(p[1]), w2 = __builtin_bswap32(p[2]);
            for (int j = 0; j < 32; j++) {
                const uint64_t a = g * 32 + (uint64_t)j;
                if (a < a0 || a >= a1) continue;
                const int q = j >> 4, rr = j & 15;
                const uint32_t hi = q ? w1 : w0, lo = q ? w2 : w1;
                const uint32_t win = rr ? __builtin_amdgcn_alignbit(hi, lo, 32 - 2 * rr) : hi;
                atomicAdd(&counts[win >> sh], 1u);
            }
        }
    }
}

static dp_ctx* kidx_owner(dp_ctx* ctx) { return ctx->owner ? ctx->owner : ctx; }

// ---- round 5: the index built in shares, one per rank, and all-gathered ---------------------------------------------------------
__global__ void kidx_off_rebase(uint64_t* __restrict__ off, uint64_t kmer_lo, uint64_t kmer_hi, uint64_t add, uint64_t nk, uint64_t total) {
    const uint64_t i = kmer_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kmer_hi) off[i] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) off[nk] = total;
}
// After dp_kindex_build_sorted with a shard: *d_pos / *d_pos_hi hold this rank's entries, `off` the bucket starts of its k-mers
// relative to its first entry, `counts` its k-mers' counts.  Afterwards all four describe the whole index (the entry arrays are
// replaced by full-size ones), identical on every rank up to the order of the entries inside a bucket.  Collective.
static int kidx_gather_shares(dp_ctx* ctx, dp_comm* comm, const dp_kindex_shard& sh, int k, int fmt, uint32_t* counts, uint64_t* off,
                              void** d_pos, void** d_pos_hi, uint64_t* n_pos) {
    const int N = sh.n_ranks, me = sh.rank;
    const uint64_t nk = (uint64_t)1 << (2 * k);
    std::vector<uint64_t> kfirst((size_t)N + 1);
    for (int q = 0; q <= N; q++) kfirst[(size_t)q] = std::min<uint64_t>(nk, (uint64_t)sh.digit_first[(size_t)q] << sh.kmer_shift);
    const size_t elem = fmt == 8 ? 8 : 4;
    void *full = nullptr, *full_hi = nullptr;
    // (a rank that leaves with an error must not leave its peers waiting in the collectives below: dp_comm.hip's rule)
    auto fail = [&](int rc) {
        if (full) dp_dev_free(full);
        if (full_hi) dp_dev_free(full_hi);
        dp_comm_abort(comm);
        return rc;
    };
    if (dp_dev_malloc(&full, sh.total * elem + 64) != hipSuccess) return fail(dp_fail(ctx, DP_ERR_HIP, "k-mer position index: no memory for the gathered entries"));
    if (fmt == 5 && dp_dev_malloc(&full_hi, sh.total + 64) != hipSuccess) return fail(dp_fail(ctx, DP_ERR_HIP, "k-mer position index: no memory for the gathered entries"));
    if (int rc = dp_comm_allgather_ranges(comm, ctx, full, elem, sh.entry_first.data(), *d_pos)) return fail(rc);
    if (fmt == 5)
        if (int rc = dp_comm_allgather_ranges(comm, ctx, full_hi, 1, sh.entry_first.data(), *d_pos_hi)) return fail(rc);
    const uint64_t mine_k = kfirst[(size_t)me + 1] - kfirst[(size_t)me];
    hipLaunchKernelGGL(kidx_off_rebase, dim3((unsigned)std::max<uint64_t>(1, (mine_k + 255) / 256)), dim3(256), 0, ctx->stream, off, kfirst[(size_t)me],
                       kfirst[(size_t)me + 1], sh.entry_first[(size_t)me], nk, sh.total);
    if (hipGetLastError() != hipSuccess) return fail(dp_fail(ctx, DP_ERR_HIP, "kidx_off_rebase"));
    if (int rc = dp_comm_allgather_ranges(comm, ctx, off, 8, kfirst.data(), nullptr)) return fail(rc);
    if (int rc = dp_comm_allgather_ranges(comm, ctx, counts, 4, kfirst.data(), nullptr)) return fail(rc);
    dp_dev_free(*d_pos);
    if (*d_pos_hi) dp_dev_free(*d_pos_hi);
    *d_pos = full;
    *d_pos_hi = full_hi;
    *n_pos = sh.total;
    return DP_OK;
}

// Builds the index for k on the context that owns the reads (once; callers on borrowing contexts wait on its mutex).
// Returns DP_OK when the index is ready, 1 when it cannot be used for this k / read set (the caller scans), <0 on error.
int dp_kindex_ensure(dp_ctx* ctx, int k) {
    dp_ctx* ow = kidx_owner(ctx);
    if (!ow->kidx) {
        static std::mutex create_mu;
        std::lock_guard<std::mutex> lk(create_mu);
        if (!ow->kidx) ow->kidx = new dp_kindex();
    }
    dp_kindex* ix = ow->kidx;
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->k == k && (ix->built || ix->unavailable)) return ix->built ? DP_OK : 1;
    hipSetDevice(ctx->device);
    ix->built = false;
    ix->unavailable = false;
    ix->k = k;
    const size_t nk = (size_t)1 << (2 * k);
    // round 5, multi-GPU: every rank sorts the k-mers of its own share of the first-digit buckets and the shares are all-gathered
    // (RCCL over xGMI, device to device) - 1 / N of the build's three passes per rank instead of all of them on every rank.
    // DP_KINDEX_SHARD=0: every rank builds everything, as until round 4.
    // (DP_KINDEX_SHARD=force: also with a communicator of ONE rank - the test hook that takes the RCCL flavour of the gather, grouped
    // ncclBroadcasts between device buffers, through a real librccl on a one-GPU box)
    const char* se = getenv("DP_KINDEX_SHARD");
    // Over an RCCL communicator the shares are OPT-IN (DP_KINDEX_SHARD=1): their exchange - grouped ncclBroadcasts with several roots -
    // has run between in-process ranks (peer copies) and through a one-rank librccl, never yet between two GPUs (no multi-GPU node has
    // been available to this repository); until a run with two ranks has shown the index digests equal, every rank of a multi-process
    // job builds the whole index itself, as until round 4.
    const bool opt_in = se && (se[0] == '1' || se[0] == 'f');
    const bool sharded = ow->kx_comm && (dp_comm_size(ow->kx_comm) > 1 || (se && se[0] == 'f')) && !(se && se[0] == '0') &&
                         (opt_in || !dp_comm_is_rccl(ow->kx_comm));
    // With a communicator this call is COLLECTIVE, and what decides a rank's way through it is the rank's own (free HBM differs per
    // GPU: parked cache blocks, other processes; an allocation fails on one rank only).  Every verdict a rank reaches on its own - no room
    // for the index, an allocation that failed, a share that could not be built - goes into the byte the ranks exchange BEFORE anybody
    // leaves: everyone gathers, or everyone builds alone / falls back, or everyone fails; nobody waits for a rank that has left.
    bool mem_ok = true;
    {
        // resident: 8 B per k-mer start + 8 B per table entry; transient: two 4-byte count tables.  Leave 4 GiB for the rounds.
        size_t free_b = 0, total_b = 0;
        hipMemGetInfo(&free_b, &total_b);
        free_b += dp_dev_cached_bytes();
        // (the sorted build holds a second 8 B per base while it runs; when that does not fit, the atomic scatter path is
        // what is left and needs only the index itself)
        const uint64_t need = ow->total_bases * 8 + (uint64_t)nk * 16 + ((uint64_t)4 << 30);
        int max_k = 14;
        if (const char* e = getenv("DP_KINDEX_MAX_K")) max_k = std::min(14, atoi(e));
        mem_ok = !(k > max_k || need > free_b + ix->pos.cap + ix->off.cap);
        if (!mem_ok && !sharded) {
            ix->unavailable = true;
            return 1;
        }
    }
    // Preferred: the radix-sort build (dp_kbuild.hip) - every pass streams HBM with coalesced traffic, and the k-mer
    // histogram falls out of its last pass (kept for dp_kmer_values).  DP_KINDEX_ATOMIC=1 or an unsupported k: the
    // count -> offsets -> atomic scatter below.
    {
        void* d_cnt = nullptr;
        int pre = DP_OK;  // this rank's own verdict before the build: 0 go on, 1 no room (scan instead), < 0 an allocation failed
        if (!mem_ok) pre = 1;
        else if (dev_reserve(ctx, ix->off, (nk + 1) * 8)) pre = DP_ERR_HIP;
        else if (dp_dev_malloc(&d_cnt, nk * 4) != hipSuccess) pre = dp_fail(ctx, DP_ERR_HIP, "k-mer position index: no memory for the count table");
        if (pre != DP_OK && !sharded) return pre;
        void *d_pos = nullptr, *d_pos_hi = nullptr;
        uint64_t n_pos = 0;
        float ms = 0;
        int fmt = 8, pbits = 32;
        dp_kindex_shard shard;
        if (sharded) {
            shard.rank = dp_comm_rank(ow->kx_comm);
            shard.n_ranks = dp_comm_size(ow->kx_comm);
        }
        int rc = pre != DP_OK ? pre
                              : dp_kindex_build_sorted(ctx, ow, k, (uint32_t*)d_cnt, (uint64_t*)ix->off.p, &d_pos, &d_pos_hi, &fmt, &pbits, &n_pos, &ms,
                                                       sharded ? &shard : nullptr);
        if (sharded) {
            // the ranks agree on how it went before anybody waits in a collective for a rank that fell back
            const uint8_t mine = (uint8_t)(rc == 0 ? 0 : rc > 0 ? 1 : 2);
            const uint8_t* all = nullptr;
            const uint64_t* sizes = nullptr;
            if (int xr = dp_allgather_blobs(ow->kx_comm, ctx, &mine, 1, &all, &sizes)) {  // (aborts the communicator itself)
                if (rc == 0) {
                    dp_dev_free(d_pos);
                    if (d_pos_hi) dp_dev_free(d_pos_hi);
                }
                if (d_cnt) dp_dev_free(d_cnt);
                return xr;
            }
            uint8_t worst = 0;
            for (int q = 0; q < shard.n_ranks; q++) worst = std::max(worst, all[q]);
            if (worst == 0) {
                rc = kidx_gather_shares(ctx, ow->kx_comm, shard, k, fmt, (uint32_t*)d_cnt, (uint64_t*)ix->off.p, &d_pos, &d_pos_hi, &n_pos);
            } else {
                if (rc == 0) {
                    dp_dev_free(d_pos);
                    if (d_pos_hi) dp_dev_free(d_pos_hi);
                    d_pos = d_pos_hi = nullptr;
                }
                // (a rank could not build its share: everybody builds the whole index on its own - no collective in that - or, where a
                // rank FAILED, falls back alike; the rank without room scans, as it would have without a communicator)
                rc = worst == 2 && rc >= 0 ? 1 : rc;
                if (rc >= 0 && pre == DP_OK)
                    rc = dp_kindex_build_sorted(ctx, ow, k, (uint32_t*)d_cnt, (uint64_t*)ix->off.p, &d_pos, &d_pos_hi, &fmt, &pbits, &n_pos, &ms, nullptr);
            }
            if (pre == 1 && rc >= 0) {
                ix->unavailable = true;
                return 1;
            }
        }
        if (rc < 0) {
            if (d_cnt) dp_dev_free(d_cnt);
            return rc;
        }
        if (rc == 0) {
            if (ix->pos.p) dp_dev_free(ix->pos.p);
            if (ix->pos_hi.p) dp_dev_free(ix->pos_hi.p);
            ix->pos.p = d_pos;
            ix->pos.cap = n_pos * (fmt == 8 ? 8 : 4) + 64;
            ix->pos_hi.p = d_pos_hi;
            ix->pos_hi.cap = d_pos_hi ? n_pos + 64 : 0;
            ix->pos_fmt = fmt;
            ix->pbits = pbits;
            ix->n_pos = n_pos;
            ix->built = true;
            ix->build_ms = ms;
            if (ow->d_kcounts) dp_dev_free(ow->d_kcounts);
            ow->d_kcounts = d_cnt;  // the histogram of exactly these k-mers: dp_kmer_values takes it from here
            ow->kcounts_k = k;
            return DP_OK;
        }
        if (d_cnt) dp_dev_free(d_cnt);
    }
    // count -> offsets -> scatter, on the CALLER's stream (the owner's buffers are only written here, under the mutex)
    void *d_counts = nullptr, *d_tmp = nullptr, *d_counts1 = nullptr;
    struct Temps {  // released on every way out, error returns included
        void **a, **b, **c;
        ~Temps() {
            for (void** p : {a, b, c})
                if (*p) dp_dev_free(*p);
        }
    } temps{&d_counts, &d_tmp, &d_counts1};
    if (ow->d_kcounts && ow->kcounts_k == k) {  // dp_kmer_values counted exactly these k-mers already
        d_counts = ow->d_kcounts;
        ow->d_kcounts = nullptr;
        ow->kcounts_k = 0;
    } else {
        DP_HIP(dp_dev_malloc(&d_counts, nk * 4));
        DP_HIP(hipMemsetAsync(d_counts, 0, nk * 4, ctx->stream));
        if (ow->n_reads)
            hipLaunchKernelGGL(kidx_count_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const uint8_t*)ow->d_packed.p,
                               (const uint64_t*)ow->d_boff.p, (const uint32_t*)ow->d_len.p, ow->n_reads, k, (uint32_t*)d_counts);
        DP_HIP(hipGetLastError());
    }
    if (dev_reserve(ctx, ix->off, (nk + 1) * 8)) return DP_ERR_HIP;
    size_t tmp_bytes = 0;
    rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t*)d_counts, (uint64_t*)ix->off.p, (uint64_t)0, nk + 1, rocprim::plus<uint64_t>(),
                            ctx->stream);
    DP_HIP(dp_dev_malloc(&d_tmp, tmp_bytes + 16));
    // nk + 1 outputs: the extra input element is never added into an output, but it must be readable -> the buffer holds nk+1
    DP_HIP(dp_dev_malloc(&d_counts1, (nk + 1) * 4));
    DP_HIP(hipMemcpyAsync(d_counts1, d_counts, nk * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DP_HIP(hipMemsetAsync((uint8_t*)d_counts1 + nk * 4, 0, 4, ctx->stream));
    DP_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, (uint32_t*)d_counts1, (uint64_t*)ix->off.p, (uint64_t)0, nk + 1,
                                   rocprim::plus<uint64_t>(), ctx->stream));
    uint64_t total = 0;
    DP_HIP(hipMemcpyAsync(&total, (uint64_t*)ix->off.p + nk, 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    if (dev_reserve(ctx, ix->pos, total * 8 + 64)) return DP_ERR_HIP;
    ix->pos_fmt = 8;
    ix->pbits = 32;
    DP_HIP(hipMemsetAsync(d_counts, 0, nk * 4, ctx->stream));
    if (ow->n_reads)
        hipLaunchKernelGGL(kidx_scatter_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const uint8_t*)ow->d_packed.p,
                           (const uint64_t*)ow->d_boff.p, (const uint32_t*)ow->d_len.p, ow->n_reads, k, (const uint64_t*)ix->off.p,
                           (uint32_t*)d_counts, (uint64_t*)ix->pos.p);
    DP_HIP(hipGetLastError());
    DP_HIP(dp_stream_sync(ctx));
    ix->n_pos = total;
    ix->k = k;
    ix->built = true;
    return DP_OK;
}

extern "C" int dp_kindex_set_comm(dp_ctx* ctx, dp_comm* comm) {
    if (!ctx) return DP_ERR_ARG;
    kidx_owner(ctx)->kx_comm = comm;
    return DP_OK;
}

__device__ __forceinline__ unsigned long long kidx_mix(unsigned long long x) {  // splitmix64's finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ void kidx_digest_kernel(const uint64_t* __restrict__ off, const KxPos pos, uint64_t nk, unsigned long long* __restrict__ out) {
    unsigned long long a = 0, b = 0;
    for (uint64_t km = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; km < nk; km += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o0 = off[km], o1 = off[km + 1];
        a += kidx_mix(km * 0x9e3779b97f4a7c15ull + o0);
        for (uint64_t i = o0; i < o1; i++) b += kidx_mix(kidx_mix(km + 1) ^ kx_entry(pos, i));
    }
    atomicAdd(&out[1], a);
    atomicAdd(&out[2], b);
}
extern "C" int dp_kindex_digest(dp_ctx* ctx, int k, uint64_t* out) {
    if (!ctx || !out) return DP_ERR_ARG;
    dp_kindex* ix = kidx_owner(ctx)->kidx;
    if (!ix || !ix->built || ix->k != k) return dp_fail(ctx, DP_ERR_STATE, "dp_kindex_digest: no index built for this k");
    hipSetDevice(ctx->device);
    unsigned long long* d = nullptr;
    DP_HIP(dp_dev_malloc((void**)&d, 64));
    DP_HIP(hipMemsetAsync(d, 0, 64, ctx->stream));
    hipLaunchKernelGGL(kidx_digest_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const uint64_t*)ix->off.p, ix->view(), (uint64_t)1 << (2 * k), d);
    DP_HIP(hipGetLastError());
    DP_HIP(hipMemcpyAsync(out, d, 24, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    out[0] = ix->n_pos;
    dp_dev_free(d);
    return DP_OK;
}

void dp_kindex_free(dp_ctx* ctx) {
    if (ctx->owner || !ctx->kidx) return;
    if (ctx->kidx->off.p) dp_dev_free(ctx->kidx->off.p);
    if (ctx->kidx->pos.p) dp_dev_free(ctx->kidx->pos.p);
    if (ctx->kidx->pos_hi.p) dp_dev_free(ctx->kidx->pos_hi.p);
    delete ctx->kidx;
    ctx->kidx = nullptr;
}

// ---- per round -------------------------------------------------------------------------------------------------------
// The counting step of a round, straight from the index (no sort, no host round trip before the survivors are known):
//   kidx_count   every occurrence of every seed k-mer bumps the counter of the read item it falls into and of the extra
//                items (query windows) of its read;
//   kidx_offsets one single-pass scan (decoupled look-back) over the items: segment offsets of the survivors, their
//                compacted list, the totals;
//   kidx_fill    the occurrences of surviving items go, unordered, into the item's own segment slice as (position, seed);
//   kidx_sortwrite  one wave per survivor sorts its slice by position and turns it into [gap, seed, ..., gap] in place.
// Since round 5 the default form of the first three is (see "round 5" below): kidx_walk_bin writes every hit as a record into the bin of
// its read and touches no counter, kidx_offsets' tiles count their own bins' records in LDS before they scan, kidx_bin_fill places the
// survivors' records with LDS slot counters; the kernels named above remain as DP_KX_BINS=0 and as the fallback of a round whose bins
// overflowed.

#define KX_PARTS 4  // waves that share one seed's bucket

// extra items (query windows) of a read form a linked list: head[read] = item + 1, next[item] = previous head
__global__ void kidx_link_extra(const dp_scan_item* __restrict__ items, uint32_t n_read_items, uint32_t n_extra,
                                uint32_t* __restrict__ head, uint32_t* __restrict__ next) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_extra) return;
    next[e] = atomicExch(&head[items[n_read_items + e].read], e + 1);
}
// start of a round's counting step in one launch: the work area (fill cursors, tile status, ticket, hit slots), the item
// counters and the totals go to zero, and the extra items are linked to their reads (head[] is all zero between calls)
struct kidx_prepare {
    enum { THREADS = 256 };
    static __device__ void run(uint32_t* __restrict__ work, uint32_t n_work, uint32_t* __restrict__ counts, uint32_t n_counts,
                             uint32_t* __restrict__ totals16, dp_scan_item* __restrict__ items, uint32_t n_read_items,
                             uint32_t n_extra, uint32_t* __restrict__ head, uint32_t* __restrict__ next,
                             const dp_scan_item* __restrict__ extra_src) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_work) work[i] = 0;
    if (i < n_counts) counts[i] = 0;
    if (i < 16) totals16[i] = 0;
    if (i < n_extra) {
        // extra_src: the round's extra items still sit in the caller's pinned staging block - this thread brings its item over
        // (no upload of its own for ten kilobytes)
        dp_scan_item it = extra_src ? extra_src[i] : items[n_read_items + i];
        if (extra_src) items[n_read_items + i] = it;
        next[i] = atomicExch(&head[it.read], i + 1);
    }
}
};
struct kidx_unlink_extra {
    enum { THREADS = 256 };
    static __device__ void run(const dp_scan_item* __restrict__ items, uint32_t n_read_items, uint32_t n_extra,
                                  uint32_t* __restrict__ head) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_extra) return;
    head[items[n_read_items + e].read] = 0;
}
};

#ifdef DP_PROF_BUILD
#define KX_PROFILING 1
#else
#define KX_PROFILING 0
#endif
// ---- hit records of the count pass (round 4) -------------------------------------------------------------------------
// [63] the read carries extra items (query windows)  [62] the hit counts for its read item (in range, inside the item)
// [61:38] read  [37:24] rank among the read's hits (the counter's value before this hit)  [23:0] position
#define KX_REC_X (1ull << 63)
#define KX_REC_V (1ull << 62)
struct KxRec {
    unsigned long long* rec;  // [64 * shard_cap] (null: no records this round)
    uint32_t* kb;             // [2 * groups]: where the group's records start (0xffffffff: its shard was full), how many
    uint32_t* flags;          // [0] a shard was full
    uint32_t shard_cap;
    // the count walk's short cut (round 4): views served whole (not top-level) have a k-mer at every indexed position, so "inside
    // the item" is "the read is not ignored" - a byte per read instead of its 16-byte item - and only reads in [qlo, qlo + qspan)
    // carry extra items (the round's query reads are consecutive): two dependent table reads per hit fewer.  ign == null: off.
    const uint8_t* ign;
    uint32_t qlo, qspan;
};
__device__ __forceinline__ unsigned long long kx_rec(bool x, bool v, uint32_t r, uint32_t rank, uint32_t p) {
    return (x ? KX_REC_X : 0ull) | (v ? KX_REC_V : 0ull) | ((unsigned long long)(r & 0xffffffu) << 38) |
           ((unsigned long long)(rank & 0x3fffu) << 24) | (unsigned long long)(p & 0xffffffu);
}

// the fill pass from records: same thread layout as the count pass (a group's lanes over its stretch), no bucket walk
struct kidx_fill_rec {
    enum { THREADS = 256 };
    static __device__ void run(const KxRec R, uint32_t n_groups, uint32_t lps, const dp_scan_item* __restrict__ items, uint32_t lo,
                               uint32_t n_read_items, uint32_t min_seeds, const uint32_t* __restrict__ head,
                               const uint32_t* __restrict__ next, const uint32_t* __restrict__ counts, uint32_t* __restrict__ fillc,
                               const uint64_t* __restrict__ segoff, int32_t* __restrict__ segs, const uint64_t* __restrict__ totals,
                               uint64_t seg_cap) {
    if (R.flags[0] || totals[0] > seg_cap) return;  // (the host repeats the fill with the bucket walk / a larger buffer)
    const int lane = dp_lane();
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t grp, s, j0, step;
    if (lps == 64) {
        grp = w;
        s = w / KX_PARTS;
        j0 = 4u * (uint32_t)lane;  // (four consecutive records per lane and trip: two 16-byte loads; the buffer ends in 64 bytes of slack)
        step = 64;
    } else {
        grp = w * 4 + ((uint32_t)lane >> 4);
        s = grp;
        j0 = 4u * ((uint32_t)lane & 15u);
        step = 16;
    }
    if (grp >= n_groups) return;
    const uint32_t at = R.kb[2 * (size_t)grp], gn = R.kb[2 * (size_t)grp + 1];
    if (at == 0xffffffffu) return;
    for (uint32_t jb = j0; jb < gn; jb += 4 * step) {
        unsigned long long e[4];
        {
            struct __attribute__((packed, aligned(8))) Rec4 {
                unsigned long long a, b, c, d;
            };
            const Rec4 q = *(const Rec4*)(R.rec + (size_t)at + jb);
            e[0] = q.a;
            e[1] = jb + 1 < gn ? q.b : 0ull;
            e[2] = jb + 2 < gn ? q.c : 0ull;
            e[3] = jb + 3 < gn ? q.d : 0ull;
        }
        uint32_t cnt[4];
        uint64_t so[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t it = (uint32_t)(e[u] >> 38) & 0xffffffu;
            const bool val = (e[u] & KX_REC_V) != 0;
            cnt[u] = val ? counts[it - lo] : 0u;
            so[u] = val ? segoff[it - lo] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(e[u] >> 38) & 0xffffffu, p = (uint32_t)e[u] & 0xffffffu;
            if ((e[u] & KX_REC_V) && cnt[u] >= min_seeds) {
                const uint64_t to = so[u] + 2ull * ((uint32_t)(e[u] >> 24) & 0x3fffu);
                segs[to] = (int32_t)p;
                segs[to + 1] = (int32_t)s;
            }
            if (e[u] & KX_REC_X) {
                for (uint32_t x = head[r]; x; x = next[x - 1]) {  // the round's extra items on this read (query windows)
                    const uint32_t it = n_read_items + x - 1;
                    const dp_scan_item xi = items[it];
                    if (p - xi.start < xi.n_kmers && p >= xi.start && counts[it] >= xi.min_seeds) {
                        const uint32_t slot = atomicAdd(&fillc[it], 1u);
                        const uint64_t to = segoff[it] + 2ull * slot;
                        segs[to] = (int32_t)(p - xi.start);
                        segs[to + 1] = (int32_t)s;
                    }
                }
            }
        }
    }
}
};

template <bool FILL>
struct kidx_walk {
    enum { THREADS = 256 };
    // (round 4: a launch may be narrower than its work - `n_waves` wave-sized pieces, walked with the grid's stride: the count
    // walk's 160 k lanes of dependent random loads and atomics are what slows every other round's kernels, HISTORY.md 5.7)
    static __device__ void run(const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off,
                               const KxPos pos, const dp_scan_item* __restrict__ items, uint32_t lo, uint32_t hi,
                               uint32_t n_read_items, const uint32_t* __restrict__ head, const uint32_t* __restrict__ next,
                               uint32_t* __restrict__ counts, uint32_t* __restrict__ fillc, const uint64_t* __restrict__ segoff,
                               int32_t* __restrict__ segs, unsigned long long* __restrict__ n_hits, uint32_t lps,
                               unsigned long long* __restrict__ dbg, const KxRec R, uint32_t n_waves) {
        const uint32_t stride = gridDim.x * (blockDim.x >> 6);
        for (uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < n_waves; w += stride)
            one(w, seeds, n_seeds, off, pos, items, lo, hi, n_read_items, head, next, counts, fillc, segoff, segs, n_hits, lps, dbg, R);
    }
    static __device__ void one(const uint32_t w, const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off,
                                                 const KxPos pos, const dp_scan_item* __restrict__ items,
                                                 uint32_t lo, uint32_t hi, uint32_t n_read_items, const uint32_t* __restrict__ head,
                                                 const uint32_t* __restrict__ next, uint32_t* __restrict__ counts,
                                                 uint32_t* __restrict__ fillc, const uint64_t* __restrict__ segoff,
                                                 int32_t* __restrict__ segs, unsigned long long* __restrict__ n_hits, uint32_t lps,
                                                 unsigned long long* __restrict__ dbg, const KxRec R) {
    const int lane = dp_lane();
    // DP_KX_DEBUG: when a wave started, had its bucket bounds, its entries, its items, and was done (100 MHz ticks since dbg[15],
    // sums over waves in dbg[0..4], maxima in dbg[5..9], waves in dbg[10])
#define KX_TICK(i_)                                                                                     \
    if (KX_PROFILING && dbg) {                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                      \
        const unsigned long long n_ = wall_clock64();                                                   \
        if (lane == 0) dbg[8 * (size_t)w + (i_)] = n_;                                                  \
    }
    KX_TICK(0)
    // lps = lanes per seed.  64: KX_PARTS waves share one seed's bucket (dense seeds: buckets of hundreds to thousands);
    // 16: four seeds per wave (k = 13 at config 2: ~20 occurrences per seed - a whole wave per quarter bucket left 59 lanes idle
    // and made 40 k waves of a 10 k-seed round)
    // a "group" is what walks one stretch of a bucket together: a seed (16 lanes), or one of the KX_PARTS waves of a seed
    uint32_t s, i0, i1, step, n, grp, gfirst;
    uint64_t o;
    int gleader;
    if (lps == 64) {
        s = w / KX_PARTS;
        const uint32_t part = w % KX_PARTS;
        if (s >= n_seeds) return;
        o = off[seeds[s]];
        n = (uint32_t)(off[(uint64_t)seeds[s] + 1] - o);
        const uint32_t per = (n + KX_PARTS - 1) / KX_PARTS;
        gfirst = min(n, part * per);
        i0 = part * per + 4u * (uint32_t)lane;  // (a lane takes FOUR CONSECUTIVE entries per trip: one 16-byte load, kx_entry4)
        i1 = min(n, part * per + per);
        step = 64;
        grp = w;
        gleader = lane == 0;
    } else {
        s = w * 4 + ((uint32_t)lane >> 4);
        if (s >= n_seeds) return;
        o = off[seeds[s]];
        n = (uint32_t)(off[(uint64_t)seeds[s] + 1] - o);
        gfirst = 0;
        i0 = 4u * ((uint32_t)lane & 15u);
        i1 = n;
        step = 16;
        grp = s;
        gleader = (lane & 15) == 0;
    }
    KX_TICK(1)
    // Hit records (round 4): the count pass keeps every hit - {read, position, the rank the read's counter returned} - in a
    // stretch of R.rec of its group's own, so that the fill pass is one streaming pass over records (R.rec[at + j] -> the
    // survivor's slice at slot `rank`) instead of a second walk over seeds, buckets, items and counters.  Stretches are handed out
    // from 64 shards (the seed-occurrence counters of before, now read back: one returning atomic per group); a shard that is full
    // sets R.flags[0] and the round's fill pass walks the buckets as it used to.
    uint32_t rec_at = 0xffffffffu;
    if (!FILL) {
        const uint32_t gn = i1 > gfirst ? i1 - gfirst : 0u;
        unsigned long long old = 0;
        if (gleader && gn) old = atomicAdd(&n_hits[grp & 63u], (unsigned long long)gn);  // (64 slots: 10 k same-address atomics serialise)
        if (R.rec) {
            if (gleader) {
                if (old + gn <= R.shard_cap) {
                    rec_at = (grp & 63u) * R.shard_cap + (uint32_t)old;
                } else if (gn) {
                    R.flags[0] = 1u;
                }
                R.kb[2 * (size_t)grp] = rec_at;
                R.kb[2 * (size_t)grp + 1] = gn;
            }
            rec_at = (uint32_t)__shfl((int)rec_at, lps == 64 ? 0 : (lane & 48), 64);
        }
    }
    // Four entries per lane and trip: a hit is a chain of dependent loads (entry -> the read's item and list head -> counter), each
    // link a trip to HBM through a TLB that an 8 GB table defeats; the links of four entries travel together instead of one
    // after the other (a seed of config 2 has 20-40 entries: one trip of a 16-lane group instead of three)
    for (uint32_t ib = i0; ib < i1; ib += 4 * step) {
        uint64_t e[4];
        bool v[4];
        kx_entry4(pos, o + ib, e);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            v[u] = ib + (uint32_t)u < i1;
            e[u] = v[u] ? e[u] : 0ull;
        }
        KX_TICK(2)
        dp_scan_item item[4];
        uint32_t hd[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(e[u] >> 32);
            in[u] = v[u] && r >= lo && r < hi;
            if (!FILL && R.ign) {
                item[u] = dp_scan_item{};
                item[u].n_kmers = (in[u] && !R.ign[r]) ? 0xffffffffu : 0u;
                hd[u] = (v[u] && r - R.qlo < R.qspan) ? head[r] : 0u;
            } else {
                item[u] = in[u] ? items[r - lo] : dp_scan_item{};
                hd[u] = v[u] ? head[r] : 0u;
            }
        }
        KX_TICK(3)
        uint32_t cnt[4] = {0, 0, 0, 0};
        uint64_t so[4] = {0, 0, 0, 0};
        if (FILL) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t r = (uint32_t)(e[u] >> 32);
                in[u] = in[u] && (uint32_t)e[u] < item[u].n_kmers;
                if (in[u]) {
                    cnt[u] = counts[r - lo];
                    so[u] = segoff[r - lo];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(e[u] >> 32), p = (uint32_t)e[u];
            if (!FILL) {
                // (ignored reads carry n_kmers == 0; a top-level read with len % 4 == 0 four k-mers less)
                const bool valid = in[u] && p < item[u].n_kmers;
                if (R.rec) {
                    uint32_t rank = 0;
                    if (valid) rank = atomicAdd(&counts[r - lo], 1u);
                    if (v[u] && rec_at != 0xffffffffu)
                        R.rec[(size_t)rec_at + (ib + (uint32_t)u - gfirst)] = kx_rec(hd[u] != 0, valid, r, rank, p);
                } else if (valid) {
                    atomicAdd(&counts[r - lo], 1u);
                }
            } else if (in[u] && cnt[u] >= item[u].min_seeds) {
                const uint32_t slot = atomicAdd(&fillc[r - lo], 1u);
                const uint64_t at = so[u] + 2ull * slot;
                segs[at] = (int32_t)p;
                segs[at + 1] = (int32_t)s;
            }
            for (uint32_t x = hd[u]; x; x = next[x - 1]) {  // the round's extra items on this read (query windows)
                const uint32_t it = n_read_items + x - 1;
                const dp_scan_item xi = items[it];
                if (p - xi.start < xi.n_kmers && p >= xi.start) {
                    if (!FILL) {
                        atomicAdd(&counts[it], 1u);
                    } else if (counts[it] >= xi.min_seeds) {
                        const uint32_t slot = atomicAdd(&fillc[it], 1u);
                        const uint64_t at = segoff[it] + 2ull * slot;
                        segs[at] = (int32_t)(p - xi.start);
                        segs[at + 1] = (int32_t)s;
                    }
                }
            }
        }
    }
    KX_TICK(4)
#undef KX_TICK
}
};

// ---- round 5: the counting step without scattered atomics - hits binned by read range, counted in LDS ---------------------------
// The count walk of rounds 1 - 4 bumps a counter per hit: 450 k memory-side atomics per config-2 round (10 M in the dense regime),
// each a 64-byte request of its own - 6.8 x the step's algorithmic bytes (profiles/kindex_traffic.json, round 4).  Here the walk
// writes every hit that counts as an 8-byte record into the BIN of its read (bin = item >> bshift: a few hundred bins of 512 .. 16 k
// consecutive reads) and touches no counter:
//   kidx_walk_bin   a workgroup of 16 waves walks its seeds' buckets twice.  Sweep 0 counts its hits per bin in LDS; one returning
//                   atomic per (workgroup, bin touched) reserves a stretch of the bin's records - a few thousand atomics on a few
//                   hundred addresses per round instead of one per hit; sweep 1 (bucket entries from the L2 now) writes the records.
//                   Hits on the round's query windows (extra items) are counted as before (a few thousand per round) and kept in a
//                   list of their own.
//   kidx_bin_count  one workgroup per bin: its reads' counters live in LDS, every record is one LDS atomic; the counts go to
//                   memory as plain coalesced stores.
//   kidx_offsets    unchanged.
//   kidx_bin_fill   one workgroup per bin again: the survivors' records get their slot from an LDS counter (the rank no longer
//                   travels inside the record) and go to the read's segment slice; the extra items' list is filled by the blocks
//                   behind the bins.
// A bin that is full (sized from the previous round: 1.5 x its hits) counts the rest of its hits the old way and raises the flag
// that sends the round's fill pass to the bucket-walking form (dp_kindex_refill + dp_kindex_write) - counts are right either way.
#define KX_MAXBINS 2048
#define KXB_SEED_BITS 22
#define KXB_RIB_BITS 14
struct KxBins {
    unsigned long long* rec;  // [n_bins * cap]: [59:46] read - first read of the bin, [45:24] seed, [23:0] position in the read
    uint32_t* cursor;         // [n_bins] records handed out per bin (zeroed with the work area)
    uint32_t* flags;          // [0] a bin or the extra list was full
    uint32_t cap, n_bins, bshift;
    uint4* xrec;              // hits on extra items: {item, position in the item, seed, 0}
    uint32_t* xcursor;        // [1]
    uint32_t xcap;
    const uint8_t* ign;       // the walk's short cut (dp_kindex_fast), or null
    uint32_t qlo, qspan;
    uint32_t batch;           // trips of a walk workgroup that share one reservation per bin (1; the dense regime: 8)
};
static_assert(KXB_SEED_BITS + KXB_RIB_BITS + 24 <= 64, "a binned hit record is one 64-bit word");
static_assert((1u << KXB_RIB_BITS) * 4 <= 65536, "a bin's counters live in one workgroup's LDS");
__device__ __forceinline__ unsigned long long kxb_rec(uint32_t rib, uint32_t s, uint32_t p) {
    return ((unsigned long long)rib << (KXB_SEED_BITS + 24)) | ((unsigned long long)s << 24) | (unsigned long long)(p & 0xffffffu);
}

// WAVES waves per workgroup: more waves = fewer (workgroup, bin) reservations per hit, but a workgroup that is harder to place
// beside the other rounds' kernels and whose waves all wait at its barriers for the slowest one
template <int WAVES_>
struct kidx_walk_bin {
    enum { WAVES = WAVES_, THREADS = 64 * WAVES_ };
    // four consecutive bucket entries of one lane: counted (WRITE = false) or written as records (WRITE = true)
    template <bool WRITE>
    static __device__ __forceinline__ void four(const uint64_t e[4], const bool v[4], uint32_t s, const dp_scan_item* __restrict__ items,
                                                uint32_t lo, uint32_t hi, uint32_t n_read_items, const uint32_t* __restrict__ head,
                                                const uint32_t* __restrict__ next, uint32_t* __restrict__ counts, const KxBins& B,
                                                uint32_t* hist, const uint32_t* base, uint32_t* xs) {
        bool valid[4];
        uint32_t hd[4];
        const uint32_t rmask = (1u << B.bshift) - 1u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(e[u] >> 32), p = (uint32_t)e[u];
            const bool in = v[u] && r >= lo && r < hi;
            if (B.ign) {
                // (the ignore byte is NOT looked at here: another slot's commit may flag the read between the two passes, and a record
                // slot that was counted and not written would hold a stale word.  kidx_bin_count applies it, once per read)
                valid[u] = in;
                hd[u] = (v[u] && r - B.qlo < B.qspan) ? head[r] : 0u;
            } else {
                // (ignored reads carry n_kmers == 0; a top-level read with len % 4 == 0 four k-mers less)
                valid[u] = in && p < items[in ? r - lo : 0u].n_kmers;
                hd[u] = v[u] ? head[r] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(e[u] >> 32), p = (uint32_t)e[u];
            if (valid[u]) {
                const uint32_t bin = (r - lo) >> B.bshift;
                if (!WRITE) {
                    atomicAdd(&hist[bin], 1u);
                } else {
                    const uint32_t b = base[bin];
                    if (b != 0xffffffffu) {
                        const uint32_t rank = atomicAdd(&hist[bin], 1u);
                        B.rec[(size_t)bin * B.cap + b + rank] = kxb_rec((r - lo) & rmask, s, p);
                    } else {
                        atomicAdd(&counts[r - lo], 1u);  // (the bin is full: counted the old way, kidx_bin_count adds its own on top)
                    }
                }
            }
            // the round's extra items on this read (query windows): counted in the first pass, kept as a list in the second - a stretch of
            // the list per workgroup (xs[0] counts, then ranks; xs[1] = the stretch's start: ONE returning atomic per workgroup - a
            // returning atomic per hit, 5 k per round on one address, was 40 us of this kernel)
            for (uint32_t x = hd[u]; x; x = next[x - 1]) {
                const uint32_t it = n_read_items + x - 1;
                const dp_scan_item xi = items[it];
                if (p - xi.start < xi.n_kmers && p >= xi.start) {
                    if (!WRITE) {
                        atomicAdd(&counts[it], 1u);
                        atomicAdd(&xs[0], 1u);
                    } else if (xs[1] != 0xffffffffu) {
                        B.xrec[xs[1] + atomicAdd(&xs[0], 1u)] = make_uint4(it, p - xi.start, s, 0u);
                    }
                }
            }
        }
    }
    // what lies behind a lane's first four entries.  Dense seeds (a wave per quarter bucket): the wave's stride, as ever.  Four seeds per
    // wave: a seed's 16 lanes cover 64 entries per trip - and the kernel is as long as its longest bucket (a repeat's k-mer with a few
    // thousand occurrences was 40+ dependent trips of ONE quarter wave, twice over with two passes): everything beyond a bucket's first
    // 64 entries is walked by the WHOLE wave, 256 entries per trip, one seed of the wave after the other
    template <bool WRITE>
    static __device__ __forceinline__ void tail(uint32_t lps, int lane, uint32_t w, uint32_t s, uint64_t o, uint32_t i0, uint32_t i1, uint32_t step,
                                                const KxPos pos, const dp_scan_item* __restrict__ items, uint32_t lo, uint32_t hi,
                                                uint32_t n_read_items, const uint32_t* __restrict__ head, const uint32_t* __restrict__ next,
                                                uint32_t* __restrict__ counts, const KxBins& B, uint32_t* hist, const uint32_t* base, uint32_t* xs) {
        if (lps == 64) {
            for (uint32_t ib = i0 + 4 * step; ib < i1; ib += 4 * step) {
                uint64_t e[4];
                bool v[4];
                kx_entry4(pos, o + ib, e);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = ib + (uint32_t)u < i1;
                    e[u] = v[u] ? e[u] : 0ull;
                }
                four<WRITE>(e, v, s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            }
            return;
        }
        if (!__ballot(i1 > 64u)) return;  // (wave-uniform: no bucket of this wave goes beyond its first trip)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t ng = (uint32_t)__shfl((int)i1, 16 * g, 64);
            if (ng <= 64u) continue;
            const uint32_t lo32 = (uint32_t)__shfl((int)(uint32_t)o, 16 * g, 64), hi32 = (uint32_t)__shfl((int)(uint32_t)(o >> 32), 16 * g, 64);
            const uint64_t og = ((uint64_t)hi32 << 32) | lo32;
            const uint32_t sg = w * 4 + (uint32_t)g;
            for (uint32_t ib = 64u + 4u * (uint32_t)lane; ib < ng; ib += 256u) {
                uint64_t e[4];
                bool v[4];
                kx_entry4(pos, og + ib, e);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = ib + (uint32_t)u < ng;
                    e[u] = v[u] ? e[u] : 0ull;
                }
                four<WRITE>(e, v, sg, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            }
        }
    }
    // One trip per workgroup and reservation - the sparse regime's form (workgroups of up to 8 waves), as it was before the dense regime's
    // trips, table and look-ahead were built into run() below: those cost the k = 13 walk 20 us a launch (36 -> 56 us under five slots,
    // profiles/r06/k13_against_round5.txt) and buy it nothing - its reservations are one trip long.
    static __device__ void run_one_trip(const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off, const KxPos pos,
                               const dp_scan_item* __restrict__ items, uint32_t lo, uint32_t hi, uint32_t n_read_items,
                               const uint32_t* __restrict__ head, const uint32_t* __restrict__ next, uint32_t* __restrict__ counts,
                               unsigned long long* __restrict__ n_hits, uint32_t lps, const KxBins B, uint32_t n_waves) {
        __shared__ uint32_t hist[KX_MAXBINS], base[KX_MAXBINS];
        __shared__ uint32_t xs[2];
        __shared__ unsigned long long sh_hits;
        const int lane = dp_lane();
        const uint32_t stride = gridDim.x * WAVES;
        // (every wave of a workgroup makes the same number of trips: the barriers below are the workgroup's)
        for (uint32_t wb = blockIdx.x * WAVES; wb < n_waves; wb += stride) {
            for (uint32_t t = threadIdx.x; t < B.n_bins; t += THREADS) hist[t] = 0u;
            if (threadIdx.x == 0) {
                sh_hits = 0ull;
                xs[0] = xs[1] = 0u;
            }
            __syncthreads();
            const uint32_t w = wb + (threadIdx.x >> 6);
            // the groups of kidx_walk: KX_PARTS waves per seed (dense seeds), or four seeds per wave
            uint32_t s = 0, i0 = 0, i1 = 0, step = 16, gfirst = 0;
            uint64_t o = 0;
            bool gleader = false;
            if (w < n_waves) {
                if (lps == 64) {
                    s = w / KX_PARTS;
                    if (s < n_seeds) {
                        const uint32_t part = w % KX_PARTS;
                        o = off[seeds[s]];
                        const uint32_t n = (uint32_t)(off[(uint64_t)seeds[s] + 1] - o);
                        const uint32_t per = (n + KX_PARTS - 1) / KX_PARTS;
                        gfirst = min(n, part * per);
                        i0 = part * per + 4u * (uint32_t)lane;
                        i1 = min(n, part * per + per);
                        step = 64;
                        gleader = lane == 0;
                    }
                } else {
                    s = w * 4 + ((uint32_t)lane >> 4);
                    if (s < n_seeds) {
                        o = off[seeds[s]];
                        i0 = 4u * ((uint32_t)lane & 15u);
                        i1 = (uint32_t)(off[(uint64_t)seeds[s] + 1] - o);
                        gleader = (lane & 15) == 0;
                    }
                }
            }
            if (gleader && i1 > gfirst) atomicAdd(&sh_hits, (unsigned long long)(i1 - gfirst));
            // the lane's first four entries stay in registers across the barriers (a seed of config 2 has 20 - 60 entries: for most groups
            // the first trip is the only one, and the second pass reads nothing); later trips are read again (from the L2)
            uint64_t e0[4] = {0, 0, 0, 0};
            bool v0[4] = {false, false, false, false};
            if (i0 < i1) {
                kx_entry4(pos, o + i0, e0);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v0[u] = i0 + (uint32_t)u < i1;
                    e0[u] = v0[u] ? e0[u] : 0ull;
                }
                four<false>(e0, v0, s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            }
            tail<false>(lps, lane, w, s, o, i0, i1, step, pos, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < B.n_bins; t += THREADS) {
                const uint32_t c = hist[t];
                uint32_t b = 0u;
                if (c) {
                    b = atomicAdd(&B.cursor[t], c);
                    if (b + c > B.cap) {  // (its share of the bin does not fit: the whole share is counted the old way)
                        // what it reserved below the bin's end stays unwritten: marked, so that kidx_bin_count skips it
                        for (uint32_t j = b; j < B.cap; j++) B.rec[(size_t)t * B.cap + j] = ~0ull;
                        b = 0xffffffffu;
                        B.flags[0] = 1u;
                    }
                }
                base[t] = b;
                hist[t] = 0u;
            }
            if (threadIdx.x == 0 && sh_hits) atomicAdd(&n_hits[blockIdx.x & 63u], sh_hits);  // seed occurrences of the round (totals[2])
            if (threadIdx.x == THREADS - 1) {  // the workgroup's stretch of the extra items' list
                const uint32_t c = xs[0];
                uint32_t xb = 0u;
                if (c) {
                    xb = atomicAdd(B.xcursor, c);
                    if (xb + c > B.xcap) {
                        xb = 0xffffffffu;
                        B.flags[0] = 1u;
                    }
                }
                xs[1] = xb;
                xs[0] = 0u;
            }
            __syncthreads();
            if (i0 < i1) four<true>(e0, v0, s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            tail<true>(lps, lane, w, s, o, i0, i1, step, pos, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
            __syncthreads();
        }
    }
    static __device__ void run(const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off, const KxPos pos,
                               const dp_scan_item* __restrict__ items, uint32_t lo, uint32_t hi, uint32_t n_read_items,
                               const uint32_t* __restrict__ head, const uint32_t* __restrict__ next, uint32_t* __restrict__ counts,
                               unsigned long long* __restrict__ n_hits, uint32_t lps, const KxBins B, uint32_t n_waves) {
        if constexpr (WAVES != 16) {
            run_one_trip(seeds, n_seeds, off, pos, items, lo, hi, n_read_items, head, next, counts, n_hits, lps, B, n_waves);
            return;
        }
        __shared__ uint32_t hist[KX_MAXBINS], base[KX_MAXBINS];
        __shared__ uint32_t xs[2];
        __shared__ unsigned long long sh_hits;
        const int lane = dp_lane();
        const uint32_t stride = gridDim.x * WAVES;
        // B.batch trips of the workgroup share one reservation per bin (round 6, the dense regime: a seed's bucket holds a thousand
        // entries, a workgroup trip 2 - 4 seeds - a few records per bin and trip, each stretch a returning atomic and a short run of
        // stores; eight trips a reservation are eight times fewer atomics and runs eight times as long).  Trips behind the first read
        // their entries again in the second pass (from the L2).
        const uint32_t T = min(32u, max(1u, B.batch));  // (the table below holds 32 trips)
        // the groups of kidx_walk: KX_PARTS waves per seed (dense seeds), or four seeds per wave
        struct Grp {
            uint32_t s, i0, i1, step, gfirst;
            uint64_t o;
            bool gleader;
        };
        // Round 6: where a seed's bucket starts and how long it is comes from an LDS table the whole workgroup fills at once, for
        // every trip of the reservation (lps == 64: a wave per quarter bucket) - two dependent loads less in front of every trip's
        // entries - and the entries of the next trip are asked for before the current trip's are worked on (below)
        __shared__ unsigned long long t_o[WAVES * 32 / KX_PARTS + 2];
        __shared__ uint32_t t_n[WAVES * 32 / KX_PARTS + 2];
        uint32_t t_first = 0u;  // seed of the table's first row
        auto group_of = [&](uint32_t w) {
            Grp g = {0u, 0u, 0u, 16u, 0u, 0ull, false};
            if (w < n_waves) {
                if (lps == 64) {
                    g.s = w / KX_PARTS;
                    if (g.s < n_seeds) {
                        const uint32_t part = w % KX_PARTS;
                        g.o = t_o[g.s - t_first];
                        const uint32_t n = t_n[g.s - t_first];
                        const uint32_t per = (n + KX_PARTS - 1) / KX_PARTS;
                        g.gfirst = min(n, part * per);
                        g.i0 = part * per + 4u * (uint32_t)lane;
                        g.i1 = min(n, part * per + per);
                        g.step = 64;
                        g.gleader = lane == 0;
                    }
                } else {
                    g.s = w * 4 + ((uint32_t)lane >> 4);
                    if (g.s < n_seeds) {
                        g.o = off[seeds[g.s]];
                        g.i0 = 4u * ((uint32_t)lane & 15u);
                        g.i1 = (uint32_t)(off[(uint64_t)seeds[g.s] + 1] - g.o);
                        g.gleader = (lane & 15) == 0;
                    }
                }
            }
            return g;
        };
        // (every wave of a workgroup makes the same number of trips: the barriers below are the workgroup's)
        for (uint32_t wb = blockIdx.x * WAVES * T; wb < n_waves; wb += stride * T) {
            for (uint32_t t = threadIdx.x; t < B.n_bins; t += THREADS) hist[t] = 0u;
            if (threadIdx.x == 0) {
                sh_hits = 0ull;
                xs[0] = xs[1] = 0u;
            }
            if (lps == 64) {
                t_first = wb / KX_PARTS;
                const uint32_t rows = (WAVES * min(T, 32u) + KX_PARTS - 1) / KX_PARTS + 1u;
                for (uint32_t t = threadIdx.x; t < rows; t += THREADS) {
                    const uint32_t sd = t_first + t;
                    unsigned long long o = 0ull;
                    uint32_t n = 0u;
                    if (sd < n_seeds) {
                        const uint32_t km = seeds[sd];
                        o = off[km];
                        n = (uint32_t)(off[(uint64_t)km + 1] - o);
                    }
                    t_o[t] = o;
                    t_n[t] = n;
                }
            }
            __syncthreads();
            // the lane's first four entries of the FIRST trip stay in registers across the barriers (a seed of config 2 has 20 - 60
            // entries: for most groups the first trip is the only one, and the second pass reads nothing)
            uint64_t e0[4] = {0, 0, 0, 0};
            bool v0[4] = {false, false, false, false};
            {
                Grp g = group_of(wb + (threadIdx.x >> 6));
                uint64_t e[4] = {0, 0, 0, 0};
                if (g.i0 < g.i1) kx_entry4(pos, g.o + g.i0, e);
                for (uint32_t tr = 0; tr < T; tr++) {
                    const uint32_t w = wb + tr * WAVES + (threadIdx.x >> 6);
                    // the next trip's entries are on their way while this trip's are counted
                    Grp gn = g;
                    uint64_t en[4] = {0, 0, 0, 0};
                    if (tr + 1 < T) {
                        gn = group_of(w + WAVES);
                        if (gn.i0 < gn.i1) kx_entry4(pos, gn.o + gn.i0, en);
                    }
                    if (g.gleader && g.i1 > g.gfirst) atomicAdd(&sh_hits, (unsigned long long)(g.i1 - g.gfirst));
                    bool v[4] = {false, false, false, false};
                    if (g.i0 < g.i1) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            v[u] = g.i0 + (uint32_t)u < g.i1;
                            e[u] = v[u] ? e[u] : 0ull;
                        }
                        four<false>(e, v, g.s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++) e[u] = 0ull;
                    }
                    if (tr == 0) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            e0[u] = e[u];
                            v0[u] = v[u];
                        }
                    }
                    tail<false>(lps, lane, w, g.s, g.o, g.i0, g.i1, g.step, pos, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
                    g = gn;
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = en[u];
                }
            }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < B.n_bins; t += THREADS) {
                const uint32_t c = hist[t];
                uint32_t b = 0u;
                if (c) {
                    b = atomicAdd(&B.cursor[t], c);
                    if (b + c > B.cap) {  // (its share of the bin does not fit: the whole share is counted the old way)
                        // what it reserved below the bin's end stays unwritten: marked, so that kidx_bin_count skips it
                        for (uint32_t j = b; j < B.cap; j++) B.rec[(size_t)t * B.cap + j] = ~0ull;
                        b = 0xffffffffu;
                        B.flags[0] = 1u;
                    }
                }
                base[t] = b;
                hist[t] = 0u;
            }
            if (threadIdx.x == 0 && sh_hits) atomicAdd(&n_hits[blockIdx.x & 63u], sh_hits);  // seed occurrences of the round (totals[2])
            if (threadIdx.x == THREADS - 1) {  // the workgroup's stretch of the extra items' list
                const uint32_t c = xs[0];
                uint32_t xb = 0u;
                if (c) {
                    xb = atomicAdd(B.xcursor, c);
                    if (xb + c > B.xcap) {
                        xb = 0xffffffffu;
                        B.flags[0] = 1u;
                    }
                }
                xs[1] = xb;
                xs[0] = 0u;
            }
            __syncthreads();
            {
                Grp g = group_of(wb + (threadIdx.x >> 6));
                uint64_t e[4] = {0, 0, 0, 0};
                for (uint32_t tr = 0; tr < T; tr++) {
                    const uint32_t w = wb + tr * WAVES + (threadIdx.x >> 6);
                    Grp gn = g;
                    uint64_t en[4] = {0, 0, 0, 0};
                    if (tr + 1 < T) {  // (the first trip's entries waited in registers; every later trip's are read again - from the L2 - a trip ahead)
                        gn = group_of(w + WAVES);
                        if (gn.i0 < gn.i1) kx_entry4(pos, gn.o + gn.i0, en);
                    }
                    if (g.i0 < g.i1) {
                        if (tr == 0) {
                            four<true>(e0, v0, g.s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
                        } else {
                            bool v[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                v[u] = g.i0 + (uint32_t)u < g.i1;
                                e[u] = v[u] ? e[u] : 0ull;
                            }
                            four<true>(e, v, g.s, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
                        }
                    }
                    tail<true>(lps, lane, w, g.s, g.o, g.i0, g.i1, g.step, pos, items, lo, hi, n_read_items, head, next, counts, B, hist, base, xs);
                    g = gn;
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = en[u];
                }
            }
            __syncthreads();
        }
    }
};

// RB = counters a workgroup's LDS holds (reads per bin <= RB)
template <int RB>
struct kidx_bin_count {
    enum { THREADS = 512 };
    static __device__ void run(const KxBins B, uint32_t* __restrict__ counts, uint32_t n_read_items, uint32_t lo) {
        __shared__ uint32_t cnt[RB];
        const uint32_t bin = blockIdx.x;
        if (bin >= B.n_bins) return;
        const uint32_t rb = 1u << B.bshift, first = bin << B.bshift;
        const uint32_t n = min(B.cursor[bin], B.cap);
        for (uint32_t i = threadIdx.x; i < rb; i += THREADS) cnt[i] = 0u;
        __syncthreads();
        const unsigned long long* rec = B.rec + (size_t)bin * B.cap;
        for (uint32_t jb = 4u * threadIdx.x; jb < n; jb += 4u * THREADS) {  // (four consecutive records per thread and trip)
            unsigned long long e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = jb + (uint32_t)u < n ? rec[jb + u] : ~0ull;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (e[u] != ~0ull) atomicAdd(&cnt[(uint32_t)(e[u] >> (KXB_SEED_BITS + 24))], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < rb; i += THREADS) {
            const uint32_t it = first + i;
            if (it < n_read_items) {
                const uint32_t c = cnt[i];
                // (zeroed by kidx_prepare; a full bin's overflow was counted there directly.  A read ignored by now counts nothing - the
                // walk's short cut tested this byte per hit; the fill pass goes by the count written here)
                if (B.ign && B.ign[lo + it]) counts[it] = 0u;
                else if (c) counts[it] += c;
            }
        }
    }
};

template <int RB>
struct kidx_bin_fill {
    enum { THREADS = 512, XBLOCKS = 8 };
    static __device__ void run(const KxBins B, const dp_scan_item* __restrict__ items, uint32_t n_read_items,
                               const uint32_t* __restrict__ counts, uint32_t* __restrict__ fillc, const uint64_t* __restrict__ segoff,
                               int32_t* __restrict__ segs, const uint64_t* __restrict__ totals, uint64_t seg_cap) {
        __shared__ uint32_t cnt[RB];
        if (B.flags[0] || totals[0] > seg_cap) return;  // (the host repeats the fill with the bucket walk / a larger buffer)
        if (blockIdx.x >= B.n_bins) {  // the extra items' hits: a slot from the item's fill cursor, as ever
            const uint32_t xb = blockIdx.x - B.n_bins;
            if (xb >= XBLOCKS) return;
            const uint32_t nx = min(*B.xcursor, B.xcap);
            for (uint32_t j = xb * THREADS + threadIdx.x; j < nx; j += XBLOCKS * THREADS) {
                const uint4 x = B.xrec[j];
                if (counts[x.x] >= items[x.x].min_seeds) {
                    const uint32_t slot = atomicAdd(&fillc[x.x], 1u);
                    const uint64_t to = segoff[x.x] + 2ull * slot;
                    segs[to] = (int32_t)x.y;
                    segs[to + 1] = (int32_t)x.z;
                }
            }
            return;
        }
        const uint32_t bin = blockIdx.x;
        const uint32_t rb = 1u << B.bshift, first = bin << B.bshift;
        const uint32_t n = min(B.cursor[bin], B.cap);
        // slot counters of the bin's survivors (top bit: the read does not survive - its own min_seeds, what kidx_offsets tested)
        for (uint32_t i = threadIdx.x; i < rb; i += THREADS) {
            const uint32_t it = first + i;
            bool surv = false;
            if (it < n_read_items) {
                const uint32_t c = counts[it];
                surv = c > 0u && c >= items[it].min_seeds;
            }
            cnt[i] = surv ? 0u : 0x80000000u;
        }
        __syncthreads();
        const unsigned long long* rec = B.rec + (size_t)bin * B.cap;
        for (uint32_t jb = 4u * threadIdx.x; jb < n; jb += 4u * THREADS) {
            unsigned long long e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = jb + (uint32_t)u < n ? rec[jb + u] : ~0ull;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (e[u] == ~0ull) continue;
                const uint32_t rib = (uint32_t)(e[u] >> (KXB_SEED_BITS + 24));
                if (cnt[rib] & 0x80000000u) continue;
                const uint32_t rank = atomicAdd(&cnt[rib], 1u);
                const uint64_t to = segoff[first + rib] + 2ull * rank;
                segs[to] = (int32_t)((uint32_t)e[u] & 0xffffffu);
                segs[to + 1] = (int32_t)((uint32_t)(e[u] >> 24) & ((1u << KXB_SEED_BITS) - 1u));
            }
        }
    }
};

// status word of a tile: flag << 62 | survivors << 38 | segment ints (flag 1 = the tile's own sums, 2 = inclusive prefix)
#define KX_TILE 1024
#ifndef KX_IPT
#define KX_IPT 4
#endif
struct kidx_offsets {
    enum { THREADS = KX_TILE };
    // B.rec != null (round 5, bins of at most a tile's 4 096 reads): the tile counts the records of its own bins in LDS first - what
    // kidx_bin_count does in a launch of its own (every launch of a round costs a five-slot round 1.5 - 1.9 us whatever it does:
    // profiles/r05/ab_extra_launches_five_slots.txt) - and stores the counts it then scans
    static __device__ void run(const dp_scan_item* __restrict__ items, uint32_t* __restrict__ counts,
                                                       uint32_t n, unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket,
                                                       uint64_t* __restrict__ segoff, uint32_t* __restrict__ s_item,
                                                       uint32_t* __restrict__ s_count, uint64_t* __restrict__ s_off,
                                                       uint4* __restrict__ s_pack, uint64_t* __restrict__ totals,
                                                       uint32_t n_read_items, uint32_t* __restrict__ max_count,
                                                       const unsigned long long* __restrict__ n_hits,
                                                       unsigned long long* __restrict__ host_totals, const KxBins B, uint32_t lo) {
    __shared__ uint32_t shA[16], shB[16];
    __shared__ uint32_t tile_s;
    __shared__ unsigned long long excl_s;
    __shared__ uint32_t tcnt[KX_TILE * KX_IPT];
    // (a launch shared with other rounds has the largest round's grid: blocks beyond this round's own tiles take no ticket)
    if (blockIdx.x >= (n + KX_TILE * KX_IPT - 1) / (KX_TILE * KX_IPT)) return;
    if (threadIdx.x == 0) tile_s = atomicAdd(ticket, 1u);  // tiles start in ticket order: a predecessor is always running or done
    __syncthreads();
    const uint32_t tile = tile_s;
    // KX_IPT consecutive items per thread: a quarter of the tiles, a quarter of the look-back chain
    const uint32_t i0 = (tile * KX_TILE + threadIdx.x) * KX_IPT;
    uint32_t cc[KX_IPT], seg = 0, fl = 0, segv[KX_IPT], flv[KX_IPT], cmax = 0;
    // (every item's two loads are asked for before the first is looked at: inside an `if (i < n)` each item was a trip to memory of its
    // own - a microsecond per item and thread, which is what made 8 and 16 items per thread slower than 4)
    uint32_t msv[KX_IPT];
    if (B.rec) {
        const uint32_t t0 = tile * (KX_TILE * KX_IPT);  // first item of the tile = first read of its first bin
        for (uint32_t i = threadIdx.x; i < KX_TILE * KX_IPT; i += THREADS) tcnt[i] = 0u;
        __syncthreads();
        const uint32_t b0 = t0 >> B.bshift, nb = (KX_TILE * KX_IPT) >> B.bshift;
        for (uint32_t b = b0; b < b0 + nb && b < B.n_bins; b++) {
            const uint32_t nrec = min(B.cursor[b], B.cap), base = (b - b0) << B.bshift;
            const unsigned long long* rec = B.rec + (size_t)b * B.cap;
            for (uint32_t jb = 4u * threadIdx.x; jb < nrec; jb += 4u * THREADS) {  // (four consecutive records per thread and trip)
                unsigned long long e[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = jb + (uint32_t)u < nrec ? rec[jb + u] : ~0ull;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (e[u] != ~0ull) atomicAdd(&tcnt[base + (uint32_t)(e[u] >> (KXB_SEED_BITS + 24))], 1u);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < KX_IPT; u++) {
        const uint32_t i = min(i0 + (uint32_t)u, n - 1u);
        cc[u] = counts[i];
        msv[u] = items[i].min_seeds;
    }
    if (B.rec) {
        // (a read item's count = its bin's records + what a full bin counted the old way; a read ignored by now counts nothing)
#pragma unroll
        for (int u = 0; u < KX_IPT; u++) {
            const uint32_t i = i0 + (uint32_t)u;
            if (i < n_read_items) {
                cc[u] = (B.ign && B.ign[lo + i]) ? 0u : cc[u] + tcnt[threadIdx.x * KX_IPT + u];
                counts[i] = cc[u];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < KX_IPT; u++) {
        const bool in = i0 + (uint32_t)u < n;
        cc[u] = in ? cc[u] : 0u;
        flv[u] = in && cc[u] >= msv[u] ? 1u : 0u;
        segv[u] = flv[u] ? 2u * cc[u] + 1u : 0u;
        if (flv[u]) cmax = max(cmax, cc[u]);
        seg += segv[u];
        fl += flv[u];
    }
    // block scans of both quantities (segment lengths of a tile stay far below 2^32: checked by the caller's caps)
    const int lane = dp_lane(), wave = threadIdx.x >> 6;
    uint32_t xs = (uint32_t)wave_incl_sum_dpp((int)seg), xf = (uint32_t)wave_incl_sum_dpp((int)fl);
    if (lane == 63) {
        shA[wave] = xs;
        shB[wave] = xf;
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t a = lane < 16 ? shA[lane] : 0u, b = lane < 16 ? shB[lane] : 0u;
        a = (uint32_t)wave_incl_sum_dpp((int)a);
        b = (uint32_t)wave_incl_sum_dpp((int)b);
        if (lane < 16) {
            shA[lane] = a;
            shB[lane] = b;
        }
    }
    __syncthreads();
    if (wave > 0) {
        xs += shA[wave - 1];
        xf += shB[wave - 1];
    }
    const uint32_t tot_s = shA[15], tot_f = shB[15];
    {
        // (one atomic per wave that has a survivor: every surviving thread's own was thousands of atomics on ONE address, which the
        // kernel's end waited for)
        const uint32_t wmax = (uint32_t)wave_max_dpp((int)cmax);
        if (lane == 0 && wmax > 0) atomicMax(max_count, wmax);
    }
    if (wave == 0) {  // (the whole wave looks back: dp_wave_lookback)
        const unsigned long long own = ((unsigned long long)tot_f << 38) | (unsigned long long)tot_s;
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&status[0], (2ull << 62) | own, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&status[tile], (1ull << 62) | own, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            excl = dp_wave_lookback(status, tile, lane);
            if (lane == 0) __hip_atomic_store(&status[tile], (2ull << 62) | (excl + own), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            excl_s = excl;
            if (((uint64_t)tile + 1) * KX_TILE * KX_IPT >= n) {  // last tile: totals
                const unsigned long long all = excl + own;
                totals[0] = all & ((1ull << 38) - 1);
                totals[1] = all >> 38;
            }
        }
    }
    __syncthreads();
    const unsigned long long excl = excl_s;
    const uint64_t seg_base = excl & ((1ull << 38) - 1), surv_base = excl >> 38;
    uint64_t so = seg_base + (uint64_t)(xs - seg), slot = surv_base + (uint64_t)(xf - fl);
#pragma unroll
    for (int u = 0; u < KX_IPT; u++) {
        const uint32_t i = i0 + u;
        if (i < n) {
            segoff[i] = so;
            if (i == n_read_items) totals[5] = so;  // where the extra items' (query windows') segments start
            if (flv[u]) {
                s_item[slot] = i;
                s_count[slot] = cc[u];
                s_off[slot] = so;
                s_pack[slot] = make_uint4(i, cc[u], (uint32_t)so, (uint32_t)(so >> 32));  // the same triple, as the host fetches it
                slot++;
            }
            so += segv[u];
            if (i == n - 1) segoff[n] = so;
        }
    }
    if (tile == 0 && threadIdx.x < 64) {  // seed occurrences of the round (kidx_walk<false> has finished: stream order); here, at the
                                          // end, because every other tile waits for tile 0's prefix
        unsigned long long h = n_hits[threadIdx.x];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
        if (threadIdx.x == 0) totals[2] = h;
    }
    // the round's totals go to the host's pinned block with the tile that finishes last (no launch of their own): every tile
    // publishes its stores and counts itself in; the one that counts the last sees them all
    if (host_totals) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t n_tiles = (n + KX_TILE * KX_IPT - 1) / (KX_TILE * KX_IPT);
            __threadfence();
            const uint32_t seen = __hip_atomic_fetch_add(&ticket[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (seen == n_tiles) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
                for (int w = 0; w < 8; w++)
                    host_totals[w] = __hip_atomic_load((const unsigned long long*)&totals[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
};

// One survivor's slice: c unordered (position, seed) pairs at segs[out ..] -> sorted by position -> [gap, seed, ..., gap] in place (and in the
// host mirror for an extra item).  Executed by ONE wave with `keys` (CAP words of LDS) to itself.  WG1: the wave is the whole workgroup
// (kidx_sortwrite: its barriers are workgroup barriers, its loads plain); otherwise it is one wave of a larger workgroup whose other
// waves have just stored the pairs (kidx_bin_fill_sort: wave-level ordering, loads through the L2 - a neighbour workgroup on the same CU
// may have pulled a shared line into the vector L1 before the pairs were written).
template <int CAP, bool WG1>
__device__ __forceinline__ void kx_sort_one(unsigned long long* keys, const int lane, const uint32_t it, const uint32_t c, const uint64_t out,
                                            const int nk, int32_t* __restrict__ segs, const int k, uint32_t* __restrict__ overflow,
                                            const uint32_t n_read_items, int32_t* __restrict__ host_segs) {
#define KX_SYNC()                                 \
    {                                             \
        if (WG1) __syncthreads();                 \
        else __builtin_amdgcn_wave_barrier();     \
    }
#define KX_LD(p_) (WG1 ? (uint32_t)*(p_) : (uint32_t)__hip_atomic_load((p_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        KX_SYNC();
        if (c <= 64) {
            unsigned long long key = ~0ull;
            if ((uint32_t)lane < c) key = ((unsigned long long)KX_LD(&segs[out + 2 * (uint64_t)lane]) << 32) | KX_LD(&segs[out + 2 * (uint64_t)lane + 1]);
            int rank = 0;
            for (uint32_t l = 0; l < c; l++) {
                const unsigned long long o = __shfl(key, (int)l, 64);
                rank += o < key ? 1 : 0;
            }
            // positions are distinct (one k-mer per position), so the ranks are a permutation
            KX_SYNC();
            if ((uint32_t)lane < c) keys[rank] = key;
            KX_SYNC();
        } else if (c <= (uint32_t)CAP / 2) {
            // rank sort through LDS: every key counts the keys below it (c^2 / 64 broadcast reads per lane; c is a read's hit
            // count, a few hundred in the dense-seed regime) - no barriers, no index arithmetic
            unsigned long long* raw = keys + CAP / 2;  // unsorted copy in the upper half
            for (uint32_t j = lane; j < c; j += 64)
                raw[j] = ((unsigned long long)KX_LD(&segs[out + 2 * (uint64_t)j]) << 32) | KX_LD(&segs[out + 2 * (uint64_t)j + 1]);
            KX_SYNC();
            for (uint32_t j = lane; j < c; j += 64) {
                const unsigned long long key = raw[j];
                uint32_t rank = 0, l = 0;
                for (; l + 4 <= c; l += 4)
                    rank += (raw[l] < key ? 1u : 0u) + (raw[l + 1] < key ? 1u : 0u) + (raw[l + 2] < key ? 1u : 0u) + (raw[l + 3] < key ? 1u : 0u);
                for (; l < c; l++) rank += raw[l] < key ? 1u : 0u;
                keys[rank] = key;
            }
            KX_SYNC();
        } else if (c <= (uint32_t)CAP) {
            // more keys than half the block's LDS: bitonic network in place, padded with ~0 up to a power of two (<= CAP)
            uint32_t m = 128;
            while (m < c) m <<= 1;
            for (uint32_t j = lane; j < m; j += 64)
                keys[j] = j < c ? ((unsigned long long)KX_LD(&segs[out + 2 * (uint64_t)j]) << 32) | KX_LD(&segs[out + 2 * (uint64_t)j + 1]) : ~0ull;
            KX_SYNC();
            for (uint32_t size = 2; size <= m; size <<= 1) {
                for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                    for (uint32_t t = lane; t < m / 2; t += 64) {
                        const uint32_t i = ((t / stride) * 2 * stride) + (t % stride);
                        const bool up = (i & size) == 0;
                        const unsigned long long a = keys[i], b = keys[i + stride];
                        if ((a > b) == up) {
                            keys[i] = b;
                            keys[i + stride] = a;
                        }
                    }
                    KX_SYNC();
                }
            }
        } else {
            if (lane == 0) atomicExch(overflow, 1u);
            return;
        }
        const bool mirror = host_segs != nullptr && it >= n_read_items;
        for (uint32_t j = lane; j < c; j += 64) {
            const int p = (int)(keys[j] >> 32);
            const int prev = j ? (int)(keys[j - 1] >> 32) : -k;  // "-k": the first gap is the hit's own index
            segs[out + 2 * (uint64_t)j] = p - (prev + k);
            segs[out + 2 * (uint64_t)j + 1] = (int32_t)(uint32_t)keys[j];
            if (mirror) {
                host_segs[out + 2 * (uint64_t)j] = p - (prev + k);
                host_segs[out + 2 * (uint64_t)j + 1] = (int32_t)(uint32_t)keys[j];
            }
        }
        if (lane == 0) {
            const int last = c ? (int)(keys[c - 1] >> 32) : -k;
            segs[out + 2 * (uint64_t)c] = nk - last - 1;  // final gap (sequence/asm_amd64.s:387-392)
            if (mirror) host_segs[out + 2 * (uint64_t)c] = nk - last - 1;
        }
#undef KX_SYNC
#undef KX_LD
}

// one wave per survivor: its slice holds c unordered (position, seed) pairs -> sorted by position -> [gap, seed, ..., gap]
#define KX_SORT_LDS 4096
// (round 4's hit records - DP_KX_BINS=0 - keep a hit's rank among its read's hits in 14 bits, read and position in 24 each: the one-go
// step is only taken while no survivor has more hits than the LDS sort holds and reads / positions stay below 2^24, dp_scan.hip)
static_assert(KX_SORT_LDS <= (1 << 14), "kx_rec keeps the rank of a hit in 14 bits: the LDS sort's capacity bounds it");
// CAP = keys the block's LDS holds (the launch picks the smallest that fits the round's largest survivor: a CU then holds
// eight times as many waves for the usual few dozen hits per read as for the rare thousands)
template <int CAP>
struct kidx_sortwrite {
    enum { THREADS = 64 };
    static __device__ void run(const dp_scan_item* __restrict__ items, const uint32_t* __restrict__ sel,
                                                     const uint32_t* __restrict__ n_sel_p, const uint32_t* __restrict__ counts,
                                                     const uint64_t* __restrict__ segoff, int32_t* __restrict__ segs, int k,
                                                     uint32_t* __restrict__ overflow, uint32_t n_read_items, uint32_t n_extra,
                                                     uint32_t* __restrict__ head, int32_t* __restrict__ host_segs,
                                                     const uint64_t* __restrict__ totals, uint64_t seg_cap, uint32_t extras_only) {
    // extras_only (the dense regime: kidx_bin_sort_dense has sorted the read items' slices): the round's extra items alone - the last
    // n_extra entries of the survivor list (every extra item is on it)
    // host_segs (may be null): pinned host mirror of segs[] - the extra items' (query windows') segments are stored there as
    // well, at the same offsets: they are what the host wants of this pass, and no copy has to fetch them afterwards
    __shared__ unsigned long long keys[CAP];
    const int lane = dp_lane();
    const uint32_t n_sel = *n_sel_p;
    // the fill pass (previous launch) was the last reader of the extra items' lists: head[] back to all zero
    for (uint32_t e = blockIdx.x * 64 + lane; e < n_extra; e += gridDim.x * 64) head[items[n_read_items + e].read] = 0;
    // launched without a wait behind the counting step (round 4): the segment buffer was sized from the round before - a round
    // that needs more, or whose records did not fit, is filled and sorted again by the host's second attempt
    const bool gave_up = totals && (totals[0] > seg_cap || (uint32_t)totals[6] != 0u);
    const uint32_t sv0 = (extras_only && n_sel >= n_extra) ? n_sel - n_extra : 0u;
    for (uint32_t sv = sv0 + blockIdx.x; sv < n_sel && !gave_up; sv += gridDim.x) {
        const uint32_t it = sel[sv];
        kx_sort_one<CAP, true>(keys, lane, it, counts[it], segoff[it], (int)items[it].n_kmers, segs, k, overflow, n_read_items, host_segs);
    }
}
};

// Round 6 - the dense regime (the command's default k = 10: every read holds ~100 seed occurrences, ten million hits a round).  There
// kidx_bin_fill scattered 8-byte pairs into the reads' slices and kidx_sortwrite, a wave per read, read them back, sorted them and wrote
// them again: 80 MB of scattered stores, 80 MB read, 80 MB written, a launch of 100 k single-wave workgroups.  Here a bin is small (64
// reads, ~6 k records) and its workgroup does both in LDS: the bin's records are dealt to their reads' stretches of an LDS array (slot
// counters in LDS, as kidx_bin_fill's), a wave per read ranks its stretch (every key counts the keys below it - broadcast reads, no
// scratch: the ranks wait in registers until the wave has read everything) and the stretch goes out as [gap, seed, ..., gap],
// contiguous per read and per bin.  A bin with more records than the array holds takes its reads in groups (one more pass over the
// bin's records per group - the bins of the round's query reads).  A read with more than MAXC hits raises the overflow word, as a
// survivor beyond the sort pass's capacity always did (the host repeats fill + sort with the right tier).
// The blocks behind the bins fill the extra items' slices from their list, as kidx_bin_fill's; kidx_sortwrite(extras_only) sorts those.
template <int CAP>
struct kidx_bin_sort_dense {
    enum { THREADS = 512, WAVES = 8, RBMAX = 256, XBLOCKS = 8, MAXC = 1024, RPL = MAXC / 64 };
    // c <= 64 R keys of one read, sorted in place by one wave: every key counts the keys below it (broadcast reads of the stretch), the
    // ranks wait in registers until the wave has read everything, then the keys go to their ranks
    // (a read's hits lie at distinct positions - one k-mer starts at a position - so the position, the key's upper word, ranks alone:
    // 32-bit compares at full rate where the 64-bit ones took two issue slots each; the kernel is bound by exactly these - 65 M vector
    // instructions a launch at k = 10, profiles/r06/pmc_k10.json)
    template <int R>
    static __device__ __forceinline__ void rank_in_place(unsigned long long* my, const uint32_t c, const int lane) {
        unsigned long long key[R];
        uint32_t kp[R], rk[R];
        const uint32_t* pos = (const uint32_t*)my + 1;  // position of record l: pos[2 l]
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t j = (uint32_t)lane + 64u * (uint32_t)u;
            key[u] = j < c ? my[j] : ~0ull;
            kp[u] = (uint32_t)(key[u] >> 32);
            rk[u] = 0u;
        }
        uint32_t l = 0;
        for (; l + 8 <= c; l += 8) {  // (eight broadcast reads in flight: one read at a time is a trip through the LDS per key)
            uint32_t o[8];
#pragma unroll
            for (int x = 0; x < 8; x++) o[x] = pos[2u * (l + (uint32_t)x)];
#pragma unroll
            for (int x = 0; x < 8; x++) {
#pragma unroll
                for (int u = 0; u < R; u++) rk[u] += o[x] < kp[u] ? 1u : 0u;
            }
        }
        for (; l < c; l++) {
            const uint32_t o = pos[2u * l];
#pragma unroll
            for (int u = 0; u < R; u++) rk[u] += o < kp[u] ? 1u : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t j = (uint32_t)lane + 64u * (uint32_t)u;
            if (j < c) my[rk[u]] = key[u];  // (positions are distinct: the ranks are a permutation)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    static __device__ void run(const KxBins B, const dp_scan_item* __restrict__ items, uint32_t n_read_items,
                               const uint32_t* __restrict__ counts, uint32_t* __restrict__ fillc, const uint64_t* __restrict__ segoff,
                               int32_t* __restrict__ segs, const uint64_t* __restrict__ totals, uint64_t seg_cap, int k,
                               uint32_t* __restrict__ overflow) {
        __shared__ unsigned long long recs[CAP];
        __shared__ uint32_t cnt[RBMAX], offs[RBMAX + 1], cur[RBMAX], nks[RBMAX];
        __shared__ unsigned long long sout[RBMAX];
        __shared__ uint32_t g_hi_s;
        if (B.flags[0] || totals[0] > seg_cap) return;  // (the host repeats the fill with the bucket walk / a larger buffer)
        if (blockIdx.x >= B.n_bins) {  // the extra items' hits: a slot from the item's fill cursor, as ever
            const uint32_t xb = blockIdx.x - B.n_bins;
            if (xb >= XBLOCKS) return;
            const uint32_t nx = min(*B.xcursor, B.xcap);
            for (uint32_t j = xb * THREADS + threadIdx.x; j < nx; j += XBLOCKS * THREADS) {
                const uint4 x = B.xrec[j];
                if (counts[x.x] >= items[x.x].min_seeds) {
                    const uint32_t slot = atomicAdd(&fillc[x.x], 1u);
                    const uint64_t to = segoff[x.x] + 2ull * slot;
                    segs[to] = (int32_t)x.y;
                    segs[to + 1] = (int32_t)x.z;
                }
            }
            return;
        }
        const int lane = dp_lane(), wave = threadIdx.x >> 6;
        const uint32_t bin = blockIdx.x;
        const uint32_t rb = 1u << B.bshift, first = bin << B.bshift;
        const uint32_t n = min(B.cursor[bin], B.cap);
        for (uint32_t i = threadIdx.x; i < rb; i += THREADS) {
            const uint32_t it = first + i;
            uint32_t c = 0u;
            if (it < n_read_items) {
                // (everything a read's wave will want of memory, asked for at once: a wave sorts eight reads one after the other)
                const dp_scan_item item = items[it];
                const unsigned long long so = segoff[it];
                c = counts[it];
                if (c < item.min_seeds) c = 0u;
                nks[i] = item.n_kmers;
                sout[i] = so;
            }
            if (c > (uint32_t)MAXC || c > (uint32_t)CAP) {
                atomicExch(overflow, 1u);
                c = 0u;
            }
            cnt[i] = c;
        }
        __syncthreads();
        const unsigned long long* rec = B.rec + (size_t)bin * B.cap;
        uint32_t g_lo = 0u;
        while (g_lo < rb) {
            // the group: reads g_lo .. g_hi - 1, as many as the array holds (every read fits by itself: c <= CAP)
            if (wave == 0) {
                uint32_t acc = 0u, i = g_lo;
                for (;;) {
                    const uint32_t idx = i + (uint32_t)lane;
                    const uint32_t c = idx < rb ? cnt[idx] : 0u;
                    const uint32_t incl = (uint32_t)wave_incl_sum((int)c);
                    const bool fits = idx < rb && acc + incl <= (uint32_t)CAP;  // (a prefix of the lanes: the sums do not decrease)
                    const uint32_t nfit = (uint32_t)__popcll(__ballot(fits));
                    if (fits) offs[idx] = acc + incl - c;
                    if (nfit) acc += (uint32_t)__shfl((int)incl, (int)nfit - 1, 64);
                    i += nfit;
                    if (nfit < 64u) break;
                }
                if (lane == 0) {
                    offs[i] = acc;
                    g_hi_s = i;
                }
            }
            for (uint32_t i = threadIdx.x; i < rb; i += THREADS) cur[i] = 0u;
            __syncthreads();
            const uint32_t g_hi = g_hi_s;
            const uint32_t staged = offs[g_hi];
            if (staged) {
                for (uint32_t jb = 4u * threadIdx.x; jb < n; jb += 4u * THREADS) {
                    unsigned long long e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = jb + (uint32_t)u < n ? rec[jb + u] : ~0ull;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (e[u] == ~0ull) continue;
                        const uint32_t rib = (uint32_t)(e[u] >> (KXB_SEED_BITS + 24));
                        if (rib < g_lo || rib >= g_hi) continue;
                        const uint32_t c = cnt[rib];
                        if (!c) continue;
                        const uint32_t rank = atomicAdd(&cur[rib], 1u);
                        if (rank < c)  // key = position << 32 | seed: what kx_sort_one sorts
                            recs[offs[rib] + rank] = ((unsigned long long)((uint32_t)e[u] & 0xffffffu) << 32) |
                                                     (unsigned long long)((uint32_t)(e[u] >> 24) & ((1u << KXB_SEED_BITS) - 1u));
                    }
                }
            }
            __syncthreads();
            for (uint32_t i = g_lo + (uint32_t)wave; i < g_hi; i += WAVES) {
                const uint32_t c = cnt[i];
                if (!c) continue;
                unsigned long long* my = recs + offs[i];
                if (c <= 128u) rank_in_place<2>(my, c, lane);
                else if (c <= 256u) rank_in_place<4>(my, c, lane);
                else if (c <= 512u) rank_in_place<8>(my, c, lane);
                else rank_in_place<RPL>(my, c, lane);
                const uint64_t out = sout[i];
                const int nk = (int)nks[i];
                for (uint32_t j = lane; j < c; j += 64) {
                    const int p = (int)(my[j] >> 32);
                    const int prev = j ? (int)(my[j - 1] >> 32) : -k;  // "-k": the first gap is the hit's own index
                    segs[out + 2 * (uint64_t)j] = p - (prev + k);
                    segs[out + 2 * (uint64_t)j + 1] = (int32_t)(uint32_t)my[j];
                }
                if (lane == 0) segs[out + 2 * (uint64_t)c] = nk - (int)(my[c - 1] >> 32) - 1;  // final gap (sequence/asm_amd64.s:387-392)
            }
            __syncthreads();
            g_lo = g_hi;
        }
    }
};

// lanes per seed of kidx_walk: by the mean bucket size of the index (positions / 4^k)
static uint32_t kidx_lps(const dp_kindex* ix, int k) {
    const uint64_t nk = (uint64_t)1 << (2 * k);
    return ix->n_pos / nk >= 128 ? 64u : 16u;
}
static uint32_t kidx_walk_blocks(const dp_kindex* ix, int k, uint32_t S) {
    const uint32_t waves = kidx_lps(ix, k) == 64 ? S * KX_PARTS : (S + 3) / 4;
    return (waves + 3) / 4;
}

// workgroups of the count walk's launch: all of its work at once (default), or DP_KX_WALK_BLOCKS of them striding over it
static uint32_t kidx_walk_grid(const dp_kindex* ix, int k, uint32_t S) {
    static const uint32_t lim = (uint32_t)std::max(0L, dp_tune("kx_walk_blocks", 0));
    const uint32_t all = kidx_walk_blocks(ix, k, S);
    return lim ? std::min(lim, all) : all;
}

// Counting step of a round from the index: counts, segment offsets, compacted survivor list and totals for all items, with
// no host round trip.  d_work = [counts n | fill cursors n | tile status (tiles + 1) u64 | ticket, max count] (zeroed here).
// `one` (round 4, may be null): the whole index step in one go - the count pass keeps hit records, and behind the offsets scan the
// fill pass (from the records) and the sort/write pass are launched at once into a segment buffer sized from the round before;
// the caller waits once and repeats fill + sort (dp_kindex_refill) for the rare round that outgrew a guess.
int dp_kindex_count(dp_ctx* ctx, int k, const dp_scan_item* d_items, uint32_t lo, uint32_t hi, uint32_t n_read_items, uint32_t n_extra,
                    uint32_t* d_counts, uint64_t* d_segoff, uint32_t* s_item, uint32_t* s_count, uint64_t* s_off, uint4* s_pack,
                    uint64_t* d_totals, unsigned long long* host_totals, const dp_kindex_oneshot* one, const dp_kindex_fast* fast) {
    dp_kindex* ix = kidx_owner(ctx)->kidx;
    const uint32_t S = ctx->n_seeds, n_items = n_read_items + n_extra;
    if (n_items >= (1u << 24)) return 1;  // (the scan's status word holds 24 bits of survivors) -> scan kernels
    const uint32_t n_tiles = (n_items + KX_TILE * KX_IPT - 1) / (KX_TILE * KX_IPT);
    dp_ctx* ow = kidx_owner(ctx);
    // linked lists of the extra items: head per read (all zero outside a call), next per extra item
    {
        const void* before = ctx->d_kx_lo.p;
        if (dev_reserve(ctx, ctx->d_kx_lo, (size_t)ow->n_reads * 4 + 64)) return DP_ERR_HIP;
        if (ctx->d_kx_lo.p != before || ctx->kx_head_reads != ow->n_reads) {
            DP_HIP(hipMemsetAsync(ctx->d_kx_lo.p, 0, (size_t)ow->n_reads * 4, ctx->stream));
            ctx->kx_head_reads = ow->n_reads;
        }
    }
    if (dev_reserve(ctx, ctx->d_kx_vals, (size_t)n_extra * 4 + 64)) return DP_ERR_HIP;
    uint32_t* head = (uint32_t*)ctx->d_kx_lo.p;
    uint32_t* next = (uint32_t*)ctx->d_kx_vals.p;
    const size_t b_counts = (size_t)n_items * 4;
    if (dev_reserve(ctx, ctx->d_kx_sz, 2 * b_counts + ((size_t)n_tiles + 2) * 8 + 64 * 8 + (KX_MAXBINS + 2) * 4 + 128)) return DP_ERR_HIP;
    uint32_t* fillc = (uint32_t*)ctx->d_kx_sz.p;
    unsigned long long* status = (unsigned long long*)((uint8_t*)ctx->d_kx_sz.p + ((b_counts + 7) & ~(size_t)7));
    uint32_t* ticket = (uint32_t*)(status + n_tiles + 1);
    unsigned long long* n_hits = (unsigned long long*)(ticket + 2);  // [64]
    uint32_t* bin_cursor = (uint32_t*)(n_hits + 64);                 // [KX_MAXBINS] + the extra list's cursor (round 5, KxBins)
    const uint32_t n_work = (uint32_t)((((b_counts + 7) & ~(size_t)7) + ((size_t)n_tiles + 1) * 8 + 8 + 64 * 8 + (KX_MAXBINS + 2) * 4) / 4);
    DP_HIP(dp_mark(ctx, 0));
    {
        const uint32_t n_thr = std::max(std::max(n_work, n_items), std::max(n_extra, 16u));
        dp_launch<kidx_prepare>(ctx, dim3((n_thr + 255) / 256), dim3(256), (uint32_t*)ctx->d_kx_sz.p, n_work, d_counts, n_items,
                           (uint32_t*)d_totals, const_cast<dp_scan_item*>(d_items), n_read_items, n_extra, head, next,
                           ctx->extras_staged ? (const dp_scan_item*)ctx->h_extra.p : (const dp_scan_item*)nullptr);
        ctx->extras_staged = false;
    }
    // hit records: 64 shards of (estimated hits / 64) * 1.5 + 4096 records, two words per group
    KxRec R{nullptr, nullptr, nullptr, 0u, nullptr, 0u, 0u};
    if (fast && fast->ign) {
        R.ign = fast->ign;
        R.qlo = fast->qlo;
        R.qspan = fast->qspan;
    }
    const uint32_t lps = kidx_lps(ix, k);
    const uint32_t n_groups = lps == 64 ? S * KX_PARTS : S;
    if (one && S) {
        // (no round of this context yet: three times the seeds' share of all positions - seeds are the commoner k-mers)
        const uint64_t first_guess = (uint64_t)((double)S * (double)(ix->n_pos) / (double)((uint64_t)1 << (2 * k)) * 3.0);
        const uint64_t est = std::max<uint64_t>(one->hits_guess ? one->hits_guess : first_guess, 65536);
        const uint64_t shard = est / 64 + est / 128 + 4096;
        if (shard * 64 < 0xfffffff0ull) {
            if (dev_reserve(ctx, ctx->d_kx_keys, (size_t)shard * 64 * 8 + 64)) return DP_ERR_HIP;
            if (dev_reserve(ctx, ctx->d_kx_tmp, (size_t)n_groups * 8 + 64)) return DP_ERR_HIP;
            R.rec = (unsigned long long*)ctx->d_kx_keys.p;
            R.kb = (uint32_t*)ctx->d_kx_tmp.p;
            R.flags = (uint32_t*)(d_totals + 6);
            R.shard_cap = (uint32_t)shard;
        }
    }
    // round 5: hits binned by read range and counted in LDS instead of one memory-side atomic per hit (DP_KX_BINS=0: the records of
    // round 4).  Bins of 512 .. 16 k reads, at most 256 of them while that keeps a bin inside the LDS; the bins share the record
    // buffer of the shards (1.5 x the previous round's hits + 4096 each), the extra items' list lives in the group table's place.
    KxBins B{};
    const char* bins_env = getenv("DP_KX_BINS");  // (read per call: tests switch it between jobs of one process)
    const bool bins_on = !(bins_env && bins_env[0] == '0');
    // round 6, the dense regime (this context's previous round had 16 and more hits per read - k = 10: ~100): small bins whose
    // workgroup fills AND sorts in LDS (kidx_bin_sort_dense), eight trips of a walk workgroup per bin reservation, the count as a launch
    // of its own (one workgroup per bin instead of one per 4 096 reads inside kidx_offsets).  DP_KX_DENSE=0: off (A/B runs)
    const char* dense_env = getenv("DP_KX_DENSE");  // (read per call: tests switch it between jobs of one process)
    bool dense = false;
    if (one && R.rec && bins_on && n_read_items && S < (1u << KXB_SEED_BITS)) {
        uint32_t bshift = 9;
        while (((n_read_items + (1u << bshift) - 1) >> bshift) > 256 && bshift < KXB_RIB_BITS) bshift++;
        // (DP_KX_DENSE=1, tests: whatever the hit density - small inputs reach the dense kernels through it)
        dense = one->sort_cap <= 1024 && (dense_env && dense_env[0] == '1' ? true
                                          : !(dense_env && dense_env[0] == '0') && lps == 64 && one->hits_guess >= 16ull * n_read_items &&
                                            one->hits_guess >= (1ull << 20));
        if (dense) {
            bshift = 6;
            while (((n_read_items + (1u << bshift) - 1) >> bshift) > KX_MAXBINS && bshift < 8) bshift++;
            if (((n_read_items + (1u << bshift) - 1) >> bshift) > KX_MAXBINS) {  // (more than half a million reads: the bins of round 5)
                dense = false;
                bshift = 9;
                while (((n_read_items + (1u << bshift) - 1) >> bshift) > 256 && bshift < KXB_RIB_BITS) bshift++;
            }
        }
        const uint32_t n_bins = (n_read_items + (1u << bshift) - 1) >> bshift;
        const uint64_t total = (uint64_t)R.shard_cap * 64;
        const uint64_t xcap = std::max<uint64_t>(65536, total / 8);
        // a bin holds its share of the hits (1.5 x the round before, as the shards) - and ONE bin also holds the round's query reads,
        // consecutive reads that contain every seed of the round by construction: + 2 S (measured at config 2: mean 2.3 k records per
        // bin, 7.4 k in the query reads' bin)
        uint64_t cap = total / n_bins + std::min<uint64_t>(2 * (uint64_t)S, dense ? ((uint64_t)128 << bshift) : ~0ull) + 1024;
        if (const char* e = getenv("DP_KX_BINS_CAP")) cap = (uint64_t)std::max(16, atoi(e));  // (test hook: bins that overflow)
        if (n_bins <= KX_MAXBINS && cap < 0x7fffffffu) {
            if (dev_reserve(ctx, ctx->d_kx_tmp, std::max((size_t)n_groups * 8, (size_t)xcap * 16) + 64)) return DP_ERR_HIP;
            if (dev_reserve(ctx, ctx->d_kx_keys, (size_t)cap * n_bins * 8 + 64)) return DP_ERR_HIP;
            R.rec = (unsigned long long*)ctx->d_kx_keys.p;
            B.rec = R.rec;
            B.cursor = bin_cursor;
            B.flags = R.flags;
            B.cap = (uint32_t)cap;
            B.n_bins = n_bins;
            B.bshift = bshift;
            B.xrec = (uint4*)ctx->d_kx_tmp.p;
            B.xcursor = bin_cursor + KX_MAXBINS;
            B.xcap = (uint32_t)xcap;
            B.ign = R.ign;
            B.qlo = R.qlo;
            B.qspan = R.qspan;
            B.batch = 1u;
        } else {
            dense = false;
        }
    }
    static const bool kx_debug = dp_debug("kx");
    unsigned long long* dbg = nullptr;
    const size_t n_dbg_waves = (size_t)kidx_walk_blocks(ix, k, S) * 4;
    if (kx_debug) {
        DP_HIP(dp_dev_malloc((void**)&dbg, n_dbg_waves * 64));
        DP_HIP(hipMemsetAsync(dbg, 0, n_dbg_waves * 64, ctx->stream));
    }
    bool count_in_offsets = false;
    if (S && B.rec) {
        const uint32_t n_waves = kidx_walk_blocks(ix, k, S) * 4;
        static const int bin_waves = (int)dp_tune("kx_bin_waves", 8);
#define KX_WALK_BIN(W_)                                                                                                                            \
    dp_launch<kidx_walk_bin<W_>>(ctx, dim3((n_waves + W_ - 1) / W_), dim3(64 * W_), dp_seeds_ptr(ctx), S, (const uint64_t*)ix->off.p, ix->view(), \
                                 d_items, lo, hi, n_read_items, (const uint32_t*)head, (const uint32_t*)next, d_counts, n_hits, lps, B, n_waves)
        // (dense: 16 waves - four seeds a trip; measured at k = 10, five slots: 0.582 against 0.606 s per job, profiles/r06/k10_knobs.txt)
        if (dense) {
            // (trips per reservation: as many as make ONE workgroup per CU cover the round's waves - a 16-wave workgroup with its 105 VGPRs
            // is alone on its CU, and 313 of them on 256 CUs were two rounds of workgroups for 1.2 rounds of work)
            B.batch = std::max<uint32_t>(1u, std::min<uint32_t>(32u, (n_waves + 16u * 256u - 1) / (16u * 256u)));
            const uint32_t per = 16u * B.batch;
            dp_launch<kidx_walk_bin<16>>(ctx, dim3((n_waves + per - 1) / per), dim3(64 * 16), dp_seeds_ptr(ctx), S, (const uint64_t*)ix->off.p, ix->view(),
                                         d_items, lo, hi, n_read_items, (const uint32_t*)head, (const uint32_t*)next, d_counts, n_hits, lps, B, n_waves);
        } else if (bin_waves <= 4) KX_WALK_BIN(4);
        else if (bin_waves <= 8) KX_WALK_BIN(8);
        else KX_WALK_BIN(16);
#undef KX_WALK_BIN
        // bins no larger than a tile of kidx_offsets are counted by that kernel (one launch less per round; DP_KX_FUSE=0: as before.
        // Fill + sort in one launch for the sparse regime - DP_KX_FUSE=2 in round 5 - was slower and is gone: profiles/r05/ab12_fusions.txt)
        const char* fuse_env = getenv("DP_KX_FUSE");  // (read per call, like DP_KX_BINS: tests switch it between jobs of one process)
        const bool fuse_off = fuse_env && fuse_env[0] == '0';
        count_in_offsets = !fuse_off && !dense && (1u << B.bshift) <= KX_TILE * KX_IPT;
        if (count_in_offsets) {
        } else if (B.bshift <= 9)
            dp_launch<kidx_bin_count<512>>(ctx, dim3(B.n_bins), dim3(512), B, d_counts, n_read_items, lo);
        else if (B.bshift <= 12)
            dp_launch<kidx_bin_count<4096>>(ctx, dim3(B.n_bins), dim3(512), B, d_counts, n_read_items, lo);
        else
            dp_launch<kidx_bin_count<16384>>(ctx, dim3(B.n_bins), dim3(512), B, d_counts, n_read_items, lo);
        static const bool bins_debug = dp_debug("kx_bins");
        if (bins_debug) {  // (diagnosis: waits for the stream) how full the bins and the extra list are
            std::vector<uint32_t> cur((size_t)KX_MAXBINS + 2);
            uint64_t fl = 0;
            hipStreamSynchronize(ctx->stream);
            hipMemcpy(cur.data(), B.cursor, cur.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(&fl, d_totals + 6, 8, hipMemcpyDeviceToHost);
            uint32_t mx = 0;
            uint64_t sum = 0;
            for (uint32_t b = 0; b < B.n_bins; b++) mx = std::max(mx, cur[b]), sum += cur[b];
            fprintf(stderr, "[kx bins] seeds %u lps %u waves %u | %u bins of %u reads, cap %u each: fullest %u, records %llu | extra list %u of %u | flag %llu | hits guess %llu\n",
                    S, lps, n_waves, B.n_bins, 1u << B.bshift, B.cap, mx, (unsigned long long)sum, cur[KX_MAXBINS], B.xcap, (unsigned long long)fl,
                    (unsigned long long)(one ? one->hits_guess : 0));
        }
    } else if (S)
        dp_launch<kidx_walk<false>>(ctx, dim3(kidx_walk_grid(ix, k, S)), dim3(256), dp_seeds_ptr(ctx), S,
                           (const uint64_t*)ix->off.p, ix->view(), d_items, lo, hi, n_read_items, (const uint32_t*)head,
                           (const uint32_t*)next, d_counts, fillc, (const uint64_t*)nullptr, (int32_t*)nullptr, n_hits, kidx_lps(ix, k), dbg, R,
                           kidx_walk_blocks(ix, k, S) * 4);
    if (kx_debug) {
        std::vector<unsigned long long> h(n_dbg_waves * 8);
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(h.data(), dbg, n_dbg_waves * 64, hipMemcpyDeviceToHost);
        dp_dev_free(dbg);
        unsigned long long first = ~0ull, lastStart = 0, lastEnd = 0;
        double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
        size_t nw = 0;
        for (size_t w = 0; w < n_dbg_waves; w++) {
            const unsigned long long* t = &h[8 * w];
            if (!t[0] || !t[4]) continue;
            nw++;
            first = std::min(first, t[0]);
            lastStart = std::max(lastStart, t[0]);
            lastEnd = std::max(lastEnd, t[4]);
            for (int i = 1; i < 5; i++) {
                const double d = (double)(t[i] - t[i - 1]) / 100.0;
                sum[i] += d;
                mx[i] = std::max(mx[i], d);
            }
        }
        if (nw)
            fprintf(stderr, "[kx] %zu waves, first start .. last start %.1f us, first start .. last end %.1f us | us per phase mean/max: bounds %.1f/%.1f entries %.1f/%.1f items %.1f/%.1f counters %.1f/%.1f\n",
                    nw, (lastStart - first) / 100.0, (lastEnd - first) / 100.0, sum[1] / nw, mx[1], sum[2] / nw, mx[2], sum[3] / nw, mx[3], sum[4] / nw, mx[4]);
    }
    // totals[2] = seed occurrences in the read set, totals[3] = largest survivor count (both written by the kernel)
    {
        KxBins Bo = B;
        if (!count_in_offsets) Bo.rec = nullptr;
        dp_launch<kidx_offsets>(ctx, dim3(n_tiles), dim3(KX_TILE), d_items, d_counts, n_items, status, ticket,
                           d_segoff, s_item, s_count, s_off, s_pack, d_totals, n_read_items, (uint32_t*)(d_totals + 3),
                           (const unsigned long long*)n_hits, host_totals, Bo, lo);
    }
    DP_HIP(hipGetLastError());
    DP_HIP(dp_mark(ctx, 1));
    if (one && R.rec) {
        DP_HIP(dp_mark(ctx, 2));
        if (B.rec && dense) {
            dp_launch<kidx_bin_sort_dense<8192>>(ctx, dim3(B.n_bins + kidx_bin_sort_dense<8192>::XBLOCKS), dim3(512), B, d_items, n_read_items,
                                                 (const uint32_t*)d_counts, fillc, (const uint64_t*)d_segoff, one->d_segs, (const uint64_t*)d_totals,
                                                 one->seg_cap, k, (uint32_t*)(d_totals + 4));
        } else if (B.rec) {
            const dim3 fg(B.n_bins + kidx_bin_fill<512>::XBLOCKS), fb(512);
            if (B.bshift <= 9)
                dp_launch<kidx_bin_fill<512>>(ctx, fg, fb, B, d_items, n_read_items, (const uint32_t*)d_counts, fillc, (const uint64_t*)d_segoff,
                                              one->d_segs, (const uint64_t*)d_totals, one->seg_cap);
            else if (B.bshift <= 12)
                dp_launch<kidx_bin_fill<4096>>(ctx, fg, fb, B, d_items, n_read_items, (const uint32_t*)d_counts, fillc, (const uint64_t*)d_segoff,
                                               one->d_segs, (const uint64_t*)d_totals, one->seg_cap);
            else
                dp_launch<kidx_bin_fill<16384>>(ctx, fg, fb, B, d_items, n_read_items, (const uint32_t*)d_counts, fillc, (const uint64_t*)d_segoff,
                                                one->d_segs, (const uint64_t*)d_totals, one->seg_cap);
        } else
        dp_launch<kidx_fill_rec>(ctx, dim3(kidx_walk_blocks(ix, k, S)), dim3(256), R, n_groups, lps, d_items, lo, n_read_items, one->min_seeds,
                                 (const uint32_t*)head, (const uint32_t*)next, (const uint32_t*)d_counts, fillc, (const uint64_t*)d_segoff,
                                 one->d_segs, (const uint64_t*)d_totals, one->seg_cap);
        // (the survivor count is on the device only: the grid covers the most survivors a round of this context has had so far,
        // the kernel strides over the rest)
        const dim3 sg(dense ? std::max<uint32_t>(64u, std::min<uint32_t>(n_extra, 16384))
                            : std::max<uint32_t>(256u, std::min<uint32_t>(one->surv_guess, 16384))), sb(64);
        uint32_t* ovf = (uint32_t*)(d_totals + 4);
        const uint32_t* nsp = (const uint32_t*)(d_totals + 1);
        if (one->sort_cap <= 256)
            dp_launch<kidx_sortwrite<256>>(ctx, sg, sb, d_items, (const uint32_t*)s_item, nsp, (const uint32_t*)d_counts, (const uint64_t*)d_segoff,
                                           one->d_segs, k, ovf, n_read_items, n_extra, head, one->host_segs, (const uint64_t*)d_totals, one->seg_cap,
                                           dense ? 1u : 0u);
        else if (one->sort_cap <= 1024)
            dp_launch<kidx_sortwrite<1024>>(ctx, sg, sb, d_items, (const uint32_t*)s_item, nsp, (const uint32_t*)d_counts, (const uint64_t*)d_segoff,
                                            one->d_segs, k, ovf, n_read_items, n_extra, head, one->host_segs, (const uint64_t*)d_totals, one->seg_cap,
                                            dense ? 1u : 0u);
        else
            dp_launch<kidx_sortwrite<KX_SORT_LDS>>(ctx, sg, sb, d_items, (const uint32_t*)s_item, nsp, (const uint32_t*)d_counts,
                                                   (const uint64_t*)d_segoff, one->d_segs, k, ovf, n_read_items, n_extra, head, one->host_segs,
                                                   (const uint64_t*)d_totals, one->seg_cap, 0u);
        DP_HIP(hipGetLastError());
        DP_HIP(dp_mark(ctx, 3));
    }
    (void)k;
    return (one && !R.rec) ? 2 : DP_OK;  // 2: the one-go form was asked for and could not be launched (no seeds / too many records)
}

// Second attempt of a round whose one-go fill + sort gave up (segment buffer or record shards too small, a survivor with more hits
// than the guessed sort holds): the extra items are linked to their reads again (the sort pass has unlinked them), the fill cursors
// go back to zero, then the bucket-walking fill and the sort with the right capacity - dp_kindex_write as before round 4.
int dp_kindex_refill(dp_ctx* ctx, const dp_scan_item* d_items, uint32_t n_read_items, uint32_t n_extra) {
    const uint32_t n_items = n_read_items + n_extra;
    uint32_t* head = (uint32_t*)ctx->d_kx_lo.p;
    uint32_t* next = (uint32_t*)ctx->d_kx_vals.p;
    const dp_zero_region z = {ctx->d_kx_sz.p, (size_t)n_items * 4};
    if (int rc = dp_zero_regions(ctx, &z, 1)) return rc;
    if (n_extra) hipLaunchKernelGGL(kidx_link_extra, dim3((n_extra + 255) / 256), dim3(256), 0, ctx->stream, d_items, n_read_items, n_extra, head, next);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

// Second half, once the caller knows the totals and has sized d_segs: fill + sort/write.  Returns 1 when a survivor has more
// hits than the LDS sort holds (the caller answers the round with the scan kernels instead).
int dp_kindex_write(dp_ctx* ctx, int k, const dp_scan_item* d_items, uint32_t lo, uint32_t hi, uint32_t n_read_items, uint32_t n_extra,
                    const uint32_t* d_sel, uint32_t n_sel, uint32_t max_count, const uint32_t* d_counts, const uint64_t* d_segoff,
                    const uint64_t* d_totals, int32_t* d_segs, int32_t* host_segs) {
    dp_kindex* ix = kidx_owner(ctx)->kidx;
    dp_ctx* ow = kidx_owner(ctx);
    const uint32_t S = ctx->n_seeds;
    uint32_t* head = (uint32_t*)ctx->d_kx_lo.p;
    uint32_t* next = (uint32_t*)ctx->d_kx_vals.p;
    uint32_t* fillc = (uint32_t*)ctx->d_kx_sz.p;
    (void)ow;
    int rc = DP_OK;
    if (max_count > KX_SORT_LDS) {
        rc = 1;
    } else if (n_sel) {
        if (S)
            dp_launch<kidx_walk<true>>(ctx, dim3(kidx_walk_blocks(ix, k, S)), dim3(256), dp_seeds_ptr(ctx), S,
                               (const uint64_t*)ix->off.p, ix->view(), d_items, lo, hi, n_read_items, (const uint32_t*)head,
                               (const uint32_t*)next, (uint32_t*)d_counts, fillc, d_segoff, d_segs, (unsigned long long*)nullptr, kidx_lps(ix, k),
                               (unsigned long long*)nullptr, KxRec{nullptr, nullptr, nullptr, 0u, nullptr, 0u, 0u}, kidx_walk_blocks(ix, k, S) * 4);
        const dim3 sg(std::min<uint32_t>(n_sel, 16384)), sb(64);
        uint32_t* ovf = (uint32_t*)(d_totals + 4);
        const uint32_t* nsp = (const uint32_t*)(d_totals + 1);
        if (max_count <= 128)
            dp_launch<kidx_sortwrite<256>>(ctx, sg, sb, d_items, d_sel, nsp, d_counts, d_segoff, d_segs, k, ovf, n_read_items,
                               n_extra, head, host_segs, (const uint64_t*)nullptr, (uint64_t)0, 0u);
        else if (max_count <= 512)
            dp_launch<kidx_sortwrite<1024>>(ctx, sg, sb, d_items, d_sel, nsp, d_counts, d_segoff, d_segs, k, ovf, n_read_items,
                               n_extra, head, host_segs, (const uint64_t*)nullptr, (uint64_t)0, 0u);
        else
            dp_launch<kidx_sortwrite<KX_SORT_LDS>>(ctx, sg, sb, d_items, d_sel, nsp, d_counts, d_segoff, d_segs, k, ovf,
                               n_read_items, n_extra, head, host_segs, (const uint64_t*)nullptr, (uint64_t)0, 0u);
        DP_HIP(hipGetLastError());
    }
    if (n_extra && (rc != DP_OK || !n_sel))  // (otherwise the sort kernel has put head[] back to zero)
        dp_launch<kidx_unlink_extra>(ctx, dim3((n_extra + 255) / 256), dim3(256), d_items, n_read_items, n_extra, head);
    DP_HIP(hipGetLastError());
    return rc;
}
