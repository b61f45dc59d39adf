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

struct dp_kindex {
    std::mutex mu;
    int k = 0;
    bool built = false;
    bool unavailable = false;  // k too large for a direct-addressed table, or not enough free HBM: callers scan instead
    uint64_t n_pos = 0;
    DevBuf off;  // uint64 [4^k + 1]
    DevBuf pos;  // uint64 [n_pos]: absolute k-mer start (boff[read]*4 + position), grouped by k-mer value
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
                pos[off[kmer] + slot] = a;
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
    {
        // resident: 8 B per k-mer start + 8 B per table entry; transient: two 4-byte count tables.  Leave 4 GiB for the rounds.
        size_t free_b = 0, total_b = 0;
        hipMemGetInfo(&free_b, &total_b);
        const uint64_t need = ow->total_bases * 8 + (uint64_t)nk * 16 + ((uint64_t)4 << 30);
        int max_k = 14;
        if (const char* e = getenv("DP_KINDEX_MAX_K")) max_k = std::min(14, atoi(e));
        if (k > max_k || need > free_b + ix->pos.cap + ix->off.cap) {
            ix->unavailable = true;
            return 1;
        }
    }
    // count -> offsets -> scatter, on the CALLER's stream (the owner's buffers are only written here, under the mutex)
    void *d_counts = nullptr, *d_tmp = nullptr, *d_counts1 = nullptr;
    struct Temps {  // released on every way out, error returns included
        void **a, **b, **c;
        ~Temps() {
            for (void** p : {a, b, c})
                if (*p) hipFree(*p);
        }
    } temps{&d_counts, &d_tmp, &d_counts1};
    if (ow->d_kcounts && ow->kcounts_k == k) {  // dp_kmer_values counted exactly these k-mers already
        d_counts = ow->d_kcounts;
        ow->d_kcounts = nullptr;
        ow->kcounts_k = 0;
    } else {
        DP_HIP(hipMalloc(&d_counts, nk * 4));
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
    DP_HIP(hipMalloc(&d_tmp, tmp_bytes + 16));
    // nk + 1 outputs: the extra input element is never added into an output, but it must be readable -> the buffer holds nk+1
    DP_HIP(hipMalloc(&d_counts1, (nk + 1) * 4));
    DP_HIP(hipMemcpyAsync(d_counts1, d_counts, nk * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DP_HIP(hipMemsetAsync((uint8_t*)d_counts1 + nk * 4, 0, 4, ctx->stream));
    DP_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, (uint32_t*)d_counts1, (uint64_t*)ix->off.p, (uint64_t)0, nk + 1,
                                   rocprim::plus<uint64_t>(), ctx->stream));
    uint64_t total = 0;
    DP_HIP(hipMemcpyAsync(&total, (uint64_t*)ix->off.p + nk, 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    if (dev_reserve(ctx, ix->pos, total * 8 + 64)) return DP_ERR_HIP;
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

void dp_kindex_free(dp_ctx* ctx) {
    if (ctx->owner || !ctx->kidx) return;
    if (ctx->kidx->off.p) hipFree(ctx->kidx->off.p);
    if (ctx->kidx->pos.p) hipFree(ctx->kidx->pos.p);
    delete ctx->kidx;
    ctx->kidx = nullptr;
}

// ---- per round -------------------------------------------------------------------------------------------------------

__global__ void kidx_seed_sizes(const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off,
                                uint32_t* __restrict__ sz) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seeds) return;
    sz[s] = s < n_seeds ? (uint32_t)(off[(uint64_t)seeds[s] + 1] - off[seeds[s]]) : 0u;
}

__global__ void kidx_emit(const uint32_t* __restrict__ seeds, uint32_t n_seeds, const uint64_t* __restrict__ off,
                          const uint64_t* __restrict__ pos, const uint64_t* __restrict__ base, uint64_t* __restrict__ keys,
                          uint32_t* __restrict__ vals) {
    const int lane = dp_lane();
    const uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_seeds) return;
    const uint64_t o = off[seeds[s]];
    const uint32_t n = (uint32_t)(off[(uint64_t)seeds[s] + 1] - o);
    const uint64_t b = base[s];
    for (uint32_t i = lane; i < n; i += 64) {
        keys[b + i] = pos[o + i];
        vals[b + i] = s;
    }
}

__device__ __forceinline__ uint32_t kidx_lower_bound(const uint64_t* __restrict__ keys, uint32_t n, uint64_t x) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// counts[it] = seed occurrences whose start lies in the item's k-mer range; hit_lo[it] = index of the first one
__global__ void kidx_count_items(const dp_scan_item* __restrict__ items, uint32_t n_items, const uint64_t* __restrict__ boff,
                                 const uint64_t* __restrict__ keys, uint32_t n_hits, uint32_t* __restrict__ counts,
                                 uint32_t* __restrict__ hit_lo) {
    const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const dp_scan_item item = items[it];
    if (item.n_kmers == 0) {
        counts[it] = 0;
        hit_lo[it] = 0;
        return;
    }
    const uint64_t a0 = boff[item.read] * 4 + item.start;
    const uint32_t lo = kidx_lower_bound(keys, n_hits, a0);
    const uint32_t hi = kidx_lower_bound(keys, n_hits, a0 + item.n_kmers);
    counts[it] = hi - lo;
    hit_lo[it] = lo;
}

// [gap, seed, ..., gap] of every surviving item straight from its slice of the sorted occurrence list
__global__ void kidx_write(const dp_scan_item* __restrict__ items, const uint32_t* __restrict__ sel, uint32_t n_sel,
                           const uint64_t* __restrict__ boff, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                           const uint32_t* __restrict__ counts, const uint32_t* __restrict__ hit_lo, const uint64_t* __restrict__ segoff,
                           int32_t* __restrict__ segs, int k) {
    const int lane = dp_lane();
    const uint32_t sv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (sv >= n_sel) return;
    const uint32_t it = sel[sv];
    const dp_scan_item item = items[it];
    const uint64_t a0 = boff[item.read] * 4 + item.start;
    const uint32_t c = counts[it], lo = hit_lo[it];
    const uint64_t out = segoff[it];
    for (uint32_t j = lane; j < c; j += 64) {
        const int p = (int)(keys[lo + j] - a0);
        const int prev = j ? (int)(keys[lo + j - 1] - a0) : -k;  // "-k": the first gap is the hit's own index
        segs[out + 2 * (uint64_t)j] = p - (prev + k);
        segs[out + 2 * (uint64_t)j + 1] = (int32_t)vals[lo + j];
    }
    if (lane == 0) {
        const int last = c ? (int)(keys[lo + c - 1] - a0) : -k;
        segs[out + 2 * (uint64_t)c] = (int)item.n_kmers - last - 1;  // final gap (sequence/asm_amd64.s:387-392)
    }
}

// Counting step of a round from the index: fills counts (and the per-item slice starts) for all items.
// Returns 1 (nothing written) when the round's seeds have more than 2^31 occurrences in the read set - the sort keys and
// the per-item slices are 32-bit indexed; the caller then answers this round with the scan kernels.
int dp_kindex_count(dp_ctx* ctx, int k, const dp_scan_item* d_items, uint32_t n_items, uint32_t* d_counts, float* ms) {
    dp_kindex* ix = kidx_owner(ctx)->kidx;
    const uint32_t S = ctx->n_seeds;
    if (dev_reserve(ctx, ctx->d_kx_sz, ((size_t)S + 2) * 12 + 16)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_kx_lo, (size_t)n_items * 4 + 16)) return DP_ERR_HIP;
    uint64_t* base = (uint64_t*)ctx->d_kx_sz.p;  // 64-bit running totals: dense seed batches over tens of Gbase pass 2^32
    uint32_t* sz = (uint32_t*)(base + S + 2);
    DP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(kidx_seed_sizes, dim3((S + 1 + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_seeds.p, S,
                       (const uint64_t*)ix->off.p, sz);
    size_t tb = 0;
    rocprim::exclusive_scan(nullptr, tb, sz, base, (uint64_t)0, (size_t)S + 1, rocprim::plus<uint64_t>(), ctx->stream);
    if (dev_reserve(ctx, ctx->d_kx_tmp, tb + 64)) return DP_ERR_HIP;
    DP_HIP(rocprim::exclusive_scan(ctx->d_kx_tmp.p, tb, sz, base, (uint64_t)0, (size_t)S + 1, rocprim::plus<uint64_t>(), ctx->stream));
    if (pin_reserve(ctx, ctx->h_total, 32)) return DP_ERR_HIP;
    DP_HIP(hipMemcpyAsync((uint8_t*)ctx->h_total.p + 16, base + S, 8, hipMemcpyDeviceToHost, ctx->stream));
    DP_HIP(dp_stream_sync(ctx));
    const uint64_t H64 = *(const uint64_t*)((const uint8_t*)ctx->h_total.p + 16);
    uint64_t cap = (uint64_t)1 << 31;
    if (const char* e = getenv("DP_KINDEX_MAX_HITS")) cap = std::min<uint64_t>(cap, strtoull(e, nullptr, 10));  // (tests)
    if (H64 > cap) return 1;
    const uint32_t H = (uint32_t)H64;
    ctx->kx_hits = H;
    if (dev_reserve(ctx, ctx->d_kx_keys, ((size_t)H + 16) * 8 * 2)) return DP_ERR_HIP;
    if (dev_reserve(ctx, ctx->d_kx_vals, ((size_t)H + 16) * 4 * 2)) return DP_ERR_HIP;
    uint64_t* keys = (uint64_t*)ctx->d_kx_keys.p;
    uint64_t* keys2 = keys + H + 16;
    uint32_t* vals = (uint32_t*)ctx->d_kx_vals.p;
    uint32_t* vals2 = vals + H + 16;
    if (H) {
        hipLaunchKernelGGL(kidx_emit, dim3((S + 3) / 4), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_seeds.p, S,
                           (const uint64_t*)ix->off.p, (const uint64_t*)ix->pos.p, (const uint64_t*)base, keys, vals);
        // positions are below packed_bytes*4: sort only the bits that can differ
        unsigned bits = 1;
        while (bits < 64 && ((kidx_owner(ctx)->packed_bytes * 4) >> bits) != 0) bits++;
        size_t sb = 0;
        rocprim::radix_sort_pairs(nullptr, sb, keys, keys2, vals, vals2, (size_t)H, 0u, bits, ctx->stream);
        if (dev_reserve(ctx, ctx->d_kx_tmp, std::max(sb, tb) + 64)) return DP_ERR_HIP;
        DP_HIP(rocprim::radix_sort_pairs(ctx->d_kx_tmp.p, sb, keys, keys2, vals, vals2, (size_t)H, 0u, bits, ctx->stream));
    }
    hipLaunchKernelGGL(kidx_count_items, dim3((n_items + 255) / 256), dim3(256), 0, ctx->stream, d_items, n_items,
                       (const uint64_t*)ctx->d_boff.p, (const uint64_t*)keys2, H, d_counts, (uint32_t*)ctx->d_kx_lo.p);
    DP_HIP(hipGetLastError());
    DP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    (void)ms;
    (void)k;
    return DP_OK;
}

int dp_kindex_write(dp_ctx* ctx, int k, const dp_scan_item* d_items, const uint32_t* d_sel, uint32_t n_sel,
                    const uint32_t* d_counts, const uint64_t* d_segoff, int32_t* d_segs) {
    const uint32_t H = ctx->kx_hits;
    const uint64_t* keys2 = (const uint64_t*)ctx->d_kx_keys.p + H + 16;
    const uint32_t* vals2 = (const uint32_t*)ctx->d_kx_vals.p + H + 16;
    if (n_sel == 0) return DP_OK;
    hipLaunchKernelGGL(kidx_write, dim3((n_sel + 3) / 4), dim3(256), 0, ctx->stream, d_items, d_sel, n_sel,
                       (const uint64_t*)ctx->d_boff.p, keys2, vals2, d_counts, (const uint32_t*)ctx->d_kx_lo.p, d_segoff, d_segs, k);
    DP_HIP(hipGetLastError());
    return DP_OK;
}
