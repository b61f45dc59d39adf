// libdownpore_hip.so — the k-mer value table of `downpore overlap` / `downpore map` on the device.
// commands/overlap.go:55-93 (value per k-mer from its frequency) + util/sequtil/kmers.go:87-112 (fwd+rc merge, the 1 %
// most frequent k-mers get value 0).  On the host this is ~8 passes over 4^k-entry arrays with a random-access merge
// (1.8 s at k=13); here it is a histogram, one radix sort and two streaming kernels, and the table stays resident for
// dp_select_seeds.  Bit-identical to the host function (float64, no contraction).
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "dp_common.h"

__device__ __forceinline__ uint32_t values_rc_kmer(uint32_t seed, int k) {
    uint32_t x = ~seed;
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);
    return k >= 16 ? x : (x >> (32 - 2 * k));
}

struct ValuesToU64 {
    __host__ __device__ uint64_t operator()(uint32_t c) const { return (uint64_t)c; }
};
struct ValuesIsTie {
    uint64_t T;
    __host__ __device__ uint32_t operator()(uint64_t m) const { return m == T ? 1u : 0u; }
};

// value of every k-mer from its own count (overlap.go:73-88) and the merged fwd+rc count the 1 % cut looks at.
// kmers.go:90-96 merges in place while it walks the table: a pair (i, rc) is visited twice, so both entries end at
// 2*(count[i] + count[rc]); a palindrome is visited once and ends at 2*count[i].
__global__ void values_kernel(const uint32_t* __restrict__ counts, uint64_t n, int k, double tf, double* __restrict__ values,
                              uint64_t* __restrict__ merged) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = counts[i];
    const uint32_t rc = values_rc_kmer((uint32_t)i, k);
    merged[i] = rc == (uint32_t)i ? 2ull * a : 2ull * ((uint64_t)a + counts[rc]);
    const double targetFreq = 0.000005;
    const double freq = (double)a / tf;
    double v;
    if (a < 3) v = 0.0;
    else if (freq <= targetFreq) v = 1.0 - (targetFreq - freq);
    else v = 1.0 - (freq - targetFreq);
    values[i] = v;
}

// [0] = first index with sorted[i] >= T, [1] = first index with sorted[i] > T
__global__ void values_bounds_kernel(const uint64_t* __restrict__ sorted, uint64_t n, uint64_t T, uint64_t* __restrict__ out) {
    if (threadIdx.x > 1 || blockIdx.x) return;
    const bool upper = threadIdx.x == 1;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t v = sorted[mid];
        if (upper ? v <= T : v < T) lo = mid + 1;
        else hi = mid;
    }
    out[threadIdx.x] = lo;
}

// the most frequent k-mers lose their value: every merged count above T, and of those equal to T the ones with the
// highest k-mer ids (tie rank counted from the top; DESIGN.md 2, "tie rule")
__global__ void values_cut_kernel(const uint64_t* __restrict__ merged, const uint32_t* __restrict__ tieRank, uint64_t n, uint64_t T,
                                  uint32_t keepTies, double* __restrict__ values) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = merged[i];
    if (m > T || (m == T && tieRank[i] >= keepTies) || i == 0) values[i] = 0.0;
}

__global__ void values_zero_first(double* __restrict__ values) { values[0] = 0.0; }

// ---- round 4: the 1 % cut without a sort ---------------------------------------------------------------------------------
// The cut needs three numbers - T = the merged count at rank n - n/100 of the ascending order, how many k-mers lie above it, and
// which of the k-mers AT it are cut (the ones with the highest ids: the tie rule) - not the order itself.  The radix sort of
// 4^k 8-byte keys was 2.5 of the table's 4.1 ms.  Instead: the value pass also counts the merged counts below VH_BINS into a
// histogram (LDS per workgroup, its non-empty bins added to the global one), one workgroup reads T and the tie quota off it, a
// counting pass over the ties finds the k-mer id X from which ties are cut, and the cut is (m > T) | (m == T & i >= X).
// No wait in between (the total, T and X stay on the device).  T beyond the histogram (a k-mer set where one in a hundred
// k-mers occurs more than a thousand times): the sort path as before.
#define VH_BINS 2048
#define VH_ITEMS 16  // k-mers per thread of the value pass
// small[0] = total count (input), small[1] = T, small[2] = keepTies, small[3] = 1: T lies beyond the histogram,
// small[4] = X (first k-mer id whose tie is cut; n: none), small[5] = ties in front of the tile that holds X
__global__ __launch_bounds__(1024) void values_kernel_hist(const uint32_t* __restrict__ counts, uint64_t n, int k,
                                                           const uint64_t* __restrict__ small, double* __restrict__ values,
                                                           uint64_t* __restrict__ merged, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[VH_BINS + 1];
    for (int b = threadIdx.x; b <= VH_BINS; b += 1024) h[b] = 0;
    __syncthreads();
    const double tf = (double)small[0];
    const uint64_t base = (uint64_t)blockIdx.x * (1024 * VH_ITEMS);
#pragma unroll 4
    for (int u = 0; u < VH_ITEMS; u++) {
        const uint64_t i = base + (uint64_t)u * 1024 + threadIdx.x;
        if (i >= n) break;
        const uint32_t a = counts[i];
        const uint32_t rc = values_rc_kmer((uint32_t)i, k);
        const uint64_t m = rc == (uint32_t)i ? 2ull * a : 2ull * ((uint64_t)a + counts[rc]);
        merged[i] = m;
        const double targetFreq = 0.000005;
        const double freq = (double)a / tf;
        double v;
        if (a < 3) v = 0.0;
        else if (freq <= targetFreq) v = 1.0 - (targetFreq - freq);
        else v = 1.0 - (freq - targetFreq);
        values[i] = v;
        atomicAdd(&h[m < VH_BINS ? (uint32_t)m : VH_BINS], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= VH_BINS; b += 1024)
        if (h[b]) atomicAdd(&ghist[b], h[b]);
}

// The same pass with the reverse complement's count read from LDS instead of gathered (round 5).  counts[rc(i)] of 64 consecutive
// k-mers are 64 different cache lines (the last bases of i are the first of rc(i)): 4 useful bytes per 64-byte sector, 4.3 GB of
// traffic for a 268 MB table at k = 13 and three quarters of this kernel's 1.36 ms.  Split a k-mer into P (first three bases), x
// (the middle) and S (last three): the 64 x 64 k-mers with one x are 64 rows of 256 contiguous bytes, and their reverse complements
// are exactly the 64 x 64 k-mers with the middle rc(x) - rows of 256 bytes again.  A workgroup loads both tiles into LDS (rows
// padded to 65 words: the transposed read is conflict-free) and writes merged[] and values[] of its tile in rows.  VT_PER tiles per
// workgroup keep the number of flushes of the LDS histogram where it was.
#define VT_PER 4
__global__ __launch_bounds__(256) void values_kernel_hist_tiled(const uint32_t* __restrict__ counts, int k, const uint64_t* __restrict__ small,
                                                                double* __restrict__ values, uint64_t* __restrict__ merged,
                                                                uint32_t* __restrict__ ghist) {
    __shared__ uint32_t h[VH_BINS + 1];
    __shared__ uint32_t A[64][65], B[64][65];
    for (int b = threadIdx.x; b <= VH_BINS; b += 256) h[b] = 0;
    const double tf = (double)small[0];
    const int mid = k - 6;                            // bases of x
    const uint32_t n_tiles = 1u << (2 * mid);
    const int psh = 2 * k - 6;
    const uint32_t S = threadIdx.x & 63u, rS = values_rc_kmer(S, 3);
    for (int t = 0; t < VT_PER; t++) {
        const uint32_t x = blockIdx.x * VT_PER + t;
        if (x >= n_tiles) break;
        const uint32_t xr = mid ? values_rc_kmer(x, mid) : 0u;
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const uint32_t P = (uint32_t)it * 4 + (threadIdx.x >> 6);
            A[P][S] = counts[((size_t)P << psh) | ((size_t)x << 6) | S];
            B[P][S] = counts[((size_t)P << psh) | ((size_t)xr << 6) | S];
        }
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const uint32_t P = (uint32_t)it * 4 + (threadIdx.x >> 6);
            const size_t i = ((size_t)P << psh) | ((size_t)x << 6) | S;
            const uint32_t rP = values_rc_kmer(P, 3);
            const size_t ri = ((size_t)rS << psh) | ((size_t)xr << 6) | rP;
            const uint32_t a = A[P][S];
            const uint64_t m = ri == i ? 2ull * a : 2ull * ((uint64_t)a + B[rS][rP]);
            merged[i] = m;
            const double targetFreq = 0.000005;
            const double freq = (double)a / tf;
            double v;
            if (a < 3) v = 0.0;
            else if (freq <= targetFreq) v = 1.0 - (targetFreq - freq);
            else v = 1.0 - (freq - targetFreq);
            values[i] = v;
            atomicAdd(&h[m < VH_BINS ? (uint32_t)m : VH_BINS], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= VH_BINS; b += 256)
        if (h[b]) atomicAdd(&ghist[b], h[b]);
}

// one workgroup: T, the tie quota, "beyond the histogram"
__global__ __launch_bounds__(1024) void values_threshold_kernel(const uint32_t* __restrict__ ghist, uint64_t n, uint64_t topN,
                                                                uint64_t* __restrict__ small) {
    __shared__ unsigned long long part[1024];
    // thread t owns bins [2t, 2t+1]; cumulative counts by a block scan
    const int b0 = 2 * threadIdx.x;
    const unsigned long long c0 = b0 < VH_BINS ? ghist[b0] : 0ull, c1 = b0 + 1 < VH_BINS ? ghist[b0 + 1] : 0ull;
    part[threadIdx.x] = c0 + c1;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x], before = incl - c0 - c1;
    // T = the smallest t with #(m <= t) >= n - topN + 1 (the element at index n - topN of the ascending order)
    const unsigned long long want = n - topN + 1;
    if (threadIdx.x == 0) small[3] = part[1023] < want ? 1ull : 0ull;  // the rank lies in the overflow bin
    if (before < want && incl >= want) {
        const bool first = before + c0 >= want;
        const unsigned long long T = first ? (unsigned long long)b0 : (unsigned long long)b0 + 1;
        const unsigned long long below = first ? before : before + c0;          // #(m < T)
        const unsigned long long ties = first ? c0 : c1;                        // #(m == T)
        const unsigned long long above = n - below - ties;                      // #(m > T)
        const unsigned long long zeroTies = topN - above;                       // >= 1
        small[1] = T;
        small[2] = ties - zeroTies;                                            // keepTies: ties with a rank below this keep their value
    }
}

// ties per tile of 16 384 k-mers
__global__ __launch_bounds__(1024) void values_tie_count_kernel(const uint64_t* __restrict__ merged, uint64_t n,
                                                                const uint64_t* __restrict__ small, uint32_t* __restrict__ tileTies) {
    __shared__ uint32_t wsum[16];
    if (small[3]) return;
    const uint64_t T = small[1];
    const uint64_t base = (uint64_t)blockIdx.x * (1024 * VH_ITEMS);
    uint32_t c = 0;
#pragma unroll 4
    for (int u = 0; u < VH_ITEMS; u++) {
        const uint64_t i = base + (uint64_t)u * 1024 + threadIdx.x;
        if (i < n && merged[i] == T) c++;
    }
    c = (uint32_t)wave_sum((int)c);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; w++) t += wsum[w];
        tileTies[blockIdx.x] = t;
    }
}

// one workgroup: X = id of the tie with rank keepTies (ranks count from k-mer 0); n when every tie keeps its value
__global__ __launch_bounds__(1024) void values_tie_locate_kernel(const uint64_t* __restrict__ merged, uint64_t n, uint32_t n_tiles,
                                                                 const uint32_t* __restrict__ tileTies, uint64_t* __restrict__ small) {
    __shared__ unsigned long long part[1024];
    __shared__ uint32_t tile_s;
    __shared__ unsigned long long before_s;
    if (small[3]) return;
    const uint64_t T = small[1], keep = small[2];
    // tiles: thread t sums a contiguous slice, block scan over the slices, the slice that holds rank `keep` is walked by its thread
    const uint32_t per = (n_tiles + 1023) / 1024;
    const uint32_t lo = min(n_tiles, threadIdx.x * per), hi = min(n_tiles, lo + per);
    unsigned long long s = 0;
    for (uint32_t t = lo; t < hi; t++) s += tileTies[t];
    part[threadIdx.x] = s;
    if (threadIdx.x == 0) tile_s = 0xffffffffu;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x];
    unsigned long long run = incl - s;
    if (run <= keep && keep < incl) {
        for (uint32_t t = lo; t < hi; t++) {
            if (keep < run + tileTies[t]) {
                tile_s = t;
                before_s = run;
                break;
            }
            run += tileTies[t];
        }
    }
    __syncthreads();
    if (tile_s == 0xffffffffu) {  // keep == all ties: nothing at T is cut
        if (threadIdx.x == 0) small[4] = n;
        return;
    }
    // inside the tile: rank of every tie = before + ties in front of it (k-mer ids ascending = item order u * 1024 + thread)
    const uint64_t base = (uint64_t)tile_s * (1024 * VH_ITEMS);
    unsigned long long seen = before_s;
    for (int u = 0; u < VH_ITEMS; u++) {
        const uint64_t i = base + (uint64_t)u * 1024 + threadIdx.x;
        const bool tie = i < n && merged[i] == T;
        // ties of this row in front of this thread
        const unsigned long long m = __ballot(tie);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __shared__ uint32_t rowc[16];
        if (lane == 0) rowc[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        unsigned long long front = seen;
        for (int w = 0; w < wave; w++) front += rowc[w];
        front += (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if (tie && front == keep) small[4] = i;
        unsigned long long row = 0;
        for (int w = 0; w < 16; w++) row += rowc[w];
        seen += row;
        __syncthreads();
    }
}

__global__ void values_cut_kernel2(const uint64_t* __restrict__ merged, uint64_t n, const uint64_t* __restrict__ small,
                                   double* __restrict__ values) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || small[3]) return;
    const uint64_t m = merged[i], T = small[1], X = small[4];
    if (m > T || (m == T && i >= X) || i == 0) values[i] = 0.0;
}

extern "C" int dp_kmer_values(dp_ctx* ctx, int k, double* values_out) {
    if (!ctx || k < 1 || k > 15) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_kmer_values: k in 1..15") : DP_ERR_ARG;
    if (ctx->borrowed_reads) return dp_fail(ctx, DP_ERR_STATE, "dp_kmer_values on a borrowing context");
    hipSetDevice(ctx->device);
    const uint64_t n = (uint64_t)1 << (2 * k);
    void *d_counts = nullptr, *d_merged = nullptr, *d_sorted = nullptr, *d_rank = nullptr, *d_tmp = nullptr, *d_small = nullptr;
    auto cleanup = [&] {
        for (void* p : {d_counts, d_merged, d_sorted, d_rank, d_tmp, d_small})
            if (p) dp_dev_free(p);
    };
#define DPV(x)                                                      \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            cleanup();                                              \
            return dp_fail(ctx, DP_ERR_HIP, "dp_kmer_values: " #x, e_); \
        }                                                           \
    } while (0)
    if (dev_reserve(ctx, ctx->d_values, n * sizeof(double))) return DP_ERR_HIP;
    ctx->n_values = 0;
    ctx->values_total = 0;
    ctx->values_computed = false;
    double* values = (double*)ctx->d_values.p;
    DPV(dp_dev_malloc(&d_merged, n * 8));
    DPV(dp_dev_malloc(&d_small, 64));
    if (ctx->d_kcounts && ctx->kcounts_k == k) {
        // the k-mer position index was built first (dp_scan_prepare): its last pass left the histogram of exactly these k-mers
        d_counts = ctx->d_kcounts;
        ctx->d_kcounts = nullptr;
        ctx->kcounts_k = 0;
    } else {
        DPV(dp_dev_malloc(&d_counts, n * 4));
        DPV(hipMemsetAsync(d_counts, 0, n * 4, ctx->stream));
        int rc = dp_histogram_device(ctx, k, (uint32_t*)d_counts);
        if (rc != 0) {
            cleanup();
            return rc;
        }
    }
    // tot (overlap.go:60-63)
    size_t tb = 0;
    auto in64 = rocprim::make_transform_iterator((const uint32_t*)d_counts, ValuesToU64());
    DPV(rocprim::reduce(nullptr, tb, in64, (uint64_t*)d_small, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), ctx->stream));
    size_t tmpCap = tb + 64;
    DPV(dp_dev_malloc(&d_tmp, tmpCap));
    DPV(rocprim::reduce(d_tmp, tb, in64, (uint64_t*)d_small, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), ctx->stream));
    uint64_t tot = 0;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    const uint64_t topN = n / 100;
    bool cut_done = false;
    static const bool sort_cut = false;
    if (topN > 0 && !sort_cut) {
        // value pass + histogram, threshold, tie count, tie locate, cut: five launches, nothing read back in between
        const uint32_t tiles = (uint32_t)((n + 1024 * VH_ITEMS - 1) / (1024 * VH_ITEMS));
        void* d_h = nullptr;
        DPV(dp_dev_malloc(&d_h, (size_t)(VH_BINS + 1) * 4 + (size_t)tiles * 4 + 64));
        d_rank = d_h;  // (released by cleanup())
        uint32_t* ghist = (uint32_t*)d_h;
        uint32_t* tileTies = ghist + VH_BINS + 1;
        DPV(hipMemsetAsync(d_h, 0, (size_t)(VH_BINS + 1) * 4, ctx->stream));
        DPV(hipMemsetAsync((uint64_t*)d_small + 1, 0, 56, ctx->stream));
        if (k >= 8) {  // (below k = 8 the table is smaller than a tile: the reverse complement's count is gathered from memory)
            const uint32_t n_tiles = 1u << (2 * (k - 6));
            hipLaunchKernelGGL(values_kernel_hist_tiled, dim3((n_tiles + VT_PER - 1) / VT_PER), dim3(256), 0, ctx->stream, (const uint32_t*)d_counts, k,
                               (const uint64_t*)d_small, values, (uint64_t*)d_merged, ghist);
        } else {
            hipLaunchKernelGGL(values_kernel_hist, dim3(tiles), dim3(1024), 0, ctx->stream, (const uint32_t*)d_counts, n, k, (const uint64_t*)d_small,
                               values, (uint64_t*)d_merged, ghist);
        }
        hipLaunchKernelGGL(values_threshold_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)ghist, n, topN, (uint64_t*)d_small);
        hipLaunchKernelGGL(values_tie_count_kernel, dim3(tiles), dim3(1024), 0, ctx->stream, (const uint64_t*)d_merged, n, (const uint64_t*)d_small,
                           tileTies);
        hipLaunchKernelGGL(values_tie_locate_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint64_t*)d_merged, n, tiles,
                           (const uint32_t*)tileTies, (uint64_t*)d_small);
        hipLaunchKernelGGL(values_cut_kernel2, dim3(blocks), dim3(256), 0, ctx->stream, (const uint64_t*)d_merged, n, (const uint64_t*)d_small, values);
        DPV(hipGetLastError());
        uint64_t h_small[4] = {0, 0, 0, 0};
        DPV(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, ctx->stream));
        DPV(dp_stream_sync(ctx));
        tot = h_small[0];
        cut_done = h_small[3] == 0;  // (else: T beyond the histogram - the sort below; values[] and merged[] are as the value pass left them)
        dp_dev_free(d_h);
        d_rank = nullptr;
    } else {
        DPV(hipMemcpyAsync(&tot, d_small, 8, hipMemcpyDeviceToHost, ctx->stream));
        DPV(dp_stream_sync(ctx));
        hipLaunchKernelGGL(values_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_counts, n, k, (double)tot, values,
                           (uint64_t*)d_merged);
        DPV(hipGetLastError());
    }
    if (topN > 0 && !cut_done) {
        DPV(dp_dev_malloc(&d_sorted, n * 8));
        size_t sb = 0;
        DPV(rocprim::radix_sort_keys(nullptr, sb, (const uint64_t*)d_merged, (uint64_t*)d_sorted, (size_t)n, 0u, 34u, ctx->stream));
        if (sb + 64 > tmpCap) {
            dp_dev_free(d_tmp);
            d_tmp = nullptr;
            tmpCap = sb + 64;
            DPV(dp_dev_malloc(&d_tmp, tmpCap));
        }
        DPV(rocprim::radix_sort_keys(d_tmp, sb, (const uint64_t*)d_merged, (uint64_t*)d_sorted, (size_t)n, 0u, 34u, ctx->stream));
        uint64_t T = 0;
        DPV(hipMemcpyAsync(&T, (const uint64_t*)d_sorted + (n - topN), 8, hipMemcpyDeviceToHost, ctx->stream));
        DPV(dp_stream_sync(ctx));
        hipLaunchKernelGGL(values_bounds_kernel, dim3(1), dim3(64), 0, ctx->stream, (const uint64_t*)d_sorted, n, T, (uint64_t*)d_small);
        uint64_t bounds[2] = {0, 0};
        DPV(hipMemcpyAsync(bounds, d_small, 16, hipMemcpyDeviceToHost, ctx->stream));
        DPV(dp_stream_sync(ctx));
        dp_dev_free(d_sorted);
        d_sorted = nullptr;
        const uint64_t totalTies = bounds[1] - bounds[0], above = n - bounds[1];
        const uint64_t zeroTies = topN - above;                       // >= 1: sorted[n - topN] == T
        const uint32_t keepTies = (uint32_t)(totalTies - zeroTies);   // ties with a rank (from index 0) below this keep their value
        DPV(dp_dev_malloc(&d_rank, n * 4));
        auto flags = rocprim::make_transform_iterator((const uint64_t*)d_merged, ValuesIsTie{T});
        size_t xb = 0;
        DPV(rocprim::exclusive_scan(nullptr, xb, flags, (uint32_t*)d_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
        if (xb + 64 > tmpCap) {
            dp_dev_free(d_tmp);
            d_tmp = nullptr;
            tmpCap = xb + 64;
            DPV(dp_dev_malloc(&d_tmp, tmpCap));
        }
        DPV(rocprim::exclusive_scan(d_tmp, xb, flags, (uint32_t*)d_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->stream));
        hipLaunchKernelGGL(values_cut_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint64_t*)d_merged,
                           (const uint32_t*)d_rank, n, T, keepTies, values);
        DPV(hipGetLastError());
    } else if (topN == 0) {
        hipLaunchKernelGGL(values_zero_first, dim3(1), dim3(1), 0, ctx->stream, values);
    }
    if (values_out) DPV(hipMemcpyAsync(values_out, values, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DPV(dp_stream_sync(ctx));
    // the histogram is what the k-mer position index build counts first (dp_kindex_ensure): keep it (4^k * 4 bytes) until
    // the index has used it or the reads are replaced
    if (ctx->d_kcounts) dp_dev_free(ctx->d_kcounts);
    ctx->d_kcounts = d_counts;
    ctx->kcounts_k = k;
    d_counts = nullptr;
    cleanup();
    ctx->n_values = n;
    ctx->values_total = tot;
    ctx->values_computed = true;
#undef DPV
    return DP_OK;
}

// The resident value table (dp_kmer_values / dp_values_upload) copied to the host on the calling context's stream: lets a
// caller overlap the 4^k * 8 byte download with other set-up work on another context (dp_scan_prepare).
extern "C" int dp_values_download(dp_ctx* ctx, double* values_out, uint64_t n) {
    if (!ctx || !values_out) return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_values_download: bad arguments") : DP_ERR_ARG;
    const dp_ctx* src = ctx->owner ? ctx->owner : ctx;
    if (!src->d_values.p || src->n_values != n) return dp_fail(ctx, DP_ERR_STATE, "dp_values_download: no value table of this size resident");
    hipSetDevice(ctx->device);
    hipError_t e = hipMemcpyAsync(values_out, src->d_values.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = dp_stream_sync(ctx);
    if (e != hipSuccess) return dp_fail(ctx, DP_ERR_HIP, "dp_values_download", e);
    return DP_OK;
}

// code of a k-mer = its own count where the table holds a value, 0 where the value is 0 (count < 3, 1 % cut, k-mer 0)
__global__ void values_codes_kernel(const uint32_t* __restrict__ counts, const double* __restrict__ values, uint64_t n,
                                    uint16_t* __restrict__ codes, uint32_t* __restrict__ overflow) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = values[i] == 0.0 ? 0u : counts[i];
    if (c > 65535u) {
        *overflow = 1u;
        c = 65535u;
    }
    codes[i] = (uint16_t)c;
}

// codes of one byte (round 4): the 1 % cut takes every common k-mer's value away, so what is left nearly always counts below 255 -
// half the bytes over the link again (64 MB at k = 13).  *overflow = 1 when a valued k-mer counts 255 or more: the caller asks for
// the two-byte codes then.
__global__ void values_codes8_kernel(const uint32_t* __restrict__ counts, const double* __restrict__ values, uint64_t n,
                                     uint8_t* __restrict__ codes, uint32_t* __restrict__ overflow) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = values[i] == 0.0 ? 0u : counts[i];
    if (c > 254u) {
        *overflow = 1u;
        c = 255u;
    }
    codes[i] = (uint8_t)c;
}

static int values_download_codes_impl(dp_ctx* ctx, void* codes_out, uint64_t n, uint64_t* total_out, int* overflow_out, int bytes) {
    const dp_ctx* src = ctx->owner ? ctx->owner : ctx;
    if (!src->d_values.p || src->n_values != n || !src->d_kcounts || ((uint64_t)1 << (2 * src->kcounts_k)) != n || !src->values_computed)
        return dp_fail(ctx, DP_ERR_STATE, "dp_values_download_codes: no computed value table of this size resident");
    hipSetDevice(ctx->device);
    void* d_codes = nullptr;
    hipError_t e = dp_dev_malloc(&d_codes, n * bytes + 64);
    if (e != hipSuccess) return dp_fail(ctx, DP_ERR_HIP, "dp_values_download_codes: hipMalloc", e);
    uint32_t* d_flag = (uint32_t*)((char*)d_codes + n * bytes);
    uint32_t flag = 0;
    e = hipMemsetAsync(d_flag, 0, 4, ctx->stream);
    if (e == hipSuccess) {
        if (bytes == 1)
            hipLaunchKernelGGL(values_codes8_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)src->d_kcounts,
                               (const double*)src->d_values.p, n, (uint8_t*)d_codes, d_flag);
        else
            hipLaunchKernelGGL(values_codes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)src->d_kcounts,
                               (const double*)src->d_values.p, n, (uint16_t*)d_codes, d_flag);
        e = hipGetLastError();
    }
    // (the flag first: a caller of the one-byte form that has to fall back learns it without the codes' copy... which it gets anyway)
    if (e == hipSuccess) e = hipMemcpyAsync(codes_out, d_codes, n * bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = dp_stream_sync(ctx);
    dp_dev_free(d_codes);
    if (e != hipSuccess) return dp_fail(ctx, DP_ERR_HIP, "dp_values_download_codes", e);
    *total_out = src->values_total;
    *overflow_out = (int)flag;
    return DP_OK;
}

extern "C" int dp_values_download_codes8(dp_ctx* ctx, uint8_t* codes_out, uint64_t n, uint64_t* total_out, int* overflow_out) {
    if (!ctx || !codes_out || !total_out || !overflow_out)
        return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_values_download_codes8: bad arguments") : DP_ERR_ARG;
    return values_download_codes_impl(ctx, codes_out, n, total_out, overflow_out, 1);
}

// The value of a k-mer is a function of its own count and the total alone (overlap.go:73-88), so the table dp_kmer_values
// left resident travels as 2 bytes per k-mer: codes_out[i] = count of k-mer i, or 0 where its value is 0; *total_out = the
// sum of all counts (overlap.go:60-63).  A caller rebuilds value(i) = f(codes[i], total) bit for bit from 65536
// evaluations.  *overflow_out = 1 when a valued k-mer has a count above 65535 (use dp_values_download then).
// DP_ERR_STATE when the table was uploaded (dp_values_upload) rather than computed here: there is no histogram then.
extern "C" int dp_values_download_codes(dp_ctx* ctx, uint16_t* codes_out, uint64_t n, uint64_t* total_out, int* overflow_out) {
    if (!ctx || !codes_out || !total_out || !overflow_out)
        return ctx ? dp_fail(ctx, DP_ERR_ARG, "dp_values_download_codes: bad arguments") : DP_ERR_ARG;
    return values_download_codes_impl(ctx, codes_out, n, total_out, overflow_out, 2);
}
