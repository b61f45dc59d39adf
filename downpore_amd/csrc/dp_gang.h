// Gangs: the same launch carries the rounds of several contexts (DESIGN.md "rounds per launch").
//
// A round of `downpore overlap` (commands/overlap.go:119-195) is a chain of ~26 small, dependent, latency-bound kernels;
// consecutive rounds are independent until a read is flagged.  A gang is a set of contexts - one host thread each, exactly as
// without a gang - whose per-round kernels are issued TOGETHER: when every member has reached its next launch, one of them issues
// ONE launch on the gang's stream whose blockIdx.y selects the member (its own argument block: its own buffers, sizes, cursors).
// Nothing else changes for a member: its C-ABI calls, its buffers, its results.  A kernel that ran 334 waves for one round runs
// 4 x 334 for four rounds in the same time; the dispatches, barriers between dependent kernels and host waits of a round are
// shared by the gang.
//
// Device side: a per-round kernel is written as a struct with the block size and a static device function,
//     struct foo_k { enum { THREADS = 256 }; static __device__ void run(const int* a, uint32_t n) { ... } };
// and is launched through dp_launch<foo_k>(ctx, grid, block, args...), never through hipLaunchKernelGGL: the one __global__
// entry point, dp_multi<foo_k, N>, takes N argument blocks and calls foo_k::run with block blockIdx.y's.  Kernels must bound
// blockIdx.x by their own sizes (the grid is the largest member's) and may use gridDim.x for grid-stride loops only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

struct dp_ctx;

// ---- argument blocks -------------------------------------------------------------------------------------------------------
template <class... T>
struct DpPack;
template <>
struct DpPack<> {};
template <class H, class... T>
struct DpPack<H, T...> {
    H h;
    DpPack<T...> t;
};
template <class F>
struct DpKernelArgs;
template <class... A>
struct DpKernelArgs<void (*)(A...)> {
    typedef DpPack<typename std::remove_cv<typename std::remove_reference<A>::type>::type...> pack;
};
template <class K>
using dp_pack_of = typename DpKernelArgs<decltype(&K::run)>::pack;

#ifdef __HIPCC__
template <class K, class... Done>
__device__ __forceinline__ void dp_unpack_call(const DpPack<>&, const Done&... d) {
    K::run(d...);
}
template <class K, class H, class... T, class... Done>
__device__ __forceinline__ void dp_unpack_call(const DpPack<H, T...>& p, const Done&... d) {
    dp_unpack_call<K>(p.t, d..., p.h);
}
template <class Pack, int N>
struct DpMultiArgs {
    Pack a[N];
};
template <class K, int N>
__global__ __launch_bounds__(K::THREADS) void dp_multi(const DpMultiArgs<dp_pack_of<K>, N> m) {
    dp_unpack_call<K>(m.a[blockIdx.y]);
}
#endif

static inline void dp_pack_fill(DpPack<>&) {}
template <class H, class... T, class A0, class... A>
static inline void dp_pack_fill(DpPack<H, T...>& p, const A0& a0, const A&... a) {
    p.h = (H)a0;
    dp_pack_fill(p.t, a...);
}

// ---- the gang -----------------------------------------------------------------------------------------------------------------
#define DP_GANG_MAX 8
#define DP_GANG_ARG_BYTES 512  // largest argument block of a per-round kernel (checked at compile time per kernel)

struct DpDeposit {
    // issues the launches of the members in `who` (all deposited with this function and block size) on `stream`
    void (*issue)(hipStream_t stream, const DpDeposit* const* who, int n);
    uint32_t grid_x, block_x;
    alignas(16) unsigned char args[DP_GANG_ARG_BYTES];
};

struct dp_gang;
// member states (dp_gang.cpp part of dp_scan.hip)
enum { DPG_OFF = 0, DPG_PREP = 1, DPG_START = 2, DPG_RUN = 3, DPG_LAUNCH = 4, DPG_SYNC = 5 };
// true: the calling context is a gang member inside a round (its launches are deposited)
bool dp_gang_active(const dp_ctx* ctx);
// a member reports an error while inside a round: the gang fails for good (dp_gang.hip: dp_gang::failed)
void dp_gang_mark_failed(dp_ctx* ctx);
// the member's next launch: returns once the launch (merged with the other members') has been issued on the gang's stream
void dp_gang_deposit(dp_ctx* ctx, const DpDeposit& d);
// the member wants to wait for its work: returns when every member inside a round has arrived at a wait as well (the caller
// then waits for the gang's stream as it would for its own)
void dp_gang_sync_point(dp_ctx* ctx);
hipStream_t dp_ctx_stream(const dp_ctx* ctx);
// around a section in which the member may block on another member (see dp_gang.hip); pause returns false outside a round
bool dp_gang_pause(dp_ctx* ctx);
// dp_ctx_destroy of a context that is still a member: the gang drops it (its slot stays OFF; dp_gang_destroy skips it)
void dp_gang_forget(dp_ctx* ctx);
void dp_gang_resume(dp_ctx* ctx);
struct DpGangPause {
    dp_ctx* ctx;
    bool paused;
    explicit DpGangPause(dp_ctx* c) : ctx(c), paused(dp_gang_pause(c)) {}
    ~DpGangPause() {
        if (paused) dp_gang_resume(ctx);
    }
};

#ifdef __HIPCC__
template <class K>
static void dp_issue(hipStream_t stream, const DpDeposit* const* who, int n) {
    typedef dp_pack_of<K> Pack;
    static_assert(sizeof(Pack) <= DP_GANG_ARG_BYTES, "argument block of a per-round kernel larger than DP_GANG_ARG_BYTES");
    static_assert(std::is_trivially_copyable<Pack>::value, "kernel arguments must be trivially copyable");
    int i = 0;
    while (i < n) {
        const int m = n - i;
        uint32_t gx = 0;
        if (m == 1) {
            DpMultiArgs<Pack, 1> a;
            memcpy(&a.a[0], who[i]->args, sizeof(Pack));
            hipLaunchKernelGGL((dp_multi<K, 1>), dim3(who[i]->grid_x), dim3(who[i]->block_x), 0, stream, a);
            i += 1;
        } else if (m <= 4 || sizeof(Pack) * 8 > 3800) {
            const int c = m < 4 ? m : 4;
            DpMultiArgs<Pack, 4> a;
            for (int j = 0; j < c; j++) {
                memcpy(&a.a[j], who[i + j]->args, sizeof(Pack));
                gx = who[i + j]->grid_x > gx ? who[i + j]->grid_x : gx;
            }
            for (int j = c; j < 4; j++) memcpy(&a.a[j], who[i]->args, sizeof(Pack));
            hipLaunchKernelGGL((dp_multi<K, 4>), dim3(gx, (uint32_t)c), dim3(who[i]->block_x), 0, stream, a);
            i += c;
        } else {
            const int c = m < 8 ? m : 8;
            DpMultiArgs<Pack, 8> a;
            for (int j = 0; j < c; j++) {
                memcpy(&a.a[j], who[i + j]->args, sizeof(Pack));
                gx = who[i + j]->grid_x > gx ? who[i + j]->grid_x : gx;
            }
            for (int j = c; j < 8; j++) memcpy(&a.a[j], who[i]->args, sizeof(Pack));
            hipLaunchKernelGGL((dp_multi<K, 8>), dim3(gx, (uint32_t)c), dim3(who[i]->block_x), 0, stream, a);
            i += c;
        }
    }
}

// Launch of a per-round kernel on the context's stream - or, for a gang member inside a round, together with the same launch
// of the other members.  Arguments are converted to the parameter types of K::run.
template <class K, class... A>
static inline void dp_launch(dp_ctx* ctx, dim3 grid, dim3 block, const A&... args) {
    typedef dp_pack_of<K> Pack;
    DpDeposit d;
    d.issue = &dp_issue<K>;
    d.grid_x = grid.x;
    d.block_x = block.x;
    Pack p;
    memset((void*)&p, 0, sizeof p);
    dp_pack_fill(p, args...);
    memcpy(d.args, &p, sizeof p);
    if (dp_gang_active(ctx)) {
        dp_gang_deposit(ctx, d);
    } else {
        const DpDeposit* one = &d;
        dp_issue<K>(dp_ctx_stream(ctx), &one, 1);
    }
}
#endif
