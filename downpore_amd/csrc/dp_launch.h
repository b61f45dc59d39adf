// Launch of a per-round kernel.
//
// A per-round kernel is written as a struct with its block size and a static device function,
//     struct foo_k { enum { THREADS = 256 }; static __device__ void run(const int* a, uint32_t n) { ... } };
// and is launched through dp_launch<foo_k>(ctx, grid, block, args...): the one __global__ entry point, dp_kernel<foo_k>, takes the
// arguments as one block (converted to the parameter types of foo_k::run on the host) and calls foo_k::run with them.
// (Rounds 3 - 5 launched several contexts' rounds in one launch through this block - "gangs", selected by blockIdx.y; measured slower
// three rounds running - profiles/r03 .. r05 - and removed in round 6, HISTORY.md.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

struct dp_ctx;

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

static inline void dp_pack_fill(DpPack<>&) {}
template <class H, class... T, class A0, class... A>
static inline void dp_pack_fill(DpPack<H, T...>& p, const A0& a0, const A&... a) {
    p.h = (H)a0;
    dp_pack_fill(p.t, a...);
}

hipStream_t dp_ctx_stream(const dp_ctx* ctx);

#ifdef __HIPCC__
template <class K, class... Done>
__device__ __forceinline__ void dp_unpack_call(const DpPack<>&, const Done&... d) {
    K::run(d...);
}
template <class K, class H, class... T, class... Done>
__device__ __forceinline__ void dp_unpack_call(const DpPack<H, T...>& p, const Done&... d) {
    dp_unpack_call<K>(p.t, d..., p.h);
}
template <class K>
__global__ __launch_bounds__(K::THREADS) void dp_kernel(const dp_pack_of<K> a) {
    dp_unpack_call<K>(a);
}
template <class K, class... A>
static inline void dp_launch(dp_ctx* ctx, dim3 grid, dim3 block, const A&... args) {
    typedef dp_pack_of<K> Pack;
    static_assert(std::is_trivially_copyable<Pack>::value, "kernel arguments must be trivially copyable");
    Pack p;
    memset((void*)&p, 0, sizeof p);
    dp_pack_fill(p, args...);
    hipLaunchKernelGGL((dp_kernel<K>), grid, block, 0, dp_ctx_stream(ctx), p);
}
#endif
