// Scattered integer atomic adds (450 k bumps into 100 k counters, as the index walk's counting pass makes them) by scope:
// agent scope (what atomicAdd is) against workgroup scope on per-XCD counter copies (indexed by the hardware's XCC_ID; a line
// then never leaves its XCD's L2).  Also: returning forms, and plain stores as the floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
template <int MODE>
__global__ void bump(const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ cnt, uint32_t stride, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = idx[i];
    if (MODE == 0) atomicAdd(&cnt[t], 1u);
    if (MODE == 1) {
        const uint32_t x = __builtin_amdgcn_s_getreg(20 | (3 << 11));  // HW_REG_XCC_ID
        __hip_atomic_fetch_add(&cnt[(size_t)x * stride + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE == 2) {
        const uint32_t r = atomicAdd(&cnt[t], 1u);
        if (r == 0xffffffffu) sink[0] = r;
    }
    if (MODE == 3) {
        const uint32_t x = __builtin_amdgcn_s_getreg(20 | (3 << 11));
        const uint32_t r = __hip_atomic_fetch_add(&cnt[(size_t)x * stride + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (r == 0xffffffffu) sink[0] = r;
    }
    if (MODE == 4) cnt[t] = i;
    if (MODE == 5) {
        const uint32_t x = __builtin_amdgcn_s_getreg(20 | (3 << 11));
        __hip_atomic_fetch_add(&cnt[(size_t)x * stride + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const uint32_t n = 450000, m = 100000, stride = 100352;
    std::vector<uint32_t> h(n);
    srand(1);
    for (auto& v : h) v = (uint32_t)(((uint64_t)rand() * 2147483647ull + rand()) % m);
    uint32_t *d_idx, *d_cnt, *d_sink;
    hipMalloc(&d_idx, n * 4); hipMalloc(&d_cnt, (size_t)8 * stride * 4); hipMalloc(&d_sink, 64);
    hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"agent scope, one array", "workgroup scope, per-XCD copies", "agent scope, returning", "workgroup scope per-XCD, returning", "plain stores", "agent scope, per-XCD copies"};
    for (int mode = 0; mode < 6; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipMemset(d_cnt, 0, (size_t)8 * stride * 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const dim3 g((n + 255) / 256), b(256);
            switch (mode) {
                case 0: hipLaunchKernelGGL(bump<0>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
                case 1: hipLaunchKernelGGL(bump<1>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
                case 2: hipLaunchKernelGGL(bump<2>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
                case 3: hipLaunchKernelGGL(bump<3>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
                case 4: hipLaunchKernelGGL(bump<4>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
                case 5: hipLaunchKernelGGL(bump<5>, g, b, 0, 0, d_idx, n, d_cnt, stride, d_sink); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // check the sum
        std::vector<uint32_t> c((size_t)8 * stride);
        hipMemcpy(c.data(), d_cnt, c.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long sum = 0;
        for (auto v : c) sum += v;
        printf("%-38s %7.1f us   (sum of counters %llu%s)\n", names[mode], best * 1e3, sum, mode == 4 ? ": stores, not counts" : sum == n ? " = bumps" : " != bumps");
    }
    return 0;
}
