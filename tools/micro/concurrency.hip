// How many kernels of ONE process run at the same time?  S streams, each a chain of kernels that do nothing but sleep `us`
// microseconds on `blocks` workgroups; concurrency = S * n * us / wall time.  (HISTORY.md 5.7)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void sleeper(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

int main(int argc, char** argv) {
    const int us = argc > 1 ? atoi(argv[1]) : 50, blocks = argc > 2 ? atoi(argv[2]) : 1, n = 400;
    for (int S : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        std::vector<hipStream_t> st(S);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) hipLaunchKernelGGL(sleeper, dim3(blocks), dim3(64), 0, s, 100ull);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++)
            for (auto& s : st) hipLaunchKernelGGL(sleeper, dim3(blocks), dim3(64), 0, s, (unsigned long long)us * 100ull);
        hipDeviceSynchronize();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d streams x %d kernels of %d us on %d workgroups: %.1f ms -> %.2f kernels at once, %.1f us per kernel and stream\n", S, n, us, blocks, el * 1e3,
               (double)S * n * us * 1e-6 / el, el * 1e6 / n);
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
