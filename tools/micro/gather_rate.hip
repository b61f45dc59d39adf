// How many scattered memory requests per second does an MI355X serve?  (HISTORY.md 5.7: what the rounds' kernels compete for.)
// Threads draw pseudo-random addresses in a buffer of `gb` GiB and read 4 bytes (or add 1 atomically) at each; `ilp` independent
// requests are in flight per thread.  Prints giga-requests per second for reads, returning atomics and non-returning atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

template <int MODE, int ILP>
__global__ void gather(uint32_t* __restrict__ buf, uint64_t n_words, int iters, uint32_t* __restrict__ sink) {
    uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t a[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            a[u] = (s >> 20) % n_words;
        }
#pragma unroll
        for (int u = 0; u < ILP; u++) {
            if (MODE == 0) acc += buf[a[u]];
            else if (MODE == 1) acc += atomicAdd(&buf[a[u]], 1u);
            else if (MODE == 2) atomicAdd(&buf[a[u]], 1u);
            else acc += (uint32_t)a[u];  // (3: no memory traffic at all - the control)
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int ILP>
static double run(uint32_t* buf, uint64_t n_words, uint32_t* sink, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, buf, n_words, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, buf, n_words, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * iters * ILP / (ms * 1e-3) / 1e9;
}

#include <chrono>
// background mode: `gather_rate bg <mode 0 reads | 2 atomics> <blocks> <seconds>` keeps `blocks` workgroups gathering over 8 GiB for
// that long and prints the rate it got - run beside a job to see what the job's rounds lose to that much scattered traffic
static int background(int mode, int blocks, double seconds, double gib) {
    const uint64_t n_words = (uint64_t)(gib * (double)(1ull << 30)) / 4;
    uint32_t *buf, *sink;
    if (hipMalloc(&buf, n_words * 4) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, n_words * 4);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double reqs = 0;
    for (;;) {
        if (mode == 0) hipLaunchKernelGGL((gather<0, 4>), dim3(blocks), dim3(256), 0, 0, buf, n_words, 2000, sink);
        else if (mode == 2) hipLaunchKernelGGL((gather<2, 4>), dim3(blocks), dim3(256), 0, 0, buf, n_words, 2000, sink);
        else hipLaunchKernelGGL((gather<3, 4>), dim3(blocks), dim3(256), 0, 0, buf, n_words, 20000, sink);
        hipDeviceSynchronize();
        reqs += (double)blocks * 256 * 2000 * 4;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el >= seconds) {
            printf("background %s over %.3f GiB, %d blocks: %.2f G requests/s over %.1f s\n", mode == 0 ? "reads" : mode == 2 ? "atomics" : "arithmetic only", gib, blocks, reqs / el / 1e9, el);
            return 0;
        }
    }
}

int main(int argc, char** argv) {
    if (argc > 4 && !strcmp(argv[1], "bg")) return background(atoi(argv[2]), atoi(argv[3]), atof(argv[4]), argc > 5 ? atof(argv[5]) : 8.0);
    const double gb = argc > 1 ? atof(argv[1]) : 8.0;
    const uint64_t n_words = (uint64_t)(gb * (1ull << 30)) / 4;
    uint32_t *buf, *sink;
    if (hipMalloc(&buf, n_words * 4) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, n_words * 4);
    for (int blocks : {256, 1024, 4096, 16384}) {
        printf("%.0f GiB, %5d blocks x 256 threads: reads %6.2f (1 in flight) %6.2f (4) %6.2f (8) | returning atomics %6.2f (1) %6.2f (4) | atomics %6.2f (4)  G requests/s\n", gb, blocks,
               run<0, 1>(buf, n_words, sink, blocks, 64), run<0, 4>(buf, n_words, sink, blocks, 32), run<0, 8>(buf, n_words, sink, blocks, 16),
               run<1, 1>(buf, n_words, sink, blocks, 64), run<1, 4>(buf, n_words, sink, blocks, 32), run<2, 4>(buf, n_words, sink, blocks, 32));
    }
    return 0;
}
