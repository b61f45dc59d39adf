// How many small operations per second does the HIP runtime issue from T host threads, one stream each?
// (kernel launches, 128-byte memsets, 128-byte pinned D2H copies; a stream sync every 8 operations)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void tiny(int* p) { if (threadIdx.x == 0 && p) p[blockIdx.x] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    for (int mode = 0; mode < 3; mode++) {
        for (int T : {1, 2, 4, 8, 12}) {
            std::vector<std::thread> th;
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            double t0 = 0;
            std::vector<double> done(T);
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    hipSetDevice(0);
                    hipStream_t s;
                    hipStreamCreate(&s);
                    int* d;
                    hipMalloc(&d, 4096);
                    int* h;
                    hipHostMalloc(&h, 4096);
                    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
                    hipStreamSynchronize(s);
                    ready++;
                    while (!go.load()) {}
                    for (int i = 0; i < iters; i++) {
                        if (mode == 0) hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, s, d);
                        else if (mode == 1) hipMemsetAsync(d, 0, 128, s);
                        else hipMemcpyAsync(h, d, 128, hipMemcpyDeviceToHost, s);
                        if ((i & 7) == 7) hipStreamSynchronize(s);
                    }
                    hipStreamSynchronize(s);
                    done[t] = now();
                    hipFree(d);
                    hipHostFree(h);
                    hipStreamDestroy(s);
                });
            while (ready.load() < T) {}
            t0 = now();
            go = true;
            for (auto& x : th) x.join();
            double t1 = 0;
            for (double d : done) t1 = d > t1 ? d : t1;
            printf("%s  threads %2d: %8.0f ops/s total, %6.2f us per op per thread\n", mode == 0 ? "kernel " : mode == 1 ? "memset " : "d2h copy", T,
                   (double)T * iters / (t1 - t0), 1e6 * (t1 - t0) / iters);
        }
    }
    return 0;
}
