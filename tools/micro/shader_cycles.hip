// Does hwreg 29 (SHADER_CYCLES on RDNA) count on gfx950?  Compared with s_memrealtime (100 MHz) and s_memtime around a spin.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* o) {
    const unsigned c0 = __builtin_amdgcn_s_getreg(29 | (19 << 11));
    const unsigned long long r0 = wall_clock64(), m0 = clock64();
    unsigned x = threadIdx.x;
    for (int i = 0; i < 10000; i++) x = x * 1664525u + 1013904223u;
    const unsigned c1 = __builtin_amdgcn_s_getreg(29 | (19 << 11));
    const unsigned long long r1 = wall_clock64(), m1 = clock64();
    // cost of the reads themselves
    const unsigned long long a0 = clock64();
    unsigned s = 0;
    for (int i = 0; i < 100; i++) s += __builtin_amdgcn_s_getreg(29 | (19 << 11));
    const unsigned long long a1 = clock64();
    unsigned long long t = 0;
    for (int i = 0; i < 100; i++) t += wall_clock64();
    const unsigned long long a2 = clock64();
    if (threadIdx.x == 0) {
        o[0] = c0, o[1] = c1, o[2] = r1 - r0, o[3] = m1 - m0, o[4] = x, o[5] = a1 - a0, o[6] = a2 - a1, o[7] = s + t;
    }
}
int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("hwreg29 %llu -> %llu (diff %lld), realtime ticks %llu (= %.1f us), clock64 %llu | 100 getreg: %llu clocks, 100 realtime reads: %llu clocks\n", h[0], h[1],
           (long long)h[1] - (long long)h[0], h[2], h[2] / 100.0, h[3], h[5], h[6]);
    return 0;
}
