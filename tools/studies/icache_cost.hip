// Diagnostic (not part of the library): what does straight-line code size cost a short kernel?
// Same instruction count in both forms -- a rolled loop (small code) vs fully unrolled (N * 16 * 8 bytes of v_fma) -- and a
// different kernel launched in between so that nothing stays warm.  32 workgroups x 512 threads, like a streaming projection.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N, bool UNROLL>
__global__ __launch_bounds__(512) void chain_k(float* out, float s) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    if (UNROLL) {
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], s, (float)(n + 1));
    } else {
#pragma unroll 1
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], s, (float)(n + 1));
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += v[i];
    if (t == 123.456f) out[threadIdx.x] = t;
}
__global__ void other_k(float* out) { if (out[0] == 77.f) out[1] = 1.f; }
template <class F> float timeit(F f, float* a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float tot = 0.f;
    for (int i = 0; i < 50; ++i) {
        other_k<<<256, 256>>>(a);
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
    }
    return tot / 50 * 1e3f;
}
#define RUN(N) \
    printf("N=%3d (%5d B unrolled): rolled %.2f us   unrolled %.2f us\n", N, N * 16 * 8, \
           timeit([&] { chain_k<N, false><<<32, 512>>>(a, 1.0001f); }, a), timeit([&] { chain_k<N, true><<<32, 512>>>(a, 1.0001f); }, a));
int main() {
    float* a; hipMalloc(&a, 1 << 20); hipMemset(a, 0, 1 << 20);
    RUN(8) RUN(16) RUN(32) RUN(64) RUN(128) RUN(256)
    return 0;
}
