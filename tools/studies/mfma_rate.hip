// Diagnostic micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 on gfx950 (not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u % NACC], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks, const char* name) {
    float* out; hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 128;   // 2048 MFMAs per wave
    mfma_loop<NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) mfma_loop<NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double mfma_per_simd = (double)iters * 16 * (threads / 64) / 4.0 * ((blocks + 255) / 256);
    const double tf = (double)iters * 16 * (threads / 64) * blocks * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-34s threads=%d blocks=%d: %.1f us, %.1f TF, %.1f ns per MFMA per SIMD (= %.1f cycles @2.4GHz)\n", name, threads,
           blocks, ms * 1e3, tf, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
    hipFree(out);
}
int main() {
    run<1>(256, 256, "1 acc, 1 wave/SIMD");
    run<2>(256, 256, "2 acc, 1 wave/SIMD");
    run<4>(256, 256, "4 acc, 1 wave/SIMD");
    run<1>(512, 256, "1 acc, 2 waves/SIMD");
    run<2>(512, 256, "2 acc, 2 waves/SIMD");
    run<1>(512, 248, "1 acc, 2 waves/SIMD, 248 blocks");
    run<4>(256, 1024, "4 acc, 1 wave/SIMD x4 blocks/CU");
    return 0;
}
