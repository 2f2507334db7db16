// Diagnostic: duration of (nearly) empty kernels with the launch geometry of rowgemm / ffn (not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_k(float* out, int n) {
    extern __shared__ float sm[];
    if (n < 0) { sm[threadIdx.x] = 1.f; out[0] = sm[0]; }
}
__global__ void touch_k(float* out, const float* in, int rows) {   // read 32 KB + write 32 KB per block
    extern __shared__ float sm[];
    const int r0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 256; i += blockDim.x) sm[i] = in[(size_t)r0 * 256 + i];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 256; i += blockDim.x) out[(size_t)r0 * 256 + i] = sm[i] + 1.f;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 50 * 1e3f;
}
int main() {
    float *a, *b; hipMalloc(&a, 7936 * 256 * 4); hipMalloc(&b, 7936 * 256 * 4);
    hipFuncSetAttribute((const void*)empty_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipFuncSetAttribute((const void*)touch_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int lds : {0, 32768, 110000, 157000})
        for (int thr : {256, 512})
            printf("empty  blocks=248 threads=%d lds=%6d : %.2f us per launch (back-to-back, incl. gaps)\n", thr, lds,
                   timeit([&] { empty_k<<<248, thr, lds>>>(a, 0); }));
    printf("touch  blocks=248 threads=512 lds=110000 : %.2f us\n", timeit([&] { touch_k<<<248, 512, 110000>>>(a, b, 7936); }));
    printf("touch  blocks=248 threads=256 lds=40000  : %.2f us\n", timeit([&] { touch_k<<<248, 256, 40000>>>(a, b, 7936); }));
    return 0;
}
