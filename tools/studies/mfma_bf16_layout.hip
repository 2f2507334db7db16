// Diagnostic (not part of the library): operand layout of v_mfma_f32_32x32x16_bf16 on gfx950, and the accuracy of the
// split-bf16 product (a = hi + lo in bf16; a*w ~ hi*hi + hi*lo + lo*hi, fp32 accumulate) against double.
// Assumed layout: A lane l holds A[i = l & 31][k = 8 * (l >> 5) + 0..7], B lane l holds B[k = 8 * (l >> 5) + 0..7][j = l & 31],
// C/D as the f32 32x32 forms: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ inline unsigned short bf16_rn(float x) {      // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ void k(const float* A, const float* W, float* C, int K) {   // C[32][32] = A[32][K] * W[32][K]^T
    const int l = threadIdx.x, i = l & 31, kb = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        union { bf16x8 v; unsigned short s[8]; } ah, al, wh, wl;
        for (int j = 0; j < 8; ++j) {
            const float a = A[i * K + k0 + 8 * kb + j], w = W[i * K + k0 + 8 * kb + j];
            ah.s[j] = bf16_rn(a); al.s[j] = bf16_rn(a - bf16_f(ah.s[j]));
            wh.s[j] = bf16_rn(w); wl.s[j] = bf16_rn(w - bf16_f(wh.s[j]));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, wh.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wl.v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, wh.v, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}
int main() {
    const int K = 2304;
    float *A = (float*)malloc(32 * K * 4), *W = (float*)malloc(32 * K * 4), *C = (float*)malloc(32 * 32 * 4);
    srand(1);
    for (int i = 0; i < 32 * K; ++i) { A[i] = (rand() / (float)RAND_MAX) * 8.f - 1.f; W[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; }
    float *dA, *dW, *dC;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dW, 32 * K * 4); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W, 32 * K * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dW, dC, K);
    hipMemcpy(C, dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, maxerr32 = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0; float f32 = 0.f;
            for (int kk = 0; kk < K; ++kk) { ref += (double)A[i * K + kk] * W[j * K + kk]; f32 = fmaf(A[i * K + kk], W[j * K + kk], f32); }
            maxerr = fmax(maxerr, fabs(C[i * 32 + j] - ref)); maxref = fmax(maxref, fabs(ref)); maxerr32 = fmax(maxerr32, fabs(f32 - ref));
        }
    printf("K=%d: split-bf16 (3 products) max abs err %.3e, fp32 fma chain max abs err %.3e, max |ref| %.3e\n", K, maxerr, maxerr32, maxref);
    return maxerr < 1e-2 * maxref ? 0 : 1;
}
