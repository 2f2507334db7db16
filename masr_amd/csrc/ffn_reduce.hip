// Split-d_ff reduction of the fused feed-forward block and the block's launcher.
// The block itself is the producer/consumer kernel in ffn_pc.hip (reference conformer/positionwise.py:30-37,
// conformer/encoder.py:113-121,150-161); for few rows (streaming chunk steps) d_ff is split across workgroups into
// partial sums that the kernel below adds in a fixed order, optionally followed by the layer's next LayerNorm.
#include "common.h"

namespace masr {

static constexpr int FF_D = 256;

__device__ __forceinline__ float wsum64(float v) {
    return wave_sum_dpp(v);
}

// x <- x + scale * (sum_s partial[s] + b2), partials added in ascending s (deterministic).  One wave per row; POSTLN: the
// LayerNorm that follows the block in the layer (norm_final after the second macaron FFN, encoder.py:160-161; the post-norms of
// Squeezeformer) is applied to the finished row while it is still in registers: y <- LayerNorm(x_new) (y may alias x)
template <int POSTLN>
__global__ __launch_bounds__(256) void ffn_reduce_kernel(float* x, const float* __restrict__ partial,
                                                         const float* __restrict__ b2, int M, int nsplit, float scale,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float* y,
                                                         float eps, const float* xin) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;        // float4 index
    if (i >= (size_t)M * FF_D / 4) return;
    const size_t sstride = (size_t)M * FF_D / 4;
    const f32x4* pp = reinterpret_cast<const f32x4*>(partial) + i;
    f32x4 xv = reinterpret_cast<const f32x4*>(xin ? xin : x)[i];      // xin: the rows a head stage of the split launch updated
    const f32x4 bb = reinterpret_cast<const f32x4*>(b2)[i % (FF_D / 4)];
    f32x4 gw = f32x4{1.f, 1.f, 1.f, 1.f}, gb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (POSTLN) {
        const int lane = threadIdx.x & 63;
        gw = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
        gb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {          // eight loads in flight per round, added in ascending order
        f32x4 q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = pp[(size_t)(sp + j) * sstride];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += q[j][k];
    }
    for (; sp < nsplit; ++sp) {
        const f32x4 p = pp[(size_t)sp * sstride];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = xv[k] + scale * (acc[k] + bb[k]);
    if (POSTLN) {
        const float mean = wsum64(xv[0] + xv[1] + xv[2] + xv[3]) * (1.0f / 256.0f);
        const float d0 = xv[0] - mean, d1 = xv[1] - mean, d2 = xv[2] - mean, d3 = xv[3] - mean;
        const float var = wsum64(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + eps);
        f32x4 o;
        o[0] = d0 * rstd * gw[0] + gb[0];
        o[1] = d1 * rstd * gw[1] + gb[1];
        o[2] = d2 * rstd * gw[2] + gb[2];
        o[3] = d3 * rstd * gw[3] + gb[3];
        reinterpret_cast<f32x4*>(y)[i] = o;
    } else {
        reinterpret_cast<f32x4*>(x)[i] = xv;
    }
}

void launch_ffn_reduce(float* x, const float* partial, const float* b2, int M, int nsplit, float scale, hipStream_t s,
                       const FfnPostLn* post, const float* xin) {
    const dim3 grid((unsigned)(((size_t)M * FF_D / 4 + 255) / 256));
    if (post && post->y)
        hipLaunchKernelGGL(ffn_reduce_kernel<1>, grid, dim3(256), 0, s, x, partial, b2, M, nsplit, scale, post->lnw, post->lnb,
                           post->y, post->eps, xin);
    else
        hipLaunchKernelGGL(ffn_reduce_kernel<0>, grid, dim3(256), 0, s, x, partial, b2, M, nsplit, scale, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, 0.f, xin);
}
int launch_ffn_pc(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                  const float* b2, int M, int dff, float eps, float scale, int affine_prologue, float* partial, int nsplit,
                  hipStream_t s, int variant, const FfnPostLn* post, const FfnTail* tail, const FfnHead* head);

static int g_ffn_variant = 0;
void set_ffn_variant(int v) { g_ffn_variant = v; }   // masr_debug_set(1, v): 81 = the kernel without weight loads (MFMA-only floor)

int launch_ffn_fused(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                     const float* b2, int M, int dff, float eps, float scale, int affine_prologue, float* partial,
                     int nsplit, hipStream_t s, const FfnPostLn* post, const FfnTail* tail, const FfnHead* head, bool packed) {
    if (M <= 0) return 0;
    // packed: w1 / w2 are the fragment-ordered copies of launch_pack_ffn_pc (full, non-split launches only)
    return launch_ffn_pc(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, affine_prologue, partial, nsplit, s,
                         packed ? 2 : g_ffn_variant == 81 ? 1 : 0, post, tail, head);
}

}  // namespace masr
