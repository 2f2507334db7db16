// HBM-bound kernels of the Conformer path: LayerNorm, CMVN+conv1, depthwise-conv+LN+SiLU,
// CTC softmax/argmax and CTC collapse.  All are written for wave64 and 16-byte coalesced access.
#include <algorithm>

#include "common.h"

namespace masr {

__device__ __forceinline__ float wave_sum(float v) {
    return wave_sum_dpp(v);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// LayerNorm(256), eps inside sqrt (torch.nn.LayerNorm; reference conformer/encoder.py:63-72).
// One wave per row, float4 per lane, two-pass variance in registers.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* y, int M, float eps,
                                                           int seq_t, int pad, const int* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    size_t orow = row;
    bool zero = false;
    if (seq_t > 0) {
        const int bb = row / seq_t, t = row - bb * seq_t;
        orow = (size_t)bb * (seq_t + pad) + pad + t;
        if (lens && 4 * t >= lens[bb]) zero = true;
    }
    f32x4 o;
    if (zero) {
        o = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + eps);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
        o[0] = d0 * rstd * ww[0] + bb[0];
        o[1] = d1 * rstd * ww[1] + bb[1];
        o[2] = d2 * rstd * ww[2] + bb[2];
        o[3] = d3 * rstd * ww[3] + bb[3];
    }
    *reinterpret_cast<f32x4*>(y + orow * 256 + lane * 4) = o;
}

void launch_layernorm(const float* x, const float* w, const float* b, float* y, int M, float eps, int seq_t, int pad,
                      const int* lens, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(layernorm256_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w, b, y, M, eps, seq_t, pad, lens);
}

// y[b][pad + t] = scale * x[b][t] + bias  (Squeezeformer adaptive scale / bias in front of the conv module,
// convolution.py:109-110) written into the padded streaming layout [nseq][pad + seq_t][256] behind the cnn cache rows
__global__ __launch_bounds__(256) void affine_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* y, int M, int seq_t, int pad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int bb = row / seq_t, t = row - bb * seq_t;
    const size_t orow = (size_t)bb * (seq_t + pad) + pad + t;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + lane * 4);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ww[i] * v[i] + bv[i];
    *reinterpret_cast<f32x4*>(y + orow * 256 + lane * 4) = o;
}
void launch_affine_rows(const float* x, const float* w, const float* b, float* y, int M, int seq_t, int pad, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(affine_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, w, b, y, M, seq_t, pad);
}

// ------------------------------------------------------------------------------------------
// GlobalCMVN (utils/cmvn.py:21-32) + Conv2d(1->256, 3x3, stride 2) + ReLU
// (conformer/subsampling.py:86-87).  One workgroup per (b, t1) output row, thread = out channel;
// the 3 x F input rows are normalised once into LDS and broadcast-read.
// Output is channels-last [B, T1, F1, C] so that the conv2 implicit GEMM reads contiguous K.
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ feats, const float* __restrict__ mean,
                                                    const float* __restrict__ istd, const float* __restrict__ w9c,
                                                    const float* __restrict__ bias, float* __restrict__ out, int T,
                                                    int F, int T1, int F1, int C) {
    extern __shared__ float sm[];  // [3][F]
    const int bt = blockIdx.x;
    const int b = bt / T1, t1 = bt % T1;
    for (int i = threadIdx.x; i < 3 * F; i += blockDim.x) {
        const int kh = i / F, f = i % F;
        const float v = feats[((size_t)b * T + 2 * t1 + kh) * F + f];
        sm[i] = (v - mean[f]) * istd[f];
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= C) return;
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = w9c[i * C + c];
    const float bv = bias[c];
    float* o = out + ((size_t)bt * F1) * C + c;
    for (int f1 = 0; f1 < F1; ++f1) {
        float acc = bv;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = fmaf(w[kh * 3 + kw], sm[kh * F + 2 * f1 + kw], acc);
        if (NT) __builtin_nontemporal_store(fmaxf(acc, 0.f), &o[(size_t)f1 * C]);
        else o[(size_t)f1 * C] = fmaxf(acc, 0.f);
    }
}

// the 636 MB of conv1 output (B = 32 x 10 s) are written once and read back from HBM by conv2 whatever the caches do: streaming
// (non-temporal) stores, 135.3 -> 125.6 us in one kernel trace with both forms alternating, conv2 behind it unchanged
static int g_conv1_nt = 1;
void set_conv1_nt(int on) { g_conv1_nt = on; }

void launch_conv1(const float* feats, const float* mean, const float* istd, const float* w9c, const float* bias,
                  float* out, int B, int T, int F, int C, hipStream_t s) {
    const int T1 = (T - 1) / 2, F1 = (F - 1) / 2;
    if (B * T1 <= 0) return;
    if (g_conv1_nt)
        hipLaunchKernelGGL(conv1_kernel<1>, dim3(B * T1), dim3(256), 3 * F * sizeof(float), s, feats, mean, istd, w9c, bias,
                           out, T, F, T1, F1, C);
    else
        hipLaunchKernelGGL(conv1_kernel<0>, dim3(B * T1), dim3(256), 3 * F * sizeof(float), s, feats, mean, istd, w9c, bias,
                           out, T, F, T1, F1, C);
}

// ------------------------------------------------------------------------------------------
// Depthwise causal Conv1d(k taps, groups=C) + LayerNorm(C) + SiLU
// (conformer/convolution.py:121-126).  Input is the GLU output in the padded layout
// [nseq, pad + Tq, 256] with pad = k-1 history rows in front of every sequence (zero rows or the
// streaming cnn cache pushed through pointwise_conv1+GLU, exactly as the reference does by
// concatenating before pointwise_conv1, convolution.py:101-108).  Output [nseq*Tq, 256].
// Workgroup = 16 output rows of one sequence; thread = channel for the sliding-window conv,
// then the tile is transposed through LDS so that each wave normalises whole rows.
// ------------------------------------------------------------------------------------------
static constexpr int DW_TT = 16;
// NORM = 0: LayerNorm(C) (Conformer);  NORM = 1: eval-mode BatchNorm1d folded into scale/shift passed in
// lnw / lnb (Squeezeformer, convolution.py:62-67,137-141).  The padded layout is the same in both cases:
// KT - 1 extra rows per sequence (all in front for the causal conv, (KT-1)/2 on each side for the symmetric one).
template <int KT, int NORM>
__global__ __launch_bounds__(256) void dwconv_ln_silu_kernel(const float* __restrict__ g, const float* __restrict__ wkc,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, float* __restrict__ out,
                                                             int Tq, float eps, const float* __restrict__ gconst) {
    __shared__ __align__(16) float tile[DW_TT][256 + 4];
    const int tiles = (Tq + DW_TT - 1) / DW_TT;
    const int seq = blockIdx.x / tiles;
    const int t0 = (blockIdx.x % tiles) * DW_TT;
    const int c = threadIdx.x;
    constexpr int pad = KT - 1;
    // padded row (t0 + j) of this sequence <-> input time t0 + j - pad
    const float* gin = g + ((size_t)seq * (pad + Tq) + t0) * 256 + c;
    float w[KT], win[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) w[j] = wkc[j * 256 + c];
    const float bv = bias[c];
    const int nrows = min(DW_TT, Tq - t0);
    win[0] = 0.f;
    // gconst != nullptr (offline causal conv): the KT-1 history rows of every sequence are not materialised -- they all
    // equal glu(pointwise_conv1.bias), the image of the reference's zero left-padding (convolution.py:101-104,118-119)
    const float gc = gconst ? gconst[c] : 0.f;
#pragma unroll
    for (int j = 0; j < KT - 1; ++j) win[j + 1] = (gconst && t0 + j < pad) ? gc : gin[(size_t)j * 256];
    for (int r = 0; r < nrows; ++r) {
#pragma unroll
        for (int j = 0; j < KT - 1; ++j) win[j] = win[j + 1];
        win[KT - 1] = gin[(size_t)(pad + r) * 256];
        float acc = bv;                       // out[t] = b + sum_j w[j] * gpad[t + j]
#pragma unroll
        for (int j = 0; j < KT; ++j) acc = fmaf(w[j], win[j], acc);
        tile[r][c] = acc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 ww = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
    for (int r = wave; r < nrows; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[r][lane * 4]);
        f32x4 o;
        if (NORM == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = v[i] * ww[i] + bb[i];
        } else {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            o[0] = d0 * rstd * ww[0] + bb[0];
            o[1] = d1 * rstd * ww[1] + bb[1];
            o[2] = d2 * rstd * ww[2] + bb[2];
            o[3] = d3 * rstd * ww[3] + bb[3];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = o[i] / (1.0f + expf(-o[i]));
        *reinterpret_cast<f32x4*>(out + ((size_t)seq * Tq + t0 + r) * 256 + lane * 4) = o;
    }
}

void launch_dwconv_ln_silu(const float* g, const float* wkc, const float* bias, const float* lnw, const float* lnb,
                           float* out, int nseq, int Tq, int ktaps, float eps, hipStream_t s, const float* gconst) {
    if (nseq * Tq <= 0) return;
    const int tiles = (Tq + DW_TT - 1) / DW_TT;
    const dim3 grid(nseq * tiles), blk(256);
    if (ktaps == 15) hipLaunchKernelGGL((dwconv_ln_silu_kernel<15, 0>), grid, blk, 0, s, g, wkc, bias, lnw, lnb, out, Tq, eps, gconst);
    else if (ktaps == 7) hipLaunchKernelGGL((dwconv_ln_silu_kernel<7, 0>), grid, blk, 0, s, g, wkc, bias, lnw, lnb, out, Tq, eps, gconst);
    else if (ktaps == 31) hipLaunchKernelGGL((dwconv_ln_silu_kernel<31, 0>), grid, blk, 0, s, g, wkc, bias, lnw, lnb, out, Tq, eps, gconst);
}

// glu(bias) of pointwise_conv1 with the arithmetic of the rowgemm GLU epilogue: the constant history rows of the offline
// causal conv module
__global__ void glu_const_kernel(const float* __restrict__ bias, float* __restrict__ out) {
    const int c = threadIdx.x;
    out[c] = bias[c] * __builtin_amdgcn_rcpf(1.0f + __expf(-bias[256 + c]));
}
void launch_glu_const(const float* bias512, float* out256, hipStream_t s) {
    hipLaunchKernelGGL(glu_const_kernel, dim3(1), dim3(256), 0, s, bias512, out256);
}

void launch_dwconv_bn_silu(const float* g, const float* wkc, const float* bias, const float* scale, const float* shift,
                           float* out, int nseq, int Tq, int ktaps, hipStream_t s, const float* gconst) {
    if (nseq * Tq <= 0) return;
    const int tiles = (Tq + DW_TT - 1) / DW_TT;
    const dim3 grid(nseq * tiles), blk(256);
    if (ktaps == 31) hipLaunchKernelGGL((dwconv_ln_silu_kernel<31, 1>), grid, blk, 0, s, g, wkc, bias, scale, shift, out, Tq, 0.f, gconst);
    else if (ktaps == 15) hipLaunchKernelGGL((dwconv_ln_silu_kernel<15, 1>), grid, blk, 0, s, g, wkc, bias, scale, shift, out, Tq, 0.f, gconst);
}

// ------------------------------------------------------------------------------------------
// Efficient-Conformer StrideConformerEncoderLayer pieces (efficient_conformer/encoder.py:454-545,
// convolution.py:43-49): causal depthwise conv with stride 2 + LayerNorm + SiLU, and the residual path
// AvgPool1d(kernel 2, stride 2, ceil_mode=True, count_include_pad=False).
//   out[j] = b + sum_k w[k] * gpad[2j + k],  j < ceil(Tin / 2)   (gpad = KT-1 history rows + Tin rows)
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void dwconv_stride2_ln_silu_kernel(const float* __restrict__ g,
                                                                     const float* __restrict__ wkc,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ lnw,
                                                                     const float* __restrict__ lnb,
                                                                     float* __restrict__ out, int Tin, int Tout,
                                                                     float eps) {
    __shared__ __align__(16) float tile[DW_TT][256 + 4];
    const int tiles = (Tout + DW_TT - 1) / DW_TT;
    const int seq = blockIdx.x / tiles;
    const int t0 = (blockIdx.x % tiles) * DW_TT;
    const int c = threadIdx.x;
    const float* gin = g + ((size_t)seq * (KT - 1 + Tin)) * 256 + c;
    float w[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) w[j] = wkc[j * 256 + c];
    const float bv = bias[c];
    const int nrows = min(DW_TT, Tout - t0);
    for (int r = 0; r < nrows; ++r) {
        float acc = bv;
        const float* gp = gin + (size_t)(2 * (t0 + r)) * 256;
#pragma unroll
        for (int j = 0; j < KT; ++j) acc = fmaf(w[j], gp[(size_t)j * 256], acc);
        tile[r][c] = acc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 ww = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
    for (int r = wave; r < nrows; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[r][lane * 4]);
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
        const float rstd = 1.0f / sqrtf(var + eps);
        f32x4 o;
        o[0] = d0 * rstd * ww[0] + bb[0];
        o[1] = d1 * rstd * ww[1] + bb[1];
        o[2] = d2 * rstd * ww[2] + bb[2];
        o[3] = d3 * rstd * ww[3] + bb[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = o[i] / (1.0f + expf(-o[i]));
        *reinterpret_cast<f32x4*>(out + ((size_t)seq * Tout + t0 + r) * 256 + lane * 4) = o;
    }
}

void launch_dwconv_stride2_ln_silu(const float* g, const float* wkc, const float* bias, const float* lnw,
                                   const float* lnb, float* out, int nseq, int Tin, int ktaps, float eps, hipStream_t s) {
    const int Tout = (Tin + 1) / 2;
    if (nseq * Tout <= 0 || ktaps != 15) return;
    const int tiles = (Tout + DW_TT - 1) / DW_TT;
    hipLaunchKernelGGL(dwconv_stride2_ln_silu_kernel<15>, dim3(nseq * tiles), dim3(256), 0, s, g, wkc, bias, lnw, lnb, out,
                       Tin, Tout, eps);
}

__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ out, int T,
                                                       int Tout) {
    const int b = blockIdx.y, j = blockIdx.x, c = threadIdx.x;
    const float a = x[((size_t)b * T + 2 * j) * 256 + c];
    float v = a;
    if (2 * j + 1 < T) v = (a + x[((size_t)b * T + 2 * j + 1) * 256 + c]) / 2.0f;
    out[((size_t)b * Tout + j) * 256 + c] = v;
}

void launch_avgpool2(const float* x, float* out, int B, int T, hipStream_t s) {
    const int Tout = (T + 1) / 2;
    if (B * Tout <= 0) return;
    hipLaunchKernelGGL(avgpool2_kernel, dim3(Tout, B), dim3(256), 0, s, x, out, T, Tout);
}

// ------------------------------------------------------------------------------------------
// Squeezeformer TimeReductionLayer1D, depthwise part (time_reduction.py:53-66): pad-masked input,
// Conv1d(k=5, stride=2, padding=3, groups=C); only the first L = ceil(T/2) outputs are kept (:68-74).
// out[b][j][c] = bias[c] + sum_k w[k][c] * xm[b][2j + k - 3][c]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_reduce_dw_kernel(const float* __restrict__ x, const float* __restrict__ w5c,
                                                             const float* __restrict__ bias,
                                                             const int* __restrict__ lens, float* __restrict__ out,
                                                             int T, int L, int mstride) {
    const int b = blockIdx.y, j = blockIdx.x, c = threadIdx.x;
    const int len = lens ? lens[b] : 0x7fffffff;
    float acc = bias[c];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int t = 2 * j + k - 3;
        if (t >= 0 && t < T && (long)mstride * t < len) acc = fmaf(w5c[k * 256 + c], x[((size_t)b * T + t) * 256 + c], acc);
    }
    out[((size_t)b * L + j) * 256 + c] = acc;
}

void launch_time_reduce_dw(const float* x, const float* w5c, const float* bias, const int* lens, float* out, int B, int T,
                           int mstride, hipStream_t s) {
    const int L = (T + 1) / 2;
    if (B * L <= 0) return;
    hipLaunchKernelGGL(time_reduce_dw_kernel, dim3(L, B), dim3(256), 0, s, x, w5c, bias, lens, out, T, L, mstride);
}

// Squeezeformer recovery (encoder.py:199-205): x = recover_tensor + Linear(repeat_interleave(x, 2))[:, :T]
// (Linear commutes with the repeat: y = Linear(x_reduced) is computed once per reduced frame)
__global__ __launch_bounds__(256) void recover_add_kernel(const float* __restrict__ saved, const float* __restrict__ y,
                                                          float* __restrict__ x, int T, int L) {
    const int b = blockIdx.y, t = blockIdx.x, c = threadIdx.x;
    x[((size_t)b * T + t) * 256 + c] = saved[((size_t)b * T + t) * 256 + c] + y[((size_t)b * L + (t >> 1)) * 256 + c];
}

void launch_recover_add(const float* saved, const float* y, float* x, int B, int T, int L, hipStream_t s) {
    if (B * T <= 0) return;
    hipLaunchKernelGGL(recover_add_kernel, dim3(T, B), dim3(256), 0, s, saved, y, x, T, L);
}

// ------------------------------------------------------------------------------------------
// CTC head tail: softmax over V (loss/ctc.py:62-70) + per-frame argmax / max prob
// (ctc_greedy_decoder.py:20-21).  One workgroup per frame; the row lives in registers.
// argmax is taken on the logits with first-index tie breaking (numpy argmax semantics).
// ------------------------------------------------------------------------------------------
// per-thread row slice of the vocabulary kernels below (256 threads): 32 values cover V <= 8192 (the shipped vocabularies), 64
// cover V <= 16384 (the template argument; launch_* pick by V)
template <int SM_MAXPT>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(float* logits, int V, int ldv, int write_probs,
                                                             int* __restrict__ idx, float* __restrict__ maxp) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    __shared__ float red_s[4];
    const int row = blockIdx.x;
    float* x = logits + (size_t)row * ldv;
    float v[SM_MAXPT];
    float m = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SM_MAXPT; ++i) {
        const int j = threadIdx.x + i * 256;
        v[i] = j < V ? x[j] : -INFINITY;
        if (v[i] > m) { m = v[i]; mi = j; }
    }
    // block argmax (value desc, index asc)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red_v[wave] = m; red_i[wave] = mi; }
    __syncthreads();
    m = red_v[0]; mi = red_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (red_v[w] > m || (red_v[w] == m && red_i[w] < mi)) { m = red_v[w]; mi = red_i[w]; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXPT; ++i) {
        v[i] = expf(v[i] - m);   // exp(-inf) = 0 for the tail
        sum += v[i];
    }
    sum = wave_sum(sum);
    if (lane == 0) red_s[wave] = sum;
    __syncthreads();
    sum = red_s[0] + red_s[1] + red_s[2] + red_s[3];
    const float inv = 1.0f / sum;
    if (write_probs) {
#pragma unroll
        for (int i = 0; i < SM_MAXPT; ++i) {
            const int j = threadIdx.x + i * 256;
            if (j < V) x[j] = v[i] / sum;
        }
    }
    if (threadIdx.x == 0) {
        if (idx) idx[row] = mi;
        if (maxp) maxp[row] = inv;   // exp(0) / sum
    }
}

void launch_softmax_argmax(float* logits, int M, int V, int ldv, int write_probs, int* idx, float* maxp,
                           hipStream_t s) {
    if (M <= 0) return;
    if (V <= 8192) hipLaunchKernelGGL(softmax_argmax_kernel<32>, dim3(M), dim3(256), 0, s, logits, V, ldv, write_probs, idx, maxp);
    else hipLaunchKernelGGL(softmax_argmax_kernel<64>, dim3(M), dim3(256), 0, s, logits, V, ldv, write_probs, idx, maxp);
}

// ------------------------------------------------------------------------------------------
// CTC best-path collapse (ctc_greedy_decoder.py:20-30): drop repeats, drop blanks; the score is
// the SEQUENTIAL fp32 sum (python sum over np.float32) of the max-probs of all non-blank frames,
// divided by their count.  T' <= ~1250, one lane per utterance is plenty.
// ------------------------------------------------------------------------------------------
// One wave per utterance: 64 frames per step; a frame emits a token when it is non-blank and differs from
// its predecessor (ballot + prefix popcount gives the output slot); the score stays a strictly sequential
// fp32 accumulation (lane 0 walks the non-blank max-probs staged in LDS, in frame order).
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int* __restrict__ idx, const float* __restrict__ maxp,
                                                          const int* __restrict__ nframes, int Tp, int blank,
                                                          int* tokens, int* ntok, float* score, int ldt, int ld_in,
                                                          const int* __restrict__ in_rows) {
    extern __shared__ float nbp[];          // [Tp] max-probs of the non-blank frames, compacted in frame order
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = nframes ? min(nframes[b], Tp) : Tp;
    // in_rows (serving pool): utterance b reads row in_rows[b] of history matrices whose rows are ld_in frames apart
    const size_t src = (size_t)(in_rows ? in_rows[b] : b) * ld_in;
    const int* ib = idx + src;
    const float* pb = maxp + src;
    int* tb = tokens + (size_t)b * ldt;     // ldt = Tp, or Tp + 2 for packed rows [tokens | count | score bits] (ntok == nullptr)
    int cnt = 0, nb = 0, carry = -1;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < n;
        const int id = in ? ib[t] : blank;
        int prev = __shfl_up(id, 1, 64);
        if (lane == 0) prev = carry;
        const bool nonblank = in && id != blank;
        const bool emit = nonblank && id != prev;
        const unsigned long long me = __ballot(emit), mn = __ballot(nonblank);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (emit) tb[cnt + __popcll(me & below)] = id;
        if (nonblank) nbp[nb + __popcll(mn & below)] = pb[t];
        cnt += __popcll(me);
        nb += __popcll(mn);
        carry = __shfl(id, 63, 64);
    }
    for (int t = cnt + lane; t < Tp; t += 64) tb[t] = -1;
    __syncthreads();
    if (lane == 0) {
        float acc = 0.f;
        for (int i = 0; i < nb; ++i) acc = acc + nbp[i];
        const float sc = nb > 0 ? acc / (float)nb : 0.f;
        if (ntok) {
            ntok[b] = cnt;
            score[b] = sc;
        } else {
            tb[Tp] = cnt;
            tb[Tp + 1] = __float_as_int(sc);
        }
    }
}

void launch_ctc_collapse(const int* idx, const float* maxp, const int* nframes, int B, int Tp, int blank, int* tokens,
                         int* ntok, float* score, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), (size_t)Tp * sizeof(float), s, idx, maxp, nframes, Tp,
                       blank, tokens, ntok, score, Tp, Tp, (const int*)nullptr);
}

// the same with ONE packed int32 row per utterance: rows [B, Tp + 2] = tokens (-1 padded) | token count | score bits
void launch_ctc_collapse_rows(const int* idx, const float* maxp, const int* nframes, int B, int Tp, int blank, int* rows,
                              hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), (size_t)Tp * sizeof(float), s, idx, maxp, nframes, Tp,
                       blank, rows, (int*)nullptr, (float*)nullptr, Tp + 2, Tp, (const int*)nullptr);
}

// ... and with the (argmax, max prob) frames read from rows in_rows[b] of history matrices [*, ld_in] (masr_pool_step): Tp = the
// longest history of the call, rows [B, Tp + 2]
void launch_ctc_collapse_hist(const int* idx, const float* maxp, const int* nframes, const int* in_rows, int ld_in, int B, int Tp,
                              int blank, int* rows, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), (size_t)Tp * sizeof(float), s, idx, maxp, nframes, Tp,
                       blank, rows, (int*)nullptr, (float*)nullptr, Tp + 2, ld_in, in_rows);
}

// Segment copies of the serving pool (masr_pool_step): segment g moves seg[g].n rows of `width` floats from src row seg[g].src to
// dst row seg[g].dst (feature frames into / out of the sessions' rows of the frame pool; width 1: (argmax, max prob) frames into the
// histories, both arrays in one launch)
__global__ __launch_bounds__(256) void copy_segments_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            const float* __restrict__ src2, float* __restrict__ dst2,
                                                            const PoolSeg* __restrict__ seg, int width) {
    const PoolSeg g = seg[blockIdx.x];
    const long total = (long)g.n * width;
    const float* a = src + (size_t)g.src * width;
    float* b = dst + (size_t)g.dst * width;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)gridDim.y * 256) {
        b[i] = a[i];
        if (src2) dst2[(size_t)g.dst * width + i] = src2[(size_t)g.src * width + i];
    }
}
void launch_copy_segments(const float* src, float* dst, const float* src2, float* dst2, const PoolSeg* seg, int nseg,
                          int max_rows, int width, hipStream_t s) {
    if (nseg <= 0 || max_rows <= 0) return;
    const long total = (long)max_rows * width;
    const int gy = (int)std::min<long>(64, (total + 255) / 256);
    hipLaunchKernelGGL(copy_segments_kernel, dim3(nseg, gy), dim3(256), 0, s, src, dst, src2, dst2, seg, width);
}

// argmax / max over rows of an existing probability matrix (np.argmax semantics: first maximum)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ p, int V, int* __restrict__ idx,
                                                          float* __restrict__ maxp) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const float* x = p + (size_t)blockIdx.x * V;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int j = threadIdx.x; j < V; j += 256) {
        const float v = x[j];
        if (v > m) { m = v; mi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { red_v[threadIdx.x >> 6] = m; red_i[threadIdx.x >> 6] = mi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = red_v[0]; mi = red_i[0];
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > m || (red_v[w] == m && red_i[w] < mi)) { m = red_v[w]; mi = red_i[w]; }
        idx[blockIdx.x] = mi;
        if (maxp) maxp[blockIdx.x] = m;
    }
}

void launch_argmax_rows(const float* probs, int M, int V, int* idx, float* maxp, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(256), 0, s, probs, V, idx, maxp);
}

// samples -> fbank frames (snip_edges) -> encoder frames (two 3x3/stride-2 convs)
__global__ void frame_counts_kernel(const int* __restrict__ nsamp, int B, int* nfr, int* nenc, int halve) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = nsamp[b];
    const int T = n >= 400 ? 1 + (n - 400) / 160 : 0;
    if (nfr) nfr[b] = T;
    if (nenc) {
        const int n4 = T >= 7 ? ((T - 1) / 2 - 1) / 2 : 0;
        nenc[b] = halve ? (n4 + 1) / 2 : n4;
    }
}

void launch_frame_counts(const int* nsamp, int B, int* nfr, int* nenc, int halve, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(frame_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, nsamp, B, nfr, nenc, halve);
}

// stream caches -> reference layouts (encoder.py:404-419): att [L,H,t,2dk], cnn [L,1,d,pad]
// rate = 2: the layer keeps its cache at half the frame rate; the reference layout repeats every entry (encoder.py:347)
__global__ void export_att_kernel(const float* __restrict__ cache, float* __restrict__ out, int H, int cap, int t,
                                  int dk, int rate, int shift) {
    const int l = blockIdx.z, h = blockIdx.y, j = blockIdx.x;
    const int d = H * dk;
    const float* row = cache + ((size_t)l * cap + (j + shift) / rate) * 2 * d;
    float* o = out + (((size_t)l * H + h) * t + j) * 2 * dk;
    for (int i = threadIdx.x; i < 2 * dk; i += blockDim.x) o[i] = i < dk ? row[h * dk + i] : row[d + h * dk + (i - dk)];
}
void launch_export_att(const float* cache, float* out, int L, int H, int cap, int t, int dk, hipStream_t s, int rate,
                       int shift) {
    hipLaunchKernelGGL(export_att_kernel, dim3(t, H, L), dim3(128), 0, s, cache, out, H, cap, t, dk, rate, shift);
}
__global__ void export_cnn_kernel(const float* __restrict__ cache, float* __restrict__ out, int pad, int d) {
    const int l = blockIdx.x;
    for (int i = threadIdx.x; i < pad * d; i += blockDim.x) {
        const int c = i / pad, j = i % pad;
        out[(size_t)l * pad * d + i] = cache[((size_t)l * pad + j) * d + c];
    }
}
// one layer whose cache has `used` rows, exported right-aligned into a [d][total] block (zeros in front, encoder.py:372)
__global__ void export_cnn_layer_kernel(const float* __restrict__ cache, float* __restrict__ out, int used, int total, int d) {
    for (int i = threadIdx.x; i < used * d; i += blockDim.x) {
        const int c = i / used, j = i % used;
        out[(size_t)c * total + (total - used) + j] = cache[(size_t)j * d + c];
    }
}
void launch_export_cnn_layer(const float* cache, float* out, int used, int total, int d, hipStream_t s) {
    hipLaunchKernelGGL(export_cnn_layer_kernel, dim3(1), dim3(256), 0, s, cache, out, used, total, d);
}
void launch_export_cnn(const float* cache, float* out, int L, int pad, int d, hipStream_t s) {
    hipLaunchKernelGGL(export_cnn_kernel, dim3(L), dim3(256), 0, s, cache, out, pad, d);
}

// wave-wide max / min with the result in every lane: an inclusive DPP scan (row shifts + row broadcasts, lanes without a source keep
// their own value) whose last lane holds the reduction
__device__ __forceinline__ float wave_max_f32_dpp(float v) {
#define MASR_DPP_MAXF(ctrl, rows) \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), (ctrl), (rows), 0xf, false)))
    MASR_DPP_MAXF(0x111, 0xf); MASR_DPP_MAXF(0x112, 0xf); MASR_DPP_MAXF(0x114, 0xf); MASR_DPP_MAXF(0x118, 0xf);
    MASR_DPP_MAXF(0x142, 0xa); MASR_DPP_MAXF(0x143, 0xc);
#undef MASR_DPP_MAXF
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i32_dpp(int v) {
#define MASR_DPP_MINI(ctrl, rows) v = min(v, __builtin_amdgcn_update_dpp(v, v, (ctrl), (rows), 0xf, false))
    MASR_DPP_MINI(0x111, 0xf); MASR_DPP_MINI(0x112, 0xf); MASR_DPP_MINI(0x114, 0xf); MASR_DPP_MINI(0x118, 0xf);
    MASR_DPP_MINI(0x142, 0xa); MASR_DPP_MINI(0x143, 0xc);
#undef MASR_DPP_MINI
    return __builtin_amdgcn_readlane(v, 63);
}

// ------------------------------------------------------------------------------------------
// Vocabulary pruning for the CTC prefix beam search (get_pruned_log_probs of the third-party
// ctc_beam_search_decoder: sort descending, keep the shortest prefix whose cumulative probability reaches
// cutoff_prob, at most top_n entries; log(p + FLT_MIN)).  One workgroup per frame, top_n rounds of a
// block-wide arg-max over the register-resident row (ties -> smaller index).
// ------------------------------------------------------------------------------------------
template <int SM_MAXPT>
__global__ __launch_bounds__(256) void topk_prune_kernel(const float* __restrict__ probs, int V, int top_n,
                                                         float cutoff_prob, int* __restrict__ out_idx,
                                                         float* __restrict__ out_logp, int* __restrict__ out_cnt, int blank,
                                                         float* __restrict__ out_blank_lp) {
    __shared__ float red_v[4];
    __shared__ int red_i[4];
    const float* x = probs + (size_t)blockIdx.x * V;
    // ln p(blank) of the frame as the decoder's min_cutoff rule takes it (std::log(prob[blank_id]), no FLT_MIN added)
    if (out_blank_lp && threadIdx.x == 0) out_blank_lp[blockIdx.x] = logf(x[blank]);
    float v[SM_MAXPT];
#pragma unroll
    for (int i = 0; i < SM_MAXPT; ++i) {
        const int j = threadIdx.x + i * 256;
        v[i] = j < V ? x[j] : -1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double cum = 0.0;
    int cnt = 0;
    for (int k = 0; k < top_n; ++k) {
        float m = -1.f;
        int mi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SM_MAXPT; ++i) {
            const int j = threadIdx.x + i * 256;
            if (v[i] > m) { m = v[i]; mi = j; }
        }
        {   // wave arg-max (value desc, index asc) on DPP row shifts / broadcasts: twelve VALU instructions where the shuffle form was
            // twelve ds_bpermute round trips per round -- this kernel sits between a pass's CTC head and its prefix search
            const float wm = wave_max_f32_dpp(m);
            mi = wave_min_i32_dpp(m == wm ? mi : 0x7fffffff);
            m = wm;
        }
        __syncthreads();
        if (lane == 0) { red_v[wave] = m; red_i[wave] = mi; }
        __syncthreads();
        m = red_v[0]; mi = red_i[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (red_v[w] > m || (red_v[w] == m && red_i[w] < mi)) { m = red_v[w]; mi = red_i[w]; }
        if (m < 0.f) break;                              // vocabulary exhausted (V < top_n)
        if (threadIdx.x == 0) {
            out_idx[(size_t)blockIdx.x * top_n + k] = mi;
            out_logp[(size_t)blockIdx.x * top_n + k] = logf(m + 1.17549435e-38f);
        }
        // remove the winner from its owner's registers
#pragma unroll
        for (int i = 0; i < SM_MAXPT; ++i)
            if (threadIdx.x + i * 256 == mi) v[i] = -1.f;
        cum += (double)m;
        ++cnt;
        if (cutoff_prob < 1.0f && cum >= (double)cutoff_prob) break;
    }
    if (threadIdx.x == 0) out_cnt[blockIdx.x] = cnt;
}

void launch_topk_prune(const float* probs, int M, int V, int top_n, float cutoff_prob, int* out_idx, float* out_logp,
                       int* out_cnt, int blank, float* out_blank_lp, hipStream_t s) {
    if (M <= 0) return;
    if (V <= 8192)
        hipLaunchKernelGGL(topk_prune_kernel<32>, dim3(M), dim3(256), 0, s, probs, V, top_n, cutoff_prob, out_idx, out_logp, out_cnt,
                           blank, out_blank_lp);
    else
        hipLaunchKernelGGL(topk_prune_kernel<64>, dim3(M), dim3(256), 0, s, probs, V, top_n, cutoff_prob, out_idx, out_logp, out_cnt,
                           blank, out_blank_lp);
}

// ------------------------------------------------------------------------------------------
// Streaming cache maintenance for n lock-step streams in ONE launch each (encoder.py:404-419 does
// torch.concat / slicing per stream): append this chunk's k|v rows to every stream's KV cache, and
// move the conv-module history rows between the per-stream cnn caches and the padded LN buffer.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void kv_append_kernel(const AttSeq* __restrict__ seqs, const float* __restrict__ qkv,
                                                        int Tq) {
    const int i = blockIdx.y, r = blockIdx.x;
    const AttSeq sq = seqs[i];
    // cache row = [k(256) | v(256)], the new rows go to positions nk - nq .. nk - 1
    float* dst = const_cast<float*>(sq.k) + (size_t)(sq.nk - sq.nq + r) * 512;
    const float* src = qkv + ((size_t)i * Tq + r) * 768 + 256;
    reinterpret_cast<f32x4*>(dst)[threadIdx.x] = reinterpret_cast<const f32x4*>(src)[threadIdx.x];
}
void launch_kv_append(const AttSeq* seqs, const float* qkv, int n, int Tq, hipStream_t s) {
    if (n * Tq <= 0) return;
    hipLaunchKernelGGL(kv_append_kernel, dim3(Tq, n), dim3(128), 0, s, seqs, qkv, Tq);
}

// Efficient-Conformer grouped layers keep PLANAR key / value caches ([cap][256] each, zero behind the last row) so that the
// flat [T,256] -> [T/3,4,192] regrouping of cache + chunk (attention.py:147-155 before pad4group) is a plain reinterpretation
__global__ __launch_bounds__(64) void kv_append_planar_kernel(const PlaneCopy* __restrict__ pc, int Tq) {
    const PlaneCopy c = pc[blockIdx.y];
    const int r = blockIdx.x;
    reinterpret_cast<f32x4*>(c.dst_k + (size_t)r * 256)[threadIdx.x] = reinterpret_cast<const f32x4*>(c.src_k + (size_t)r * 256)[threadIdx.x];
    reinterpret_cast<f32x4*>(c.dst_v + (size_t)r * 256)[threadIdx.x] = reinterpret_cast<const f32x4*>(c.src_v + (size_t)r * 256)[threadIdx.x];
}
void launch_kv_append_planar(const PlaneCopy* pc, int n, int Tq, hipStream_t s) {
    if (n * Tq <= 0) return;
    hipLaunchKernelGGL(kv_append_planar_kernel, dim3(Tq, n), dim3(64), 0, s, pc, Tq);
}

// stream cache (planar K | V planes) -> reference layout [H, t, 2dk] of one layer
__global__ void export_att_planar_kernel(const float* __restrict__ kplane, const float* __restrict__ vplane,
                                         float* __restrict__ out, int H, int t, int dk) {
    const int h = blockIdx.y, j = blockIdx.x;
    const int d = H * dk;
    float* o = out + ((size_t)h * t + j) * 2 * dk;
    for (int i = threadIdx.x; i < 2 * dk; i += blockDim.x)
        o[i] = i < dk ? kplane[(size_t)j * d + h * dk + i] : vplane[(size_t)j * d + h * dk + (i - dk)];
}
void launch_export_att_planar(const float* kplane, const float* vplane, float* out, int H, int t, int dk, hipStream_t s) {
    if (t <= 0) return;
    hipLaunchKernelGGL(export_att_planar_kernel, dim3(t, H), dim3(128), 0, s, kplane, vplane, out, H, t, dk);
}

// dir 0: lnpad[i][0..pad) <- cache_i ;  dir 1: cache_i <- lnpad[i][Tq .. Tq+pad)
__global__ __launch_bounds__(64) void cnn_cache_move_kernel(float* const* __restrict__ caches, float* lnpad, int Tq,
                                                            int pad, int dir) {
    const int i = blockIdx.y, r = blockIdx.x;
    float* c = caches[i] + (size_t)r * 256;
    float* l = lnpad + ((size_t)i * (Tq + pad) + (dir ? Tq + r : r)) * 256;
    if (dir) reinterpret_cast<f32x4*>(c)[threadIdx.x] = reinterpret_cast<const f32x4*>(l)[threadIdx.x];
    else reinterpret_cast<f32x4*>(l)[threadIdx.x] = reinterpret_cast<const f32x4*>(c)[threadIdx.x];
}
void launch_cnn_cache_move(float* const* caches, float* lnpad, int n, int Tq, int pad, int dir, hipStream_t s) {
    if (n * pad <= 0) return;
    hipLaunchKernelGGL(cnn_cache_move_kernel, dim3(pad, n), dim3(64), 0, s, caches, lnpad, Tq, pad, dir);
}

// Streaming conv module front, one launch instead of three (cache -> history rows | LayerNorm of the new rows | new cache):
//   v(i, tp) = tp < pad ? cache_rd[i][tp] : f(x[i][tp - pad])          f = LayerNorm (Conformer) or scale * x + bias (Squeezeformer)
//   lnpad[i][tp] = v(i, tp)   for tp in [0, pad + Tq)
//   cache_wr[i][r] = v(i, Tq + r)   for r in [0, pad)                  (convolution.py:100-108: new_cache = x[:, :, -lorder:])
// cache_rd / cache_wr are the two halves of a double-buffered cache (the caller flips them per call), so rows are independent:
// one wave per padded row, no ordering between them.
template <int AFFINE>
__global__ __launch_bounds__(256) void conv_hist_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* const* __restrict__ cache_rd,
                                                        float* const* __restrict__ cache_wr, float* __restrict__ lnpad,
                                                        int n, int Tq, int pad, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = Tq + pad;
    if (row >= n * per) return;
    const int i = row / per, tp = row - i * per;
    f32x4 o;
    if (tp < pad) {
        o = *reinterpret_cast<const f32x4*>(cache_rd[i] + (size_t)tp * 256 + lane * 4);
    } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)i * Tq + tp - pad) * 256 + lane * 4);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + lane * 4);
        if (AFFINE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ww[k] * v[k] + bb[k];
        } else {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            o[0] = d0 * rstd * ww[0] + bb[0];
            o[1] = d1 * rstd * ww[1] + bb[1];
            o[2] = d2 * rstd * ww[2] + bb[2];
            o[3] = d3 * rstd * ww[3] + bb[3];
        }
    }
    *reinterpret_cast<f32x4*>(lnpad + (size_t)row * 256 + lane * 4) = o;
    if (tp >= Tq) *reinterpret_cast<f32x4*>(cache_wr[i] + (size_t)(tp - Tq) * 256 + lane * 4) = o;
}
void launch_conv_hist(const float* x, const float* w, const float* b, float* const* cache_rd, float* const* cache_wr,
                      float* lnpad, int n, int Tq, int pad, int affine, float eps, hipStream_t s) {
    const int rows = n * (Tq + pad);
    if (rows <= 0) return;
    if (affine)
        hipLaunchKernelGGL(conv_hist_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, cache_rd, cache_wr, lnpad, n, Tq,
                           pad, eps);
    else
        hipLaunchKernelGGL(conv_hist_kernel<0>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, cache_rd, cache_wr, lnpad, n, Tq,
                           pad, eps);
}

// ---- probes of engine.hip's side-stream selection: a kernel that holds its queue for `ticks` of the 100 MHz wall clock, and one
// that does nothing (which of two streams share a hardware queue: the empty kernel on one waits for the spin on the other)
__global__ void queue_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void queue_nop_kernel() {}
void launch_queue_spin(long long ticks, hipStream_t s) { hipLaunchKernelGGL(queue_spin_kernel, dim3(1), dim3(1), 0, s, ticks); }
void launch_queue_nop(hipStream_t s) { hipLaunchKernelGGL(queue_nop_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace masr
