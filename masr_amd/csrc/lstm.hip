// DeepSpeech2 recurrent stack pieces (reference masr/model_utils/deepspeech2/encoder.py:36-45,96-129):
// nn.LSTM(batch_first, 1 layer, uni- or bi-directional) over a packed sequence + LayerNorm.
//
// The input projection W_ih . x_t + b_ih + b_hh of ALL timesteps is one MFMA GEMM (gemm_f32.hip); what is
// left is the recurrence  gates_t = Gx_t + W_hh . h_{t-1}  (PyTorch gate order i, f, g, o),
// c_t = sigma(f) c_{t-1} + sigma(i) tanh(g),  h_t = sigma(o) tanh(c_t).  One launch per timestep, both
// directions in it (blockIdx.y): a wave owns ONE hidden unit (its 4 gate rows of W_hh stay in registers,
// 16 values per lane) and walks over the batch; 1024 units / 4 waves = 256 workgroups = one per CU.
// W_hh (16 MB per direction) is re-streamed from L2 / Infinity Cache every step -- the recurrence is
// HBM/L2-bound by construction at batch 1; a persistent cross-CU kernel is the next step.
// pack_padded_sequence semantics: a sequence advances only while t < len (the reverse direction therefore
// starts at its own last frame), padded outputs are zero.
#include "common.h"

namespace masr {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int H>
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                        const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                        float* __restrict__ c, float* __restrict__ out,
                                                        const int* __restrict__ lens, int B, int T, int step, int ndir) {
    constexpr int PL = H / 64;                      // W_hh values per lane and gate
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int j = blockIdx.x * 4 + wave;
    const int t = dir ? T - 1 - step : step;
    float w[4][PL];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wr = whh + ((size_t)dir * 4 * H + (size_t)g * H + j) * H + lane * PL;
#pragma unroll
        for (int k = 0; k < PL; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + k);
            w[g][k] = v[0]; w[g][k + 1] = v[1]; w[g][k + 2] = v[2]; w[g][k + 3] = v[3];
        }
    }
    // batch loop: the four gate pre-activations of (unit j, sequence b) end up in lane (b & 63); the gate epilogue then
    // runs once per 64 sequences with one sequence per lane, so its dependent global loads cost one latency, not B
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = min(64, B - b0);
        float mine[4] = {0.f, 0.f, 0.f, 0.f};
        for (int bb = 0; bb < nb; ++bb) {
            const float* hp = h_prev + ((size_t)dir * B + b0 + bb) * H + lane * PL;
            float hv[PL];
#pragma unroll
            for (int k = 0; k < PL; k += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(hp + k);
                hv[k] = v[0]; hv[k + 1] = v[1]; hv[k + 2] = v[2]; hv[k + 3] = v[3];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < PL; ++k) a = fmaf(w[g][k], hv[k], a);
                a = wave_sum_dpp(a);
                if (lane == bb) mine[g] = a;
            }
        }
        if (lane < nb) {
            const int b = b0 + lane;
            const size_t sidx = ((size_t)dir * B + b) * H + j;
            const bool active = !lens || t < lens[b];
            const float* gr = gx + ((size_t)b * T + t) * (ndir * 4 * H) + (size_t)dir * 4 * H + j;
            const float gi = gr[0] + mine[0], gf = gr[H] + mine[1], gg = gr[2 * H] + mine[2], go = gr[3 * H] + mine[3];
            const float c_old = c[sidx];
            const float c_new = sigm(gf) * c_old + sigm(gi) * tanhf(gg);
            const float h_new = sigm(go) * tanhf(c_new);
            float* o = out + ((size_t)b * T + t) * (ndir * H) + dir * H + j;
            if (active) {
                c[sidx] = c_new;
                h_next[sidx] = h_new;
                *o = h_new;
            } else {
                h_next[sidx] = h_prev[sidx];
                *o = 0.f;
            }
        }
    }
}

// Batched form of the step (B > 4): 8 hidden units x 4 gates of a workgroup against the h_{t-1} rows of 16 * BT sequences on
// the matrix cores (v_mfma_f32_16x16x4_f32: D[16 sequences, 16 columns] += h[16, 4] . W^T[4, 16]; the 16 columns of tile p are
// gate 2p of the 8 units followed by gate 2p+1 of the same units).  1024 / 8 units x directions = 256 workgroups at two
// directions, one per CU, so the whole chip pulls W_hh.  8 waves split K = 1024 (128 each), partial tiles are summed through
// LDS, then one thread per (sequence, unit) applies the gate non-linearities.  Both operands are fetched as 16-byte vectors
// along k (lane l: row/col l % 16, k block 4 * (l / 16)); the k order inside an MFMA sum is free, so element i of every lane's
// vector feeds MFMA i of a group of four.
template <int H, int BT>
__global__ __launch_bounds__(512) void lstm_step_mfma_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                             const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                             float* __restrict__ c, float* __restrict__ out,
                                                             const int* __restrict__ lens, int B, int T, int step, int ndir) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    __shared__ float red[8][BT * 2 * 4][64];          // [wave][sequence tile, gate pair, acc reg][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * 8;
    const int t = dir ? T - 1 - step : step;
    const int col = lane & 15, q = lane >> 4;
    const int k0 = wave * (H / 8);
    f32x4v acc[BT][2];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[bt][p] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // column `col` of gate-pair tile p: gate 2p + (col >> 3), unit j0 + (col & 7)
    const float* wbase = whh + ((size_t)dir * 4 * H + (size_t)(col >> 3) * H + j0 + (col & 7)) * H + k0 + 4 * q;
    const float* hbase = h_prev + (size_t)dir * B * H + k0 + 4 * q;
#pragma unroll
    for (int kb = 0; kb < H / 8; kb += 16) {      // fully unrolled: all 8 x (2 + BT) vector loads can be in flight at once
        f32x4 wv[2], hv[BT];
#pragma unroll
        for (int p = 0; p < 2; ++p) wv[p] = *reinterpret_cast<const f32x4*>(wbase + (size_t)(2 * p) * H * H + kb);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
            const int b = min(bt * 16 + col, B - 1);
            hv[bt] = *reinterpret_cast<const f32x4*>(hbase + (size_t)b * H + kb);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    acc[bt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[bt][i], wv[p][i], acc[bt][p], 0, 0, 0);
    }
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(bt * 2 + p) * 4 + r][lane] = acc[bt][p][r];
    __syncthreads();
    // D layout of the 16x16 tile: lane l, register r -> row (sequence) 4 * (l / 16) + r, column l % 16
    for (int e = threadIdx.x; e < BT * 16 * 8; e += 512) {       // one thread per (sequence, unit)
        const int b = e >> 3, u = e & 7;
        if (b >= B) continue;
        const int bt = b >> 4, r = b & 3, lq = (b >> 2) & 3;      // b = 16 bt + 4 lq + r
        float dot[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int l = lq * 16 + (g & 1) * 8 + u;              // lane that holds column (gate g & 1, unit u) of tile g >> 1
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[w][(bt * 2 + (g >> 1)) * 4 + r][l];
            dot[g] = a;
        }
        const int j = j0 + u;
        const size_t sidx = ((size_t)dir * B + b) * H + j;
        const bool active = !lens || t < lens[b];
        const float* gr = gx + ((size_t)b * T + t) * (ndir * 4 * H) + (size_t)dir * 4 * H + j;
        const float gi = gr[0] + dot[0], gf = gr[H] + dot[1], gg = gr[2 * H] + dot[2], go = gr[3 * H] + dot[3];
        const float c_old = c[sidx];
        const float c_new = sigm(gf) * c_old + sigm(gi) * tanhf(gg);
        const float h_new = sigm(go) * tanhf(c_new);
        float* o = out + ((size_t)b * T + t) * (ndir * H) + dir * H + j;
        if (active) {
            c[sidx] = c_new;
            h_next[sidx] = h_new;
            *o = h_new;
        } else {
            h_next[sidx] = h_prev[sidx];
            *o = 0.f;
        }
    }
}

void launch_lstm_step(const float* gx, const float* whh, const float* h_prev, float* h_next, float* c, float* out,
                      const int* lens, int B, int T, int H, int step, int ndir, hipStream_t s) {
    if (H != 1024) return;
    if (B > 4 && B <= 32) {           // matrix-core form: 8 units per workgroup, 16 * BT sequences
        const dim3 grid(H / 8, ndir), blk(512);
        if (B <= 16) hipLaunchKernelGGL((lstm_step_mfma_kernel<1024, 1>), grid, blk, 0, s, gx, whh, h_prev, h_next, c, out, lens, B, T, step, ndir);
        else hipLaunchKernelGGL((lstm_step_mfma_kernel<1024, 2>), grid, blk, 0, s, gx, whh, h_prev, h_next, c, out, lens, B, T, step, ndir);
        return;
    }
    hipLaunchKernelGGL(lstm_step_kernel<1024>, dim3(H / 4, ndir), dim3(256), 0, s, gx, whh, h_prev, h_next, c, out, lens, B,
                       T, step, ndir);
}

// LayerNorm over rows of arbitrary width N <= 8192 (deepspeech2/encoder.py:33,44): one workgroup per row
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* y, int N, float eps) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * N;
    float v[32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int k = threadIdx.x + i * 256;
        v[i] = k < N ? xr[k] : 0.f;
        s += v[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s = wave_sum_dpp(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)N;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int k = threadIdx.x + i * 256;
        const float d = k < N ? v[i] - mean : 0.f;
        v[i] = d;
        q += d * d;
    }
    q = wave_sum_dpp(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
    float* yr = y + (size_t)blockIdx.x * N;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int k = threadIdx.x + i * 256;
        if (k < N) yr[k] = v[i] * rstd * w[k] + b[k];
    }
}

void launch_layernorm_generic(const float* x, const float* w, const float* b, float* y, int M, int N, float eps,
                              hipStream_t s) {
    if (M <= 0 || N > 8192) return;
    hipLaunchKernelGGL(layernorm_generic_kernel, dim3(M), dim3(256), 0, s, x, w, b, y, N, eps);
}

// encoder frame counts of Conv2dSubsampling4Pure (conv.py:21): ((len - 1) / 2 - 1) / 2
__global__ void ds2_lens_kernel(const int* __restrict__ lens, int B, int Tq, int* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = max(0, min(Tq, ((lens[b] - 1) / 2 - 1) / 2));
}
void launch_ds2_lens(const int* lens, int B, int Tq, int* out, hipStream_t s) {
    hipLaunchKernelGGL(ds2_lens_kernel, dim3((B + 63) / 64), dim3(64), 0, s, lens, B, Tq, out);
}

}  // namespace masr
