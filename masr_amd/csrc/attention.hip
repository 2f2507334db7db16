// Relative-position multi-head self-attention of the reference Conformer
// (conformer/attention.py:190-251; rel_shift is removed there, :245-247):
//     score[i,j] = ((q_i + u) . k_j + (q_i + v) . p_j) / sqrt(d_k),  p_j = W_pos * PE(pos0 + j)
//     attn = softmax_j(mask(score)),  masked entries -> 0 ;  out_i = sum_j attn[i,j] v_j
//
// MI355X mapping (fp32, v_mfma_f32_32x32x2_f32, wave64):
//  * one wave owns 32 queries of one (sequence, head); 4 waves (128 queries) per workgroup share
//    LDS tiles of 32 keys: K, P (positional keys, from the table precomputed at weight load) and V.
//  * both score terms are ONE contraction over a concatenated 128-wide dimension:
//        S^T[key, query] = [k_j | p_j] . [q_i + u | q_i + v]
//    computed TRANSPOSED (A operand = keys, B operand = queries) so that every lane owns ONE query
//    column and 16 keys of it: the online softmax needs no LDS and only a lane^32 exchange.
//  * the probabilities stay in the accumulator registers and are fed straight back as the B operand
//    of O^T[d, query] = V^T[d, key] . P^T[key, query]: the MFMA k-slot of step r in lane half h is
//    key (r&3)+8(r>>2)+4h -- exactly the row this lane holds in C-register r.  V is read from LDS
//    with that key order as the A operand.  No shuffle, no LDS round trip for P.
//  * 1/sqrt(64) = 0.125 is folded into the query fragment (exact power of two).
#include "common.h"

namespace masr {

static constexpr int DK = 64;
static constexpr int KP_LD = 68;   // padded row of the K / P tiles (floats): 16B slot = (17*row + ..) mod 16
static constexpr int V_LD = 68;

__global__ __launch_bounds__(256) void attention_kernel(const AttSeq* __restrict__ seqs, int q_stride, int kv_stride,
                                                        const float* __restrict__ ptab,
                                                        const float* __restrict__ bias_u,
                                                        const float* __restrict__ bias_v, int chunk_size,
                                                        int pos_stride) {
    __shared__ __align__(16) float Ks[32 * KP_LD];
    __shared__ __align__(16) float Ps[32 * KP_LD];
    __shared__ __align__(16) float Vs[32 * V_LD];

    const AttSeq sq = seqs[blockIdx.z];
    const int head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 128;
    if (q0 >= sq.nq) return;                 // whole workgroup exits together
    const int qi = q0 + wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < sq.nq;
    const int q_abs = sq.q_abs0 + (q_ok ? qi : sq.nq - 1);

    // ---- query fragment: this lane's 32 dims {8g+4h+s} of (q+u)/8 and (q+v)/8 ----------------
    f32x4 qu[8], qv[8];
    {
        const float* qrow = sq.q + (size_t)(q_ok ? qi : sq.nq - 1) * q_stride + head * DK;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(qrow + 8 * g + 4 * h);
            const f32x4 u = *reinterpret_cast<const f32x4*>(bias_u + head * DK + 8 * g + 4 * h);
            const f32x4 v = *reinterpret_cast<const f32x4*>(bias_v + head * DK + 8 * g + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qu[g][s] = (q[s] + u[s]) * 0.125f;
                qv[g][s] = (q[s] + v[s]) * 0.125f;
            }
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // visible keys for this query: j < klen and (chunk mask) j < (q_abs / cs + 1) * cs
    int jlim = sq.klen;
    if (chunk_size > 0) jlim = min(jlim, (q_abs / chunk_size + 1) * chunk_size);

    const int ntile = (sq.nk + 31) / 32;
    // staging assignment: 512 float4 per 32x64 tile, 2 per thread
    const int srow = tid >> 4;            // 0..15 (+16)
    const int sc4 = (tid & 15) * 4;       // float offset 0..60
    // register-prefetched staging: tile kt+1 is fetched from global memory while tile kt is multiplied
    f32x4 pk[2], pp[2], pv[2];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = kt * 32 + srow + 16 * i;
            const int jc = min(j, sq.nk - 1);                      // clamped address, masked below
            pk[i] = *reinterpret_cast<const f32x4*>(sq.k + (size_t)jc * kv_stride + head * DK + sc4);
            pv[i] = *reinterpret_cast<const f32x4*>(sq.v + (size_t)jc * kv_stride + head * DK + sc4);
            pp[i] = *reinterpret_cast<const f32x4*>(ptab + (size_t)(sq.pos0 + jc * pos_stride) * 256 + head * DK + sc4);
            if (j >= sq.nk) { pk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[i] = pk[i]; pp[i] = pk[i]; }
        }
    };
    fetch(0);
    for (int kt = 0; kt < ntile; ++kt) {
        const int j0 = kt * 32;
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 16 * i;
            *reinterpret_cast<f32x4*>(&Ks[r * KP_LD + sc4]) = pk[i];
            *reinterpret_cast<f32x4*>(&Ps[r * KP_LD + sc4]) = pp[i];
            *reinterpret_cast<f32x4*>(&Vs[r * V_LD + sc4]) = pv[i];
        }
        __syncthreads();
        if (kt + 1 < ntile) fetch(kt + 1);

        // ---- S^T tile: rows = 32 keys, cols = this wave's 32 queries -------------------------
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        const float* kb = &Ks[(lane & 31) * KP_LD + 4 * h];
        const float* pb = &Ps[(lane & 31) * KP_LD + 4 * h];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qu[g][s], st, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 pf = *reinterpret_cast<const f32x4*>(pb + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[s], qv[g][s], st, 0, 0, 0);
        }

        // ---- mask + online softmax for this lane's query column --------------------------------
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j >= jlim) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = expf(m_run - m_safe);        // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - m_safe);                 // masked (-inf) -> 0
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }

        // ---- O^T += V^T . P^T ---------------------------------------------------------------------
        const float* vb = &Vs[(4 * h) * V_LD + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2);       // + 4h folded into vb
            const float va = vb[krow * V_LD];
            const float vb2 = vb[krow * V_LD + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va, st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2, st[r], o1, 0, 0, 0);
        }
    }

    // ---- normalise and store: lane owns query qi, dims d = (r&3) + 8(r>>2) + 4h (+32) -----------
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* orow = sq.out + (size_t)qi * 256 + head * DK;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int d = 8 * rr + 4 * h;
            f32x4 a, b;
#pragma unroll
            for (int s = 0; s < 4; ++s) { a[s] = o0[rr * 4 + s] * inv; b[s] = o1[rr * 4 + s] * inv; }
            *reinterpret_cast<f32x4*>(orow + d) = a;
            *reinterpret_cast<f32x4*>(orow + 32 + d) = b;
        }
    }
}

void launch_attention(const AttSeq* seqs, int nseq, int max_nq, int heads, int q_stride, int kv_stride,
                      const float* ptab, const float* bias_u, const float* bias_v, int chunk_size, int pos_stride,
                      hipStream_t s) {
    if (nseq <= 0 || max_nq <= 0) return;
    hipLaunchKernelGGL(attention_kernel, dim3((max_nq + 127) / 128, heads, nseq), dim3(256), 0, s, seqs, q_stride,
                       kv_stride, ptab, bias_u, bias_v, chunk_size, pos_stride);
}

// Sequence descriptors for the full-context batch path: q/k/v interleaved in one [B*Tp, 768] buffer
// (fused QKV projection), keys j valid iff mstride*j < len_b (subsampled pad mask, subsampling.py:112;
// mstride = 8 between the Squeezeformer time reduction and recovery).
__global__ void attseq_full_kernel(AttSeq* seqs, const float* qkv, float* out, const int* __restrict__ lens, int B,
                                   int Tp, int mstride) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    AttSeq s;
    s.q = qkv + (size_t)b * Tp * 768;
    s.k = s.q + 256;
    s.v = s.q + 512;
    s.out = out + (size_t)b * Tp * 256;
    s.nq = Tp;
    s.nk = Tp;
    s.klen = min(Tp, (lens[b] + mstride - 1) / mstride);
    s.pos0 = 0;
    s.q_abs0 = 0;
    s.pad_ = 0;
    seqs[b] = s;
}

void launch_attseq_full(AttSeq* seqs, const float* qkv, float* out, const int* lens, int B, int Tp, int mstride,
                        hipStream_t s) {
    hipLaunchKernelGGL(attseq_full_kernel, dim3((B + 63) / 64), dim3(64), 0, s, seqs, qkv, out, lens, B, Tp, mstride);
}

}  // namespace masr
