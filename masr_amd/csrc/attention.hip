// Relative-position multi-head self-attention of the reference Conformer
// (conformer/attention.py:190-251; rel_shift is removed there, :245-247):
//     score[i,j] = ((q_i + u) . k_j + (q_i + v) . p_j) / sqrt(d_k),  p_j = W_pos * PE(pos0 + j)
//     attn = softmax_j(mask(score)),  masked entries -> 0 ;  out_i = sum_j attn[i,j] v_j
//
// MI355X mapping (fp32, v_mfma_f32_32x32x2_f32, wave64):
//  * one wave owns 32 queries of one (sequence, head) and every second tile of 32 keys; 8 waves (128 queries x 2 key
//    parities) per workgroup share LDS tiles: K, P (positional keys, from the table precomputed at weight load) and V.
//  * both score terms are ONE contraction over a concatenated 128-wide dimension:
//        S^T[key, query] = [k_j | p_j] . [q_i + u | q_i + v]
//    computed TRANSPOSED (A operand = keys, B operand = queries) so that every lane owns ONE query
//    column and 16 keys of it: the online softmax needs no LDS and only a lane^32 exchange.
//  * the probabilities stay in the accumulator registers and are fed straight back as the B operand
//    of O^T[d, query] = V^T[d, key] . P^T[key, query]: the MFMA k-slot of step r in lane half h is
//    key (r&3)+8(r>>2)+4h -- exactly the row this lane holds in C-register r.  V is read from LDS
//    with that key order as the A operand.  No shuffle, no LDS round trip for P.
//  * 1/sqrt(64) = 0.125 is folded into the query fragment (exact power of two).
//  * the softmax exponentials use __expf (v_exp_f32): 2.5 us of the kernel's 41.7 us at B = 32 x 10 s against the libm expf,
//    encoder output moves by 4e-6 (tools/perf_ab.py of round 2); all three attention kernels use the same function.
#include "common.h"

namespace masr {

static constexpr int DK = 64;

// ATT_XLANE 1: cross-lane exchanges of attention_kernel on the VALU instead of through the LDS crossbar (__shfl_xor lowers to
// ds_bpermute_b32, one LDS round trip each): the lane ^ 32 exchange of the online softmax with v_permlane32_swap_b32 (gfx950),
// the 16-lane sum of the folded score constant with four DPP adds (quad_perm, row_half_mirror, row_mirror -- the same operand
// pairs as the xor butterfly, so the same bits).  0 keeps the shuffles (A/B builds).
#ifndef ATT_XLANE
#define ATT_XLANE 1
#endif
#ifndef ATT_XCD_MAP
#define ATT_XCD_MAP 1
#endif
__device__ __forceinline__ float att_xor32(float v, int h) {
#if ATT_XLANE
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: lanes 32.. hold lanes 0..31; r[1]: lanes 0..31 hold lanes 32..
    return __builtin_bit_cast(float, h ? r[0] : r[1]);
#else
    return __shfl_xor(v, 32, 64);
#endif
}
__device__ __forceinline__ float att_sum16(float c) {
#if ATT_XLANE
#define ATT_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, false))
    c += ATT_DPP(c, 0xB1);    // quad_perm [1,0,3,2]   (lane ^ 1)
    c += ATT_DPP(c, 0x4E);    // quad_perm [2,3,0,1]   (lane ^ 2)
    c += ATT_DPP(c, 0x141);   // row_half_mirror: quad q <-> quad 1 - q of each 8 lanes (all lanes of a quad hold its sum)
    c += ATT_DPP(c, 0x140);   // row_mirror: half <-> half of the 16 lanes
#undef ATT_DPP
    return c;
#else
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
    c += __shfl_xor(c, 4, 64);
    c += __shfl_xor(c, 8, 64);
    return c;
#endif
}
static constexpr int KP_LD = 68;   // padded row of the K / P tiles (floats): 16B slot = (17*row + ..) mod 16
static constexpr int V_LD = 68;

// 8 waves per workgroup: wave w = (query group w & 3, key parity w >> 2).  The two waves of a query group walk the even and the
// odd key tiles with their own online-softmax state and are merged once at the end, so every SIMD holds two waves (the softmax
// VALU work of one overlaps the MFMAs of the other) although B*H*T'/32 is only ~4 waves per CU at B = 32 x 10 s.
//
// FOLD = 1 (default): the two score terms share the query, so the positional keys are folded into the keys while the tile
// is staged:  (q+u).k + (q+v).p  =  q.(k+p) + (u.k + v.p).  The workgroup stages K' = k + p and the per-key constant
// c_j = (u.k_j + v.p_j)/8 (16 lanes per key row, 4 xor-shuffles); the score accumulator starts at c_j and ONE 64-wide
// contraction replaces the 128-wide one: 64 instead of 96 MFMAs per key tile, no P tile in LDS.  Same value up to fp32
// rounding (the reference's own summation order is not reproduced by either form); FOLD = 0 (masr_debug_set key 14) keeps the
// two-term contraction for A/B runs.
template <int FOLD>
__global__ __launch_bounds__(512) void attention_kernel(const AttSeq* __restrict__ seqs, int q_stride, int kv_stride,
                                                        const float* __restrict__ ptab,
                                                        const float* __restrict__ bias_u,
                                                        const float* __restrict__ bias_v, int chunk_size,
                                                        int pos_stride, int nqb_1d, int heads_1d, int nseq_1d) {
    __shared__ __align__(16) float lds_att[2 * 32 * KP_LD * 2 + 2 * 32 * V_LD + 64];
    float* Ks = lds_att;                       // [2 tiles][32][68]
    float* Ps = Ks + 2 * 32 * KP_LD;
    float* Vs = Ps + 2 * 32 * KP_LD;
    float* Cs = Vs + 2 * 32 * V_LD;            // FOLD: [2 tiles][32] per-key constants

    // 1-D launch (ATT_XCD_MAP): workgroup ids go round-robin over the 8 XCDs, so id -> (XCD = id % 8, k = id / 8); the query
    // blocks of one (sequence, head) take consecutive k on the SAME XCD -- its key / value rows are fetched into that XCD's L2
    // once, not once per query block -- and an XCD serves the (sequence, head) pairs p = XCD + 8 m, i.e. only heads XCD % 4:
    // the positional rows of a head stay in one L2 as before.  3-D launches (nqb_1d = 0) keep (query block, head, sequence).
    int qb = blockIdx.x, head = blockIdx.y, seq = blockIdx.z;
    if (nqb_1d > 0) {
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        qb = k % nqb_1d;
        const int pr = xcd + 8 * (k / nqb_1d);
        head = pr % heads_1d;
        seq = pr / heads_1d;
        if (seq >= nseq_1d) return;          // padding of the pair count to a multiple of 8 (whole workgroup)
    }
    const AttSeq sq = seqs[seq];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave & 3, kh = wave >> 2;
    const int q0 = qb * 128;
    if (q0 >= sq.nq) return;                 // whole workgroup exits together
    const int qi = q0 + qg * 32 + (lane & 31);
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
                qu[g][s] = FOLD ? q[s] * 0.125f : (q[s] + u[s]) * 0.125f;
                qv[g][s] = (q[s] + v[s]) * 0.125f;         // unused (dead) when FOLD
            }
        }
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // visible keys for this query: j < klen and (chunk mask) j < (q_abs / cs + 1) * cs
    // chunk_size counts frames of the encoder's input rate; a layer at 1/pos_stride of that rate sees rows and columns
    // 0, s, 2s, ... of the chunk mask (time_reduction.py:62, efficient_conformer/encoder.py:254-256): key j is visible to
    // query i iff s*j < ((s*i) / chunk + 1) * chunk
    int jlim = sq.klen;
    if (chunk_size > 0) jlim = min(jlim, (((q_abs * pos_stride) / chunk_size + 1) * chunk_size + pos_stride - 1) / pos_stride);
    const int ntile = (sq.nk + 31) / 32;
    const int npair = (ntile + 1) / 2;
    // staging assignment: two 32x64 tiles per iteration; thread t -> tile t >> 8, 512 float4 per tile, 2 per thread
    const int sti = tid >> 8;             // which tile of the pair this thread stages
    const int srow = (tid & 255) >> 4;    // 0..15 (+16)
    const int sc4 = (tid & 15) * 4;       // float offset 0..60
    f32x4 su = {0.f, 0.f, 0.f, 0.f}, sv = su;               // FOLD: u / 8 and v / 8 for this thread's 4 staged dims
    if (FOLD) {
        su = *reinterpret_cast<const f32x4*>(bias_u + head * DK + sc4);
        sv = *reinterpret_cast<const f32x4*>(bias_v + head * DK + sc4);
#pragma unroll
        for (int s = 0; s < 4; ++s) { su[s] *= 0.125f; sv[s] *= 0.125f; }
    }
    // register-prefetched staging: pair kp+1 is fetched from global memory while pair kp is multiplied
    f32x4 pk[2], pp[2], pv[2];
#if ATT_XLANE
    // raw buffer loads: the sequence's key / value rows and its positional rows behind descriptors in SGPRs, this thread's row and
    // column as ONE constant byte offset per operand, the tile pair as a scalar offset -- no 64-bit address arithmetic per load
    // next to the MFMAs; rows past the last key are out of the descriptor's range and read as zero
    const unsigned kv_row_bytes = (unsigned)kv_stride * 4u, p_row_bytes = (unsigned)pos_stride * 1024u;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.k), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.v), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptab + (size_t)sq.pos0 * 256), 0,
                                                                         (unsigned)sq.nk * p_row_bytes, 0x00020000);
    unsigned kvo[2], po[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned jr = (unsigned)(sti * 32 + srow + 16 * i);
        kvo[i] = jr * kv_row_bytes + (unsigned)(head * DK + sc4) * 4u;
        po[i] = jr * p_row_bytes + (unsigned)(head * DK + sc4) * 4u;
    }
    auto fetch = [&](int kp) {
        const unsigned skv = (unsigned)kp * 64u * kv_row_bytes, sp = (unsigned)kp * 64u * p_row_bytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[i], skv, 0));
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, kvo[i], skv, 0));
            pp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, po[i], sp, 0));
        }
    };
#else
    auto fetch = [&](int kp) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = (2 * kp + sti) * 32 + srow + 16 * i;
            const int jc = min(j, sq.nk - 1);                      // clamped address, masked below
            pk[i] = *reinterpret_cast<const f32x4*>(sq.k + (size_t)jc * kv_stride + head * DK + sc4);
            pv[i] = *reinterpret_cast<const f32x4*>(sq.v + (size_t)jc * kv_stride + head * DK + sc4);
            pp[i] = *reinterpret_cast<const f32x4*>(ptab + (size_t)(sq.pos0 + jc * pos_stride) * 256 + head * DK + sc4);
            if (j >= sq.nk) { pk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[i] = pk[i]; pp[i] = pk[i]; }
        }
    };
#endif
    fetch(0);
    for (int kp = 0; kp < npair; ++kp) {
        const int j0 = (2 * kp + kh) * 32;     // first key of this wave's tile
        __syncthreads();   // previous pair fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 16 * i;
            if (FOLD) {
                float c = 0.f;
                f32x4 kp4;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    c = fmaf(su[s], pk[i][s], c);
                    c = fmaf(sv[s], pp[i][s], c);
                    kp4[s] = pk[i][s] + pp[i][s];
                }
                c = att_sum16(c);
                *reinterpret_cast<f32x4*>(&Ks[sti * 32 * KP_LD + r * KP_LD + sc4]) = kp4;
                if ((tid & 15) == 0) Cs[sti * 32 + r] = c;
            } else {
                *reinterpret_cast<f32x4*>(&Ks[sti * 32 * KP_LD + r * KP_LD + sc4]) = pk[i];
                *reinterpret_cast<f32x4*>(&Ps[sti * 32 * KP_LD + r * KP_LD + sc4]) = pp[i];
            }
            *reinterpret_cast<f32x4*>(&Vs[sti * 32 * V_LD + r * V_LD + sc4]) = pv[i];
        }
        __syncthreads();
        if (kp + 1 < npair) fetch(kp + 1);
        if (j0 >= sq.nk) continue;             // (wave-uniform) odd tile past the end: nothing to add

        // ---- S^T tile: rows = 32 keys, cols = this wave's 32 queries -------------------------
        f32x16 st;
        if (FOLD) {            // rows (r & 3) + 8 (r >> 2) + 4 h: four aligned groups of four constants
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(&Cs[kh * 32 + 8 * rr + 4 * h]);
#pragma unroll
                for (int q = 0; q < 4; ++q) st[4 * rr + q] = c4[q];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
        }
        const float* kb = &Ks[kh * 32 * KP_LD + (lane & 31) * KP_LD + 4 * h];
        const float* pb = &Ps[kh * 32 * KP_LD + (lane & 31) * KP_LD + 4 * h];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qu[g][s], st, 0, 0, 0);
        }
        if (!FOLD) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 pf = *reinterpret_cast<const f32x4*>(pb + 8 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[s], qv[g][s], st, 0, 0, 0);
            }
        }

        // ---- mask + online softmax for this lane's query column --------------------------------
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j >= jlim) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, att_xor32(tmax, h));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = __expf(m_run - m_safe);        // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_safe);                 // masked (-inf) -> 0
            psum += st[r];
        }
        psum += att_xor32(psum, h);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }

        // ---- O^T += V^T . P^T ---------------------------------------------------------------------
        const float* vb = &Vs[kh * 32 * V_LD + (4 * h) * V_LD + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2);       // + 4h folded into vb
            const float va = vb[krow * V_LD];
            const float vb2 = vb[krow * V_LD + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va, st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2, st[r], o1, 0, 0, 0);
        }
    }

    // ---- merge the odd-tile state into the even-tile wave of the same query group (through the tile LDS) -------------
    __syncthreads();
    float* mg = lds_att;                                // [4 query groups][34][64 lanes] floats (34.8 KB <= Ks + Ps)
    static_assert(4 * 34 * 64 <= 2 * 2 * 32 * KP_LD, "merge buffer must fit into the K and P tiles");
    if (kh == 1) {
        float* d = mg + (size_t)qg * 34 * 64 + lane;
        d[0] = m_run;
        d[64] = l_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[(2 + r) * 64] = o0[r]; d[(18 + r) * 64] = o1[r]; }
    }
    __syncthreads();
    if (kh == 1) return;
    {
        const float* d = mg + (size_t)qg * 34 * 64 + lane;
        const float m1 = d[0], l1 = d[64];
        const float m = fmaxf(m_run, m1);
        const float ms = (m == -INFINITY) ? 0.f : m;
        const float c0 = __expf(m_run - ms), c1 = __expf(m1 - ms);     // -inf -> 0
        l_run = l_run * c0 + l1 * c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * c0 + d[(2 + r) * 64] * c1;
            o1[r] = o1[r] * c0 + d[(18 + r) * 64] * c1;
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

// Few queries per sequence (streaming chunk steps: 16 new frames against a growing key cache).  attention_kernel gives
// every query group of 32 its own wave pair and walks the key tiles two at a time, so with <= 32 queries six of the eight
// waves multiply clamped duplicates and the time grows by ~4 us per 64 cached keys.  Here the eight waves of a workgroup
// share the ONE query group and split the KEYS: wave w owns key tiles w, w+8, ... (one tile each up to 256 keys) with its own
// online-softmax state; the eight states are merged once through LDS (each wave finishes 4 of the 32 output registers).
// The K / P / V fragments are read straight from global memory in MFMA operand layout -- all loads of a tile are issued
// before the first use, no LDS staging, no workgroup barrier before the merge.  Same arithmetic as attention_kernel.
__global__ __launch_bounds__(512) void attention_fewq_kernel(const AttSeq* __restrict__ seqs, int q_stride, int kv_stride,
                                                             const float* __restrict__ ptab,
                                                             const float* __restrict__ bias_u,
                                                             const float* __restrict__ bias_v, int chunk_size,
                                                             int pos_stride) {
    __shared__ float mg[8 * 34 * 64];          // [wave][m, l, o0[16], o1[16]][lane]
    const AttSeq sq = seqs[blockIdx.y];
    const int head = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.z = query group of 32 (chunk steps: one; short offline utterances: T' / 32, see launch_attention)
    if ((int)blockIdx.z * 32 >= sq.nq) return;            // whole workgroup
    const int qi = (int)blockIdx.z * 32 + (lane & 31), h = lane >> 5;
    const bool q_ok = qi < sq.nq;
    const int qrow_i = q_ok ? qi : sq.nq - 1;
    const int q_abs = sq.q_abs0 + qrow_i;

    f32x4 qu[8], qv[8];
    {
        const float* qrow = sq.q + (size_t)qrow_i * q_stride + head * DK;
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
    int jlim = sq.klen;
    if (chunk_size > 0) jlim = min(jlim, (((q_abs * pos_stride) / chunk_size + 1) * chunk_size + pos_stride - 1) / pos_stride);
    const int ntile = (sq.nk + 31) / 32;

#if ATT_XLANE
    // raw buffer loads (as in attention_kernel): descriptors over this sequence's keys / values / positional rows, ONE constant
    // per-lane offset per operand, the tile (and, for the value rows, the key of accumulator register r) as a wave-uniform scalar
    // offset -- no per-load address arithmetic in a kernel whose time is its dependent chain; rows past the last key read as zero
    // (their keys are masked: probability 0 times 0)
    const unsigned kv_row_bytes = (unsigned)kv_stride * 4u, p_row_bytes = (unsigned)pos_stride * 1024u;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.k), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.v), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptab + (size_t)sq.pos0 * 256), 0,
                                                                         (unsigned)sq.nk * p_row_bytes, 0x00020000);
    const unsigned ko = (unsigned)(lane & 31) * kv_row_bytes + (unsigned)(head * DK + 4 * h) * 4u;
    const unsigned po = (unsigned)(lane & 31) * p_row_bytes + (unsigned)(head * DK + 4 * h) * 4u;
    const unsigned vo = (unsigned)(4 * h) * kv_row_bytes + (unsigned)(head * DK + (lane & 31)) * 4u;
#endif
    for (int t = wave; t < ntile; t += 8) {
        const int j0 = t * 32;
        f32x4 kf[8], pf[8];
        float va[16], vb[16];
#if ATT_XLANE
#pragma unroll
        for (int g = 0; g < 8; ++g)
            kf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, ko + 32u * g, (unsigned)j0 * kv_row_bytes, 0));
#pragma unroll
        for (int g = 0; g < 8; ++g)
            pf[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, po + 32u * g, (unsigned)j0 * p_row_bytes, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)(j0 + (r & 3) + 8 * (r >> 2)) * kv_row_bytes;
            va[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, vo, so, 0));
            vb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, vo + 128u, so, 0));
        }
#else
        // ---- all operand loads of this tile (clamped rows: masked keys get probability 0, times a finite value) ----------
        const int jr = min(j0 + (lane & 31), sq.nk - 1);
        const float* kp = sq.k + (size_t)jr * kv_stride + head * DK + 4 * h;
        const float* pp = ptab + (size_t)(sq.pos0 + jr * pos_stride) * 256 + head * DK + 4 * h;
#pragma unroll
        for (int g = 0; g < 8; ++g) kf[g] = *reinterpret_cast<const f32x4*>(kp + 8 * g);
#pragma unroll
        for (int g = 0; g < 8; ++g) pf[g] = *reinterpret_cast<const f32x4*>(pp + 8 * g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jv = min(j0 + (r & 3) + 8 * (r >> 2) + 4 * h, sq.nk - 1);
            const float* vp = sq.v + (size_t)jv * kv_stride + head * DK + (lane & 31);
            va[r] = vp[0];
            vb[r] = vp[32];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);

        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[g][s], qu[g][s], st, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[g][s], qv[g][s], st, 0, 0, 0);

        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j >= jlim) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, att_xor32(tmax, h));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = __expf(m_run - m_safe);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_safe);
            psum += st[r];
        }
        psum += att_xor32(psum, h);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[r], st[r], o1, 0, 0, 0);
        }
    }

    // ---- merge the eight key-split states (ascending wave order) ----------------------------------------------------------
    {
        float* d = mg + (size_t)wave * 34 * 64 + lane;
        d[0] = m_run;
        d[64] = l_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[(2 + r) * 64] = o0[r]; d[(18 + r) * 64] = o1[r]; }
    }
    __syncthreads();
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) m = fmaxf(m, mg[(size_t)w * 34 * 64 + lane]);
    const float ms = (m == -INFINITY) ? 0.f : m;
    float l = 0.f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rbase = 2 + (wave >> 2) * 16 + (wave & 3) * 4;          // my 4 registers: o0 (waves 0-3) or o1 (waves 4-7), group wave & 3
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float* d = mg + (size_t)w * 34 * 64 + lane;
        const float c = __expf(d[0] - ms);                                // -inf -> 0
        l += d[64] * c;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] += d[(rbase + s) * 64] * c;
    }
    if (q_ok) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        float* orow = sq.out + (size_t)qi * 256 + head * DK + (wave >> 2) * 32 + 8 * (wave & 3) + 4 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] *= inv;
        *reinterpret_cast<f32x4*>(orow) = acc;
    }
}

static int g_fewq = 1, g_fold = 1;
static int g_fewq_wgs = 48;       // masr_debug_set key 28: offline launches with fewer attention_kernel workgroups than this take the key-split kernel
void set_attention_fewq_wgs(int n) { g_fewq_wgs = n; }
void set_attention_fewq(int on) { g_fewq = on; }
void set_attention_fold(int on) { g_fold = on; }

void launch_attention(const AttSeq* seqs, int nseq, int max_nq, int heads, int q_stride, int kv_stride,
                      const float* ptab, const float* bias_u, const float* bias_v, int chunk_size, int pos_stride,
                      hipStream_t s) {
    if (nseq <= 0 || max_nq <= 0) return;
    const int nqb = (max_nq + 127) / 128;
    // the key-split kernel (eight waves share 32 queries and split the key tiles) also takes short offline batches, one
    // workgroup per (head, sequence, 32 queries): attention_kernel would give such a batch nqb * heads * nseq workgroups that each
    // walk ALL key tiles in pairs (one 8.4 s utterance: 8 workgroups, 22 us; here 28 workgroups of one tile per wave)
    if (g_fewq && (max_nq <= 32 || nqb * heads * nseq < g_fewq_wgs)) {
        hipLaunchKernelGGL(attention_fewq_kernel, dim3(heads, nseq, (max_nq + 31) / 32), dim3(512), 0, s, seqs, q_stride, kv_stride,
                           ptab, bias_u, bias_v, chunk_size, pos_stride);
        return;
    }
    if (g_fold && ATT_XCD_MAP)
        hipLaunchKernelGGL(attention_kernel<1>, dim3(8 * nqb * ((heads * nseq + 7) / 8)), dim3(512), 0, s, seqs, q_stride,
                           kv_stride, ptab, bias_u, bias_v, chunk_size, pos_stride, nqb, heads, nseq);
    else if (g_fold)
        hipLaunchKernelGGL(attention_kernel<1>, dim3(nqb, heads, nseq), dim3(512), 0, s, seqs, q_stride,
                           kv_stride, ptab, bias_u, bias_v, chunk_size, pos_stride, 0, heads, nseq);
    else
        hipLaunchKernelGGL(attention_kernel<0>, dim3(nqb, heads, nseq), dim3(512), 0, s, seqs, q_stride,
                           kv_stride, ptab, bias_u, bias_v, chunk_size, pos_stride, 0, heads, nseq);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: attention + [out-projection + residual -> LayerNorm -> pointwise_conv1 -> GLU] of an offline Conformer layer in ONE
// kernel (conformer/attention.py:190-251 + encoder.py:123-131 + convolution.py:98-119).  One workgroup = 32 queries of one
// sequence x ALL FOUR heads, so that the [32, 256] context rows the out-projection contracts over never leave the CU:
//   attention phase   wave w = (head w & 3, key parity w >> 2): the scheme of attention_kernel<FOLD = 1> per head -- folded
//                     positional keys, transposed scores, online softmax in registers, even / odd key tiles merged at the end --
//                     with the K' / V tiles of all four heads staged per tile pair (16 lanes per (key row, head): the same
//                     partial sums of the folded constant in the same order); the merged, normalised context goes into an LDS
//                     tile [32][256] instead of HBM;
//   chain phase       rowgemm EPI_CHAIN's three weight tiles on that tile (packed [Wo; W_pw1] fragments through buffer loads):
//                     x <- x + ctx . Wo^T + bo (to HBM and into a second LDS tile), LayerNorm + pad mask in place, pointwise_conv1,
//                     GLU -> the conv module's padded GLU buffer.
// Same operations in the same order as attention_kernel<1> followed by rowgemm_kernel<PLAIN, CHAIN, 0, 1>: bit-identical
// (tests/test_gpu_few_rows.py).  Why: both of those are short kernels whose prologue / staging / epilogue cost as much as their
// matrix work (0.38 and 0.49 matrix-pipe busy at B = 32 x 10 s) and the att [M, 256] round trip sits between them.
// masr_debug_set key 34 = 0 keeps the two launches.
// ------------------------------------------------------------------------------------------------------------------
#if MASR_EXPERIMENTS
static constexpr int AC_LD = 68;                      // K' / V tile rows (floats)
static constexpr int AC_TILE = 4 * 32 * AC_LD;        // one key tile, four heads
static constexpr int AC_ALD = 256 + 4;                // context / LayerNorm tiles
static constexpr int AC_LDS_FLOATS = 4 * AC_TILE + 2 * 4 * 32 + 64;

__global__ __launch_bounds__(512) void attn_chain_kernel(AttnChainArgs p) {
    extern __shared__ __align__(16) float acs[];
    float* Ks = acs;                         // [2 tiles][4 heads][32][68]
    float* Vs = Ks + 2 * AC_TILE;            // [2 tiles][4 heads][32][68]
    float* Cs = Vs + 2 * AC_TILE;            // [2 tiles][4 heads][32] folded per-key constants
    // workgroup -> (sequence, query block): ids go round-robin over the 8 XCDs; the query blocks of a sequence take consecutive
    // slots of ONE XCD, whose L2 then serves that sequence's keys / values to all of them
    const int id = blockIdx.x, xcd = id & 7, kslot = id >> 3;
    const int qb = kslot % p.nqb, seq = xcd + 8 * (kslot / p.nqb);
    if (seq >= p.nseq) return;
    const AttSeq sq = p.seqs[seq];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = wave & 3, kh = wave >> 2;
    const int q0 = qb * 32;
    if (q0 >= sq.nq) return;                 // whole workgroup
    const int qi = q0 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < sq.nq;
    const int q_abs = sq.q_abs0 + (q_ok ? qi : sq.nq - 1);

    f32x4 qu[8];
    {
        const float* qrow = sq.q + (size_t)(q_ok ? qi : sq.nq - 1) * p.q_stride + head * DK;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(qrow + 8 * g + 4 * h);
#pragma unroll
            for (int x = 0; x < 4; ++x) qu[g][x] = q[x] * 0.125f;
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    int jlim = sq.klen;
    if (p.chunk_size > 0) jlim = min(jlim, (((q_abs * p.pos_stride) / p.chunk_size + 1) * p.chunk_size + p.pos_stride - 1) / p.pos_stride);
    const int ntile = (sq.nk + 31) / 32;
    const int npair = (ntile + 1) / 2;
    // staging: wave w moves key rows w, w + 8, ... of the tile pair; lane = (head lane >> 4, dims 4 (lane & 15) ..)
    const int shead = lane >> 4, sc4 = (lane & 15) * 4;
    f32x4 su, sv;
    {
        su = *reinterpret_cast<const f32x4*>(p.bias_u + shead * DK + sc4);
        sv = *reinterpret_cast<const f32x4*>(p.bias_v + shead * DK + sc4);
#pragma unroll
        for (int x = 0; x < 4; ++x) { su[x] *= 0.125f; sv[x] *= 0.125f; }
    }
    const unsigned kv_row_bytes = (unsigned)p.kv_stride * 4u, p_row_bytes = (unsigned)p.pos_stride * 1024u;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.k), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sq.v), 0, (unsigned)sq.nk * kv_row_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ptab + (size_t)sq.pos0 * 256), 0,
                                                                         (unsigned)sq.nk * p_row_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)lane * 16u;                      // 16 bytes per lane: the 256 floats of a key row
    f32x4 pk[8], pp[8], pv[8];
    auto fetch = [&](int kp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned j = (unsigned)(kp * 64 + wave + 8 * i);       // wave-uniform key row (past the end: reads zero)
            pk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, lane_off, j * kv_row_bytes, 0));
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, lane_off, j * kv_row_bytes, 0));
            pp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, lane_off, j * p_row_bytes, 0));
        }
    };
    fetch(0);
    for (int kp = 0; kp < npair; ++kp) {
        const int j0 = (2 * kp + kh) * 32;     // first key of this wave's tile
        __syncthreads();   // previous pair fully consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r64 = wave + 8 * i, tl = r64 >> 5, r = r64 & 31;
            float c = 0.f;
            f32x4 kp4;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                c = fmaf(su[x], pk[i][x], c);
                c = fmaf(sv[x], pp[i][x], c);
                kp4[x] = pk[i][x] + pp[i][x];
            }
            c = att_sum16(c);
            float* base = Ks + tl * AC_TILE + (shead * 32 + r) * AC_LD + sc4;
            *reinterpret_cast<f32x4*>(base) = kp4;
            *reinterpret_cast<f32x4*>(base + 2 * AC_TILE) = pv[i];       // Vs = Ks + 2 tiles
            if ((lane & 15) == 0) Cs[(tl * 4 + shead) * 32 + r] = c;
        }
        __syncthreads();
        if (kp + 1 < npair) fetch(kp + 1);
        if (j0 >= sq.nk) continue;             // (wave-uniform) odd tile past the end

        f32x16 st;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(&Cs[(kh * 4 + head) * 32 + 8 * rr + 4 * h]);
#pragma unroll
            for (int q = 0; q < 4; ++q) st[4 * rr + q] = c4[q];
        }
        const float* kb = Ks + kh * AC_TILE + (head * 32 + (lane & 31)) * AC_LD + 4 * h;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
            for (int x = 0; x < 4; ++x) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[x], qu[g][x], st, 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j >= jlim) st[r] = -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, att_xor32(tmax, h));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = __expf(m_run - m_safe);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_safe);
            psum += st[r];
        }
        psum += att_xor32(psum, h);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }
        const float* vb = Vs + kh * AC_TILE + (head * 32 + 4 * h) * AC_LD + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2);
            const float va = vb[krow * AC_LD];
            const float vb2 = vb[krow * AC_LD + 32];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va, st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb2, st[r], o1, 0, 0, 0);
        }
    }

    // ---- chain operands that do not depend on the attention: requested now, they land under the merge --------------------
    const int frow = lane & 31, fh = h;
    const int row0 = seq * p.seq_t + q0;                    // first row of this block in the flat [B * T', 256] row space
    const int nrow = min(32, sq.nq - q0);                   // valid rows of the block
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Wp), 0, 3 * 256 * 256 * 4, 0x00020000);
    auto pld = [&](int t, int j, int g) -> f32x4 {     // fragment (slab j, group g) of this wave's tile t (clamped past the end)
        const unsigned fi = (unsigned)(((min(t, 2) * 8 + wave) * 8 + j) * 4 + g) * 256u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, fi * 4u, 0));
    };
    f32x4 pre[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[k][g] = pld(0, k, g);
    const f32x4 cgw = *reinterpret_cast<const f32x4*>(p.lnw + lane * 4);
    const f32x4 cgb = *reinterpret_cast<const f32x4*>(p.lnb + lane * 4);
    const float cbo = p.bias[wave * 32 + frow], cba = p.bias[256 + wave * 32 + frow], cbg = p.bias[512 + wave * 32 + frow];
    float res0[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = min((r & 3) + 8 * (r >> 2) + 4 * fh, nrow - 1);
        res0[r] = p.R[(size_t)(row0 + lr) * 256 + wave * 32 + frow];
    }

    // ---- merge the odd-tile state into the even-tile wave of the same head; context rows -> LDS tile -------------------------
    __syncthreads();
    float* mg = acs;                                         // [4 heads][34][64]  (K tiles: all consumed)
    float* ctx = acs + 2 * AC_TILE;                          // [32][260]          (V tiles: all consumed)
    static_assert(4 * 34 * 64 <= 2 * AC_TILE && 32 * AC_ALD <= 2 * AC_TILE, "merge buffer / context tile must fit the K / V tiles");
    if (kh == 1) {
        float* d = mg + (size_t)head * 34 * 64 + lane;
        d[0] = m_run;
        d[64] = l_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[(2 + r) * 64] = o0[r]; d[(18 + r) * 64] = o1[r]; }
    }
    __syncthreads();
    if (kh == 0) {
        const float* d = mg + (size_t)head * 34 * 64 + lane;
        const float m1 = d[0], l1 = d[64];
        const float m = fmaxf(m_run, m1);
        const float ms = (m == -INFINITY) ? 0.f : m;
        const float c0 = __expf(m_run - ms), c1 = __expf(m1 - ms);     // -inf -> 0
        l_run = l_run * c0 + l1 * c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] = o0[r] * c0 + d[(2 + r) * 64] * c1;
            o1[r] = o1[r] * c0 + d[(18 + r) * 64] * c1;
        }
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* crow = ctx + (lane & 31) * AC_ALD + head * DK;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int dd = 8 * rr + 4 * h;
            f32x4 a, b;
#pragma unroll
            for (int x = 0; x < 4; ++x) { a[x] = o0[rr * 4 + x] * inv; b[x] = o1[rr * 4 + x] * inv; }
            if (!q_ok) { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
            *reinterpret_cast<f32x4*>(crow + dd) = a;
            *reinterpret_cast<f32x4*>(crow + 32 + dd) = b;
        }
    }
    __syncthreads();                                         // context tile complete; the merge buffer is dead

    // ---- chain phase: rowgemm EPI_CHAIN on the context tile (three weight tiles, wave w = output columns 32 w .. 32 w + 31) ----
    float* red = acs;                                        // [32][260] second A tile (updated x rows, then their LayerNorm)
    const float* aa = ctx + frow * AC_ALD + 4 * fh;
    f32x16 accv;
    for (int t = 0; t < 3; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 a[2];
            a[0] = *reinterpret_cast<const f32x4*>(aa + j * 32);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(aa + j * 32 + 8 * (g + 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j % 4][g][q], acc, 0, 0, 0);
                    if (q == 3) pre[j % 4][g] = pld(t + (j + 4) / 8, (j + 4) & 7, g);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int ch = wave * 32 + frow;
        if (t == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float v = res0[r] + (acc[r] + cbo);
                if (lr < nrow) p.R2[(size_t)(row0 + lr) * 256 + ch] = v;
                red[lr * AC_ALD + ch] = v;
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int lr = wave * 4 + rr;
                bool live = lr < nrow;
                if (live && p.lens) live = p.mstride * (q0 + lr) < p.lens[seq];
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[lr * AC_ALD + lane * 4]);
                const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = wave_sum_dpp(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + p.eps);
                f32x4 o;
                o[0] = d0 * rstd * cgw[0] + cgb[0];
                o[1] = d1 * rstd * cgw[1] + cgb[1];
                o[2] = d2 * rstd * cgw[2] + cgb[2];
                o[3] = d3 * rstd * cgw[3] + cgb[3];
                if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&red[lr * AC_ALD + lane * 4]) = o;
            }
            __syncthreads();
            aa = red + frow * AC_ALD + 4 * fh;
        } else if (t == 1) {
            accv = acc;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (lr >= nrow) continue;
                const size_t crow = (size_t)seq * (p.seq_t + p.out_pad_tot) + p.out_pad_l + q0 + lr;
                const float g = acc[r] + cbg;
                p.C[crow * 256 + ch] = (accv[r] + cba) * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
            }
        }
    }
}

bool launch_attn_chain(const AttnChainArgs& a, int max_nq, hipStream_t s) {
    if (a.nseq <= 0 || max_nq <= 0) return false;
    AttnChainArgs b = a;
    b.nqb = (max_nq + 31) / 32;
    const size_t lds = (size_t)AC_LDS_FLOATS * sizeof(float);
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(attn_chain_kernel), lds, attr);
    hipLaunchKernelGGL(attn_chain_kernel, dim3(8 * b.nqb * ((a.nseq + 7) / 8)), dim3(512), lds, s, b);
    return true;
}
#endif   // MASR_EXPERIMENTS

// Sequence descriptors for the full-context batch path: q/k/v interleaved in one [B*Tp, 768] buffer
// (fused QKV projection), keys j valid iff mstride*j < len_b (subsampled pad mask, subsampling.py:112;
// mstride = 8 between the Squeezeformer time reduction and recovery).
__global__ void attseq_full_kernel(AttSeq* seqs, const float* qkv, float* out, const int* __restrict__ lens, int B,
                                   int Tp, int mstride, int valid_only) {
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
    if (valid_only) s.nq = s.nk = s.klen;      // padded queries are not computed, padded keys not even read (they are masked anyway)
    s.pos0 = 0;
    s.q_abs0 = 0;
    s.pad_ = 0;
    seqs[b] = s;
}

void launch_attseq_full(AttSeq* seqs, const float* qkv, float* out, const int* lens, int B, int Tp, int mstride,
                        hipStream_t s, int valid_only) {
    hipLaunchKernelGGL(attseq_full_kernel, dim3((B + 63) / 64), dim3(64), 0, s, seqs, qkv, out, lens, B, Tp, mstride, valid_only);
}


// ------------------------------------------------------------------------------------------------
// Grouped relative-position attention of the Efficient Conformer (efficient_conformer/attention.py:35-69,
// 120-182, group_size g = 3): q, k, v, p [T, H*dk] are zero-padded in time to a multiple of g and FLAT-reshaped
// to [T/g, H, g*dk]; attention then runs over T/g positions with d_k' = g*dk = 192 and scale 1/sqrt(192).
// Same transposed-score scheme as attention_kernel, with the head dimension templated: the query fragment
// (q only, 96 registers) stays in registers, u / v are added on the fly from LDS, O^T uses DKG/32 accumulators.
// Inputs are PLANAR buffers [B][Tpad][256] (flat == [B][T/g][H][g*dk]); P comes from the per-layer positional
// table, rows past the true length T read as zero (the reference pads P with zeros, not with PE).
// ------------------------------------------------------------------------------------------------
template <int DKG, int NW>
__global__ __launch_bounds__(64 * NW) void attention_grouped_kernel(const AttSeq* __restrict__ seqs, int row_stride,
                                                                      const float* __restrict__ ptab, int t_true,
                                                                      const float* __restrict__ bias_u,
                                                                      const float* __restrict__ bias_v, float scale,
                                                                      int chunk_size, int group) {
    constexpr int LD = DKG + 4;
    constexpr int NG = DKG / 8;       // 8-wide k groups per operand half
    constexpr int NT = DKG / 32;      // 32-row output tiles of O^T
    extern __shared__ __align__(16) float smg[];
    float* Ks = smg;                  // [32][LD]
    float* Ps = Ks + 32 * LD;
    float* Vs = Ps + 32 * LD;
    float* Us = Vs + 32 * LD;         // [DKG] u, [DKG] v of this head
    const AttSeq sq = seqs[blockIdx.z];
    const int head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 32 * NW;
    if (q0 >= sq.nq) return;
    const int qi = q0 + wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < sq.nq;
    for (int i = tid; i < 2 * DKG; i += 64 * NW) Us[i] = i < DKG ? bias_u[head * DKG + i] : bias_v[head * DKG + i - DKG];

    f32x4 qf[NG];
    {
        const float* qrow = sq.q + (size_t)(q_ok ? qi : sq.nq - 1) * row_stride + head * DKG;
#pragma unroll
        for (int g = 0; g < NG; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g + 4 * h);
    }
    f32x16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // chunk mask of the grouped positions: rows and columns 0, g, 2g, ... of the frame-rate mask (pad4group, attention.py:35-69)
    int jlim = sq.klen;
    if (chunk_size > 0) {
        const int qa = (q_ok ? qi : sq.nq - 1) * group;
        jlim = min(jlim, ((qa / chunk_size + 1) * chunk_size + group - 1) / group);
    }
    const int ntile = (sq.nk + 31) / 32;
    constexpr int F4_PER_ROW = DKG / 4;
    for (int kt = 0; kt < ntile; ++kt) {
        const int j0 = kt * 32;
        __syncthreads();
        for (int i = tid; i < 32 * F4_PER_ROW; i += 64 * NW) {
            const int r = i / F4_PER_ROW, c4 = (i % F4_PER_ROW) * 4;
            const int j = j0 + r;
            f32x4 kk = f32x4{0.f, 0.f, 0.f, 0.f}, pp = kk, vv = kk;
            if (j < sq.nk) {
                kk = *reinterpret_cast<const f32x4*>(sq.k + (size_t)j * row_stride + head * DKG + c4);
                vv = *reinterpret_cast<const f32x4*>(sq.v + (size_t)j * row_stride + head * DKG + c4);
                // flat element index of P'[j][head][c4] in the [T,256] positional-key matrix
                const size_t e = (size_t)j * row_stride + head * DKG + c4;
                // (sq.pad_ > 0: per-sequence true key count -- streams in one lock-step call may have different histories)
                if ((int)(e / 256) < (sq.pad_ > 0 ? sq.pad_ : t_true)) pp = *reinterpret_cast<const f32x4*>(ptab + (size_t)sq.pos0 * 256 + e);
            }
            *reinterpret_cast<f32x4*>(&Ks[r * LD + c4]) = kk;
            *reinterpret_cast<f32x4*>(&Ps[r * LD + c4]) = pp;
            *reinterpret_cast<f32x4*>(&Vs[r * LD + c4]) = vv;
        }
        __syncthreads();

        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        const float* kb = &Ks[(lane & 31) * LD + 4 * h];
        const float* pb = &Ps[(lane & 31) * LD + 4 * h];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
            const f32x4 pf = *reinterpret_cast<const f32x4*>(pb + 8 * g);
            const f32x4 uu = *reinterpret_cast<const f32x4*>(&Us[8 * g + 4 * h]);
            const f32x4 vv = *reinterpret_cast<const f32x4*>(&Us[DKG + 8 * g + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[g][s] + uu[s], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(pf[s], qf[g][s] + vv[s], st, 0, 0, 0);
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            st[r] = j < jlim ? st[r] * scale : -INFINITY;
            tmax = fmaxf(tmax, st[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float corr = __expf(m_run - m_safe);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __expf(st[r] - m_safe);
            psum += st[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= corr;
        const float* vb = &Vs[(4 * h) * LD + (lane & 31)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[krow * LD + 32 * t], st[r], o[t], 0, 0, 0);
        }
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* orow = sq.out + (size_t)qi * row_stride + head * DKG;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                f32x4 a;
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = o[t][rr * 4 + s] * inv;
                *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * rr + 4 * h) = a;
            }
    }
}

// Round 4: the same attention with the positional keys FOLDED into the keys while a tile is staged, like attention_kernel<FOLD = 1>
// ((q+u).k + (q+v).p = q.(k+p) + (u.k + v.p): ONE 192-wide contraction per key tile instead of two; the per-key constant is
// reduced over the 16 lanes that stage a key row), and 96 queries x 2 = 6 waves per workgroup, so that a (sequence, head) of a 10 s
// utterance (83 grouped positions) is ONE workgroup with waves on all four SIMDs.  The two waves of a query group both compute the
// score tile and its online softmax (96 MFMAs, duplicated) and each accumulates HALF of the 192 output dims (48 MFMAs): 144
// MFMAs per wave and key tile, no merge at the end, and 96 + 48 instead of 96 + 96 persistent registers -- the even / odd key
// split of attention_kernel needs q and all of O in both waves, which at d_k' = 192 does not fit the 256 registers of a
// two-waves-per-SIMD kernel (measured: 46 spilled).  The two-wave kernel above ran 288 MFMAs per wave and key tile on two SIMDs
// of a CU: 58 us per launch at 32 x 10 s for a third of attention_kernel's work (26 us); masr_debug_set key 26 = 0 keeps it for A/B.
template <int DKG>
__global__ __launch_bounds__(384) void attention_grouped_fold_kernel(const AttSeq* __restrict__ seqs, int row_stride,
                                                                     const float* __restrict__ ptab, int t_true,
                                                                     const float* __restrict__ bias_u,
                                                                     const float* __restrict__ bias_v, float scale,
                                                                     int chunk_size, int group) {
    constexpr int LD = DKG + 4;
    constexpr int NG = DKG / 8;       // 8-wide k groups of the contraction
    constexpr int NT = DKG / 64;      // 32-row output tiles of O^T per wave (half of the DKG / 32)
    constexpr int SEG = DKG / 16;     // floats of a key row staged by one of its 16 threads (12)
    static_assert(SEG % 4 == 0 && DKG % 64 == 0, "a staging thread moves whole float4s; the output dims split in two halves");
    extern __shared__ __align__(16) float smf[];
    float* Ks = smf;                  // [2 tiles][32][LD]  k + p
    float* Vs = Ks + 2 * 32 * LD;     // [2 tiles][32][LD]
    float* Cs = Vs + 2 * 32 * LD;     // [2 tiles][32]      u.k + v.p
    const AttSeq sq = seqs[blockIdx.z];
    const int head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = wave % 3, oh = wave / 3;      // query group, output half
    const int q0 = blockIdx.x * 96;
    if (q0 >= sq.nq) return;                 // whole workgroup exits together
    const int qi = q0 + qg * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool q_ok = qi < sq.nq;

    f32x4 qf[NG];
    {
        const float* qrow = sq.q + (size_t)(q_ok ? qi : sq.nq - 1) * row_stride + head * DKG;
#pragma unroll
        for (int g = 0; g < NG; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g + 4 * h);
    }
    f32x16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    int jlim = sq.klen;
    if (chunk_size > 0) {
        const int qa = (q_ok ? qi : sq.nq - 1) * group;
        jlim = min(jlim, ((qa / chunk_size + 1) * chunk_size + group - 1) / group);
    }
    const int ntile = (sq.nk + 31) / 32;
    const int npair = (ntile + 1) / 2;
    // staging: 16 threads per key row, SEG floats each; 24 rows per pass, three passes cover the 64 rows of a tile pair
    const int srow = tid >> 4, sseg = (tid & 15) * SEG;
    const float* ub = bias_u + head * DKG + sseg;
    const float* vbias = bias_v + head * DKG + sseg;
    const int t_keys = sq.pad_ > 0 ? sq.pad_ : t_true;      // true key count in frames (per sequence for lock-step streams)
    for (int kp = 0; kp < npair; ++kp) {
        __syncthreads();   // previous pair fully consumed
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            const int r64 = pass * 24 + srow;          // row of the tile pair
            if (r64 < 64) {
                const int j = kp * 64 + r64;
                f32x4 kk[SEG / 4], pp[SEG / 4], vv[SEG / 4];
#pragma unroll
                for (int c = 0; c < SEG / 4; ++c) {
                    kk[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    pp[c] = kk[c];
                    vv[c] = kk[c];
                }
                if (j < sq.nk) {
                    const size_t e = (size_t)j * row_stride + head * DKG + sseg;      // flat element index in the [T, 256] matrices
#pragma unroll
                    for (int c = 0; c < SEG / 4; ++c) {
                        kk[c] = *reinterpret_cast<const f32x4*>(sq.k + e + 4 * c);
                        vv[c] = *reinterpret_cast<const f32x4*>(sq.v + e + 4 * c);
                        // the reference pads P with zeros (not with PE) beyond the true length: frame (e + 4c) / 256
                        if ((int)((e + 4 * c) / 256) < t_keys) pp[c] = *reinterpret_cast<const f32x4*>(ptab + (size_t)sq.pos0 * 256 + e + 4 * c);
                    }
                }
                float cst = 0.f;
                float* kd = &Ks[(r64 >> 5) * 32 * LD + (r64 & 31) * LD + sseg];
                float* vd = &Vs[(r64 >> 5) * 32 * LD + (r64 & 31) * LD + sseg];
#pragma unroll
                for (int c = 0; c < SEG / 4; ++c) {
                    f32x4 kp4;
                    const f32x4 su = *reinterpret_cast<const f32x4*>(ub + 4 * c), sv = *reinterpret_cast<const f32x4*>(vbias + 4 * c);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        cst = fmaf(su[x], kk[c][x], cst);
                        cst = fmaf(sv[x], pp[c][x], cst);
                        kp4[x] = kk[c][x] + pp[c][x];
                    }
                    *reinterpret_cast<f32x4*>(kd + 4 * c) = kp4;
                    *reinterpret_cast<f32x4*>(vd + 4 * c) = vv[c];
                }
                cst = att_sum16(cst);
                if ((tid & 15) == 0) Cs[r64] = cst;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < 2; ++kt) {
            const int j0 = (2 * kp + kt) * 32;     // first key of this tile
            if (j0 >= sq.nk) break;                // (uniform) odd tile past the end
            f32x16 st;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(&Cs[kt * 32 + 8 * rr + 4 * h]);
#pragma unroll
                for (int q = 0; q < 4; ++q) st[4 * rr + q] = c4[q];
            }
            const float* kb = &Ks[kt * 32 * LD + (lane & 31) * LD + 4 * h];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
                for (int x = 0; x < 4; ++x) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[x], qf[g][x], st, 0, 0, 0);
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                st[r] = j < jlim ? st[r] * scale : -INFINITY;
                tmax = fmaxf(tmax, st[r]);
            }
            tmax = fmaxf(tmax, att_xor32(tmax, h));
            const float m_new = fmaxf(m_run, tmax);
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float corr = __expf(m_run - m_safe);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __expf(st[r] - m_safe);
                psum += st[r];
            }
            psum += att_xor32(psum, h);
            l_run = l_run * corr + psum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= corr;
            const float* vb = &Vs[kt * 32 * LD + (4 * h) * LD + (lane & 31) + oh * 32 * NT];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[krow * LD + 32 * t], st[r], o[t], 0, 0, 0);
            }
        }
    }
    if (q_ok) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* orow = sq.out + (size_t)qi * row_stride + head * DKG + oh * 32 * NT;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                f32x4 a;
#pragma unroll
                for (int x = 0; x < 4; ++x) a[x] = o[t][rr * 4 + x] * inv;
                *reinterpret_cast<f32x4*>(orow + 32 * t + 8 * rr + 4 * h) = a;
            }
    }
}

static int g_att_grouped_fold = 1;      // masr_debug_set key 26: 0 = the two-wave, two-term grouped kernel (A/B)
void set_attention_grouped_fold(int on) { g_att_grouped_fold = on; }

void launch_attention_grouped(const AttSeq* seqs, int nseq, int max_nq, int heads, int group, const float* ptab,
                              int t_true, const float* bias_u, const float* bias_v, hipStream_t s, int chunk_size) {
    if (nseq <= 0 || max_nq <= 0 || group != 3) return;
    if (g_att_grouped_fold) {
        constexpr int DKG = 192;
        const size_t lds = (size_t)(2 * 2 * 32 * (DKG + 4) + 64) * sizeof(float);
        static LdsAttr attr;
        ensure_dynamic_lds(reinterpret_cast<const void*>(attention_grouped_fold_kernel<DKG>), lds, attr);
        hipLaunchKernelGGL((attention_grouped_fold_kernel<DKG>), dim3((max_nq + 95) / 96, heads, nseq), dim3(384), lds, s, seqs,
                           heads * DKG, ptab, t_true, bias_u, bias_v, 1.0f / sqrtf((float)DKG), chunk_size, group);
        return;
    }
    constexpr int DKG = 192, NW = 2;
    const size_t lds = (size_t)(3 * 32 * (DKG + 4) + 2 * DKG) * sizeof(float);
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(attention_grouped_kernel<DKG, NW>), lds, attr);
    hipLaunchKernelGGL((attention_grouped_kernel<DKG, NW>), dim3((max_nq + 32 * NW - 1) / (32 * NW), heads, nseq),
                       dim3(64 * NW), lds, s, seqs, heads * DKG, ptab, t_true, bias_u, bias_v, 1.0f / sqrtf((float)DKG),
                       chunk_size, group);
}

// descriptors for the grouped layout: planar q / k / v / out buffers [B][Tpad][256] == [B][Tg][H][g*dk];
// grouped key j is valid iff the pad mask keeps frame g*j, i.e. mstride * g * j < len
__global__ void attseq_grouped_kernel(AttSeq* seqs, const float* q, const float* k, const float* v, float* out,
                                      const int* __restrict__ lens, int B, int Tg, int group, int mstride) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t per = (size_t)Tg * group * 256;
    AttSeq s;
    s.q = q + b * per;
    s.k = k + b * per;
    s.v = v + b * per;
    s.out = out + b * per;
    s.nq = Tg;
    s.nk = Tg;
    const int ms = mstride * group;
    s.klen = min(Tg, (lens[b] + ms - 1) / ms);
    s.pos0 = 0;
    s.q_abs0 = 0;
    s.pad_ = 0;
    seqs[b] = s;
}

void launch_attseq_grouped(AttSeq* seqs, const float* q, const float* k, const float* v, float* out, const int* lens,
                           int B, int Tg, int group, int mstride, hipStream_t s) {
    hipLaunchKernelGGL(attseq_grouped_kernel, dim3((B + 63) / 64), dim3(64), 0, s, seqs, q, k, v, out, lens, B, Tg, group,
                       mstride);
}

}  // namespace masr
