// Row-block GEMM for the K = 256 projections of the Conformer layer:
//     out[32 rows, N] = epilogue( prologue(A)[32, 256] . W[N, 256]^T + bias )
// One workgroup (8 waves, 2 per SIMD) owns 32 rows and ALL N columns: the (optionally LayerNorm-ed) A
// tile is built once in LDS and stays resident while the workgroup walks over the column groups of 256;
// wave w owns columns [32w, 32w+32) of every group, so its B operand (32 weight rows) is private to
// it and streamed through a wave-private, double-buffered LDS region with a 4-deep register prefetch
// (same machinery as ffn_pc.hip, no workgroup barrier in the main loop).  M = B*T' = 7936 rows give
// 248 workgroups ~ one per CU.  v_mfma_f32_32x32x2_f32 (exact fp32) throughout.
//
// Prologues: plain rows | per-channel affine (Squeezeformer adaptive scale/bias, optional pad masking) |
//            LayerNorm(256) | LayerNorm into the conv module's padded time layout
//            (14 zero history rows per sequence, padded frames zeroed; conformer/convolution.py:98-108).
// Epilogues: store (fused QKV)            -- conformer/attention.py:53-79
//            residual + alpha*(.) [+ pad mask]   (attention out-proj, pointwise_conv2) -- encoder.py:123-145
//            GLU (pointwise_conv1: value rows c, gate rows 256+c) -- convolution.py:117-118
//            CTC greedy: online (max, argmax, sum exp) over all V columns, never writing logits --
//            loss/ctc.py:62-70 + ctc_greedy_decoder.py:20-21
//            CHAIN (offline conformer layer): attention out-projection + residual, then -- the 32 updated rows never leaving
//            the CU -- LayerNorm + pad mask + pointwise_conv1 + GLU in the same kernel (weights [Wo; W_pw1] concatenated,
//            one continuous slab stream): encoder.py:123-131 + convolution.py:98-119
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace masr {

static constexpr int RG_BM = 32;
static constexpr int RG_K = 256;
static constexpr int RG_ALD = RG_K + 4;
static constexpr int RG_WLD = 32 + 4;
static constexpr int RG_WSLAB = 32 * RG_WLD;
static constexpr int RG_NSET = 4;

__device__ __forceinline__ float rg_wsum(float v) {
    return wave_sum_dpp(v);
}

// GSPLIT = 1 (few row blocks, streaming chunk steps): blockIdx.y owns ONE column group of 256, so that e.g. the fused
// QKV projection of 16 streams x 16 frames runs on 8 x 3 workgroups with one MFMA tile (8 weight slabs) per wave
// PACKED = 1 (offline CHAIN and CTC launches): p.Wp holds the weights in the order the waves consume them
// ([tile][wave][slab j][group g][lane][4], pack_rows_pc_kernel) and every B fragment is ONE raw buffer load of 16 bytes per lane
// straight into operand registers -- descriptor in SGPRs, constant per-lane offset, wave-uniform scalar offset per fragment
// (the FFN kernels' scheme, ffn_pc.hip): same operand values in the same MFMA order as the slab pipeline, bit-identical.
template <int PRO, int EPI, int GSPLIT, int PACKED = 0>
__global__ __launch_bounds__(512) void rowgemm_kernel(RowGemmArgs p) {
    extern __shared__ __align__(16) float sm[];
    float* at = sm;                               // [32][260] A tile
    float* wpv = at + RG_BM * RG_ALD;             // [8][2][32][36] wave-private weight slabs
    float* red = wpv + 8 * 2 * RG_WSLAB;          // CTC: [8 waves][32 rows][3];  CHAIN: second A tile [32][260] (the updated x rows)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int row0 = blockIdx.x * RG_BM;          // first OUTPUT row of this workgroup
    float* wmine = wpv + wave * 2 * RG_WSLAB;

    // ---- A tile prologue: wave w prepares rows 4w..4w+3 -----------------------------------------------
    {
        f32x4 gw = f32x4{1.f, 1.f, 1.f, 1.f}, gb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PRO != RG_PRO_PLAIN) {
            gw = *reinterpret_cast<const f32x4*>(p.lnw + lane * 4);
            gb = *reinterpret_cast<const f32x4*>(p.lnb + lane * 4);
        }
        // issue all four row loads first (clamped addresses, no branches around memory ops), then normalise:
        // a load -> wait -> reduce chain per row would expose the global latency four times
        f32x4 v4[4];
        bool live4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = row0 + wave * 4 + rr;
            bool live = row < p.M;
            size_t src_row = live ? row : 0;
            if (PRO == RG_PRO_PLAIN && p.a_seq_t > 0) {       // rows live in a per-sequence padded buffer
                const int rc = live ? row : 0;
                const int b = rc / p.a_seq_t;
                src_row = (size_t)b * p.a_seq_stride + (rc - b * p.a_seq_t);
            }
            if (PRO == RG_PRO_AFFINE && p.lens && p.seq_t > 0) {
                // Squeezeformer conv module: padded frames are zeroed AFTER the adaptive scale/bias (convolution.py:109-115)
                const int rc = live ? row : 0;
                const int b = rc / p.seq_t, t = rc - b * p.seq_t;
                live = live && p.mstride * t < p.lens[b];
            }
            if (PRO == RG_PRO_LN_PAD) {
                // output row index lives in the padded layout [nseq][pad + Tq]; history rows and padded frames are zero
                const int per = p.seq_t + p.pad;
                const int rc = live ? row : 0;
                const int b = rc / per, tp = rc - b * per;
                const int t = max(tp - p.pad, 0);
                live = live && tp >= p.pad && !(p.lens && p.mstride * t >= p.lens[b]);
                src_row = (size_t)b * p.seq_t + t;
            }
            live4[rr] = live;
            v4[rr] = *reinterpret_cast<const f32x4*>(p.A + src_row * p.lda + lane * 4);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            const f32x4 v = v4[rr];
            f32x4 o;
            if (PRO == RG_PRO_PLAIN) {
                o = v;
            } else if (PRO == RG_PRO_AFFINE) {          // adaptive scale / bias (squeezeformer ada_scale, ada_bias)
                o[0] = gw[0] * v[0] + gb[0];
                o[1] = gw[1] * v[1] + gb[1];
                o[2] = gw[2] * v[2] + gb[2];
                o[3] = gw[3] * v[3] + gb[3];
            } else {
                const float mean = rg_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = rg_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + p.eps);
                o[0] = d0 * rstd * gw[0] + gb[0];
                o[1] = d1 * rstd * gw[1] + gb[1];
                o[2] = d2 * rstd * gw[2] + gb[2];
                o[3] = d3 * rstd * gw[3] + gb[3];
            }
            if (!live4[rr]) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&at[lr * RG_ALD + lane * 4]) = o;
        }
    }

    // ---- weight tile stream: tile t = 32 weight rows starting at wrow(t), 8 slabs of 32 k -------------
    // EPI_GLU: tiles come in (value, gate) pairs -> tile 2g = rows 32w.., tile 2g+1 = rows 256 + 32w..
    const int ngroups = (EPI == RG_EPI_GLU) ? 1 : (p.N + 255) / 256;
    const int ntiles = (EPI == RG_EPI_GLU) ? 2 : (EPI == RG_EPI_CHAIN) ? 3 : (GSPLIT ? 1 : ngroups);
    const int tbase = GSPLIT ? (int)blockIdx.y : 0;
    auto wrow_of = [&](int t) -> int {
        if (EPI == RG_EPI_GLU) return (t & 1) * 256 + wave * 32;
        return (tbase + t) * 256 + wave * 32;
    };
    const int lr8 = lane >> 3, lc4 = (lane & 7) * 4;
    f32x4 pre[RG_NSET][4];
    // Every epilogue except CTC has N a multiple of 256 (768 / 256 / 512 / 768: checked by the launcher), so all weight rows
    // exist and the address is a per-lane base (one pointer per 8-row piece i) + a wave-uniform tile offset + a constant: the
    // prefetch loads between the MFMAs then cost one add each instead of a clamp / multiply / 64-bit add chain.
    const float* wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wl[i] = p.W + (size_t)(wave * 32 + lr8 + 8 * i) * RG_K + lc4;
    auto src_of = [&](int t, int j, int i) -> const float* {
        const int tt = min(t, ntiles - 1);
        if (EPI == RG_EPI_CTC) {
            const int r = min(wrow_of(tt) + lr8 + 8 * i, p.N - 1);     // clamp: rows >= N are masked in the epilogue
            return p.W + (size_t)r * RG_K + j * 32 + lc4;
        }
        const int trow = (EPI == RG_EPI_GLU) ? (tt & 1) * 256 : (tbase + tt) * 256;     // wave-uniform
        return wl[i] + (size_t)trow * RG_K + j * 32;
    };
    auto dst_of = [&](int buf, int i) -> float* { return wmine + buf * RG_WSLAB + (lr8 + 8 * i) * RG_WLD + lc4; };
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(PACKED ? p.Wp : p.W), 0, PACKED ? ((p.N + 255) / 256) * 256 * RG_K * 4 : 0, 0x00020000);
    auto pld = [&](int t, int j, int g) -> f32x4 {     // fragment (slab j, group g) of this wave's tile t (clamped past the end)
        const unsigned fi = (unsigned)((((tbase + min(t, ntiles - 1)) * 8 + wave) * 8 + j) * 4 + g) * 256u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, lane16, fi * 4u, 0));
    };

    if (PACKED) {
#pragma unroll
        for (int k = 0; k < RG_NSET; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][g] = pld(0, k, g);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[0][i] = *reinterpret_cast<const f32x4*>(src_of(0, 0, i));
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dst_of(0, i)) = pre[0][i];
#pragma unroll
        for (int k = 1; k <= RG_NSET; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[k % RG_NSET][i] = *reinterpret_cast<const f32x4*>(src_of(k / 8, k & 7, i));
    }
    __syncthreads();                                  // A tile complete

    const float* aa = at + frow * RG_ALD + 4 * fh;     // (CHAIN: switched to the second tile after tile 0)
    const float* wfrag = wmine + frow * RG_WLD + 4 * fh;

    // CTC running state of this lane's row (row = lane & 31; the two half-waves hold different columns)
    float cm = -INFINITY, cs = 0.f;
    int ci = 0x7fffffff;

    f32x16 accv;      // value tile of the GLU pair
    // CHAIN: the residual rows of the out-projection are requested now and land under tile 0's MFMAs (R is not written by this
    // kernel; R2 is the output) instead of costing a memory round trip between tile 0 and the LayerNorm
    // (sequence, frame) of the block's first row in the mask's and in the output's row space: one division per block each
    const int cb0 = p.seq_t > 0 ? row0 / p.seq_t : 0, ct0 = p.seq_t > 0 ? row0 - cb0 * p.seq_t : 0;
    const int ob0 = p.out_seq_t > 0 ? row0 / p.out_seq_t : 0, ot0 = p.out_seq_t > 0 ? row0 - ob0 * p.out_seq_t : 0;
    float res0[16];
    f32x4 cgw = {0.f, 0.f, 0.f, 0.f}, cgb = cgw;
    float cbo = 0.f, cba = 0.f, cbg = 0.f;
    if (EPI == RG_EPI_CHAIN) {
        cgw = *reinterpret_cast<const f32x4*>(p.lnw + lane * 4);
        cgb = *reinterpret_cast<const f32x4*>(p.lnb + lane * 4);
        cbo = p.bias[wave * 32 + frow];
        cba = p.bias[256 + wave * 32 + frow];
        cbg = p.bias[512 + wave * 32 + frow];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * fh, p.M - 1);
            res0[r] = p.R[(size_t)row * p.ldr + wave * 32 + frow];
        }
    }
    for (int t = 0; t < ntiles; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wp = wfrag + (j & 1) * RG_WSLAB;          // slab index t*8 + j -> buffer j & 1
            f32x4 a[2], b[2];
            a[0] = *reinterpret_cast<const f32x4*>(aa + j * 32);
            if (!PACKED) b[0] = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
                    a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(aa + j * 32 + 8 * (g + 1));
                    if (!PACKED) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + 8 * (g + 1));
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // EPI_CTC accumulates the TRANSPOSED tile (A operand = weight rows, B operand = activation
                    // rows): each lane then owns ONE output row and 16 of the tile's 32 columns, so the row-wise
                    // softmax statistics need no cross-lane butterfly (only one lane^32 exchange per tile).
                    if (EPI == RG_EPI_CTC) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(PACKED ? pre[j % RG_NSET][g][q] : b[g & 1][q], a[g & 1][q], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], PACKED ? pre[j % RG_NSET][g][q] : b[g & 1][q], acc, 0, 0, 0);
                    const int slot = g * 4 + q;
                    // slots 0,2,4,6: store piece of slab s+1 (set (j+1)%NSET) into buffer (j+1)&1;
                    // slots 8..14: refill that set with slab s+1+NSET
                    if (PACKED) {
                        if (q == 3) pre[j % RG_NSET][g] = pld(t + (j + RG_NSET) / 8, (j + RG_NSET) & 7, g);
                    } else if (slot < 8) {
                        if ((slot & 1) == 0)
                            *reinterpret_cast<f32x4*>(dst_of((j + 1) & 1, slot >> 1)) = pre[(j + 1) % RG_NSET][slot >> 1];
                    } else if ((slot & 1) == 0) {
                        pre[(j + 1) % RG_NSET][(slot - 8) >> 1] = *reinterpret_cast<const f32x4*>(
                            src_of(t + (j + 1 + RG_NSET) / 8, (j + 1 + RG_NSET) & 7, (slot - 8) >> 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the written MFMA / load / LDS-store interleave
                }
            }
        }

        // ---- per-tile epilogue (C layout: col = lane&31, row = (r&3) + 8(r>>2) + 4*fh) ---------------------
        const int col = wrow_of(t) + frow;
        if (EPI == RG_EPI_STORE || EPI == RG_EPI_RESID) {
            if (col < p.N) {
                const float bv = p.bias ? p.bias[col] : 0.f;
                // all residual loads are issued before the first store (R may alias C, so the compiler cannot
                // hoist them itself: 16 load -> wait -> store round trips otherwise)
                float res[16];
                if (EPI == RG_EPI_RESID) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * fh, p.M - 1);
                        res[r] = p.R[(size_t)row * p.ldr + col];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    float v = acc[r] + bv;
                    if (EPI == RG_EPI_RESID) {
                        if (p.mask_tp > 0) {
                            const int rc = min(row, p.M - 1);
                            const int b = rc / p.mask_tp, tt = rc - b * p.mask_tp;
                            if (p.mstride * tt >= p.lens[b]) v = 0.f;
                        }
                        v = res[r] + p.alpha * v;
                    }
                    if (row < p.M) {
                        size_t crow = row;
                        int ccol = col;
                        float* cb = p.C;
                        if (EPI == RG_EPI_STORE) {
                            if (p.out_seq_t > 0) {
                                const int b = row / p.out_seq_t, t = row - b * p.out_seq_t;
                                crow = (size_t)b * (p.out_seq_t + p.out_pad_tot) + p.out_pad_l + t;
                            }
                            if (p.plane_cols > 0) {
                                cb += (size_t)(col / p.plane_cols) * p.plane_stride;
                                ccol = col % p.plane_cols;
                            }
                        }
                        cb[crow * p.ldc + ccol] = v;
                    }
                }
            }
        } else if (EPI == RG_EPI_CHAIN) {
            const int ch = wave * 32 + frow;
            if (t == 0) {
                // x <- x + (att . Wo^T + bo): to global (residual base of pointwise_conv2) and, raw, into the second LDS tile
                const float bv = cbo;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = res0[r] + (acc[r] + bv);
                    if (row0 + lr < p.M) p.R2[(size_t)(row0 + lr) * p.ldr + ch] = v;
                    red[lr * RG_ALD + ch] = v;
                }
                __syncthreads();
                {   // LayerNorm (conv module) + pad mask in place: wave w owns rows 4w .. 4w+3
                    const f32x4 gw = cgw, gb = cgb;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int lr = wave * 4 + rr, row = row0 + lr;
                        bool live = row < p.M;
                        if (live && p.lens && p.seq_t > 0) {
                            const SeqRow q = seq_row(cb0, ct0, p.seq_t, lr);
                            live = p.mstride * q.t < p.lens[q.b];
                        }
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&red[lr * RG_ALD + lane * 4]);
                        const float mean = rg_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                        const float var = rg_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                        const float rstd = 1.0f / sqrtf(var + p.eps);
                        f32x4 o;
                        o[0] = d0 * rstd * gw[0] + gb[0];
                        o[1] = d1 * rstd * gw[1] + gb[1];
                        o[2] = d2 * rstd * gw[2] + gb[2];
                        o[3] = d3 * rstd * gw[3] + gb[3];
                        if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(&red[lr * RG_ALD + lane * 4]) = o;
                    }
                }
                __syncthreads();
                aa = red + frow * RG_ALD + 4 * fh;
            } else if (t == 1) {
                accv = acc;
            } else {
                const float bva = cba, bvg = cbg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int row = row0 + lr;
                    if (row >= p.M) continue;
                    size_t crow = row;
                    if (p.out_seq_t > 0) {
                        const SeqRow q = seq_row(ob0, ot0, p.out_seq_t, lr);
                        crow = (size_t)q.b * (p.out_seq_t + p.out_pad_tot) + p.out_pad_l + q.t;
                    }
                    const float g = acc[r] + bvg;
                    p.C[crow * p.ldc + ch] = (accv[r] + bva) * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
                }
            }
        } else if (EPI == RG_EPI_GLU) {
            if (t == 0) {
                accv = acc;
            } else {
                const int ch = wave * 32 + frow;
                const float bva = p.bias[ch], bvg = p.bias[256 + ch];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int row = row0 + lr;
                    if (row >= p.M) continue;
                    size_t crow = row;
                    if (p.out_seq_t > 0) {               // symmetric-padded layout for the non-causal depthwise conv
                        const SeqRow q = seq_row(ob0, ot0, p.out_seq_t, lr);
                        crow = (size_t)q.b * (p.out_seq_t + p.out_pad_tot) + p.out_pad_l + q.t;
                    }
                    const float g = acc[r] + bvg;
                    p.C[crow * p.ldc + ch] = (accv[r] + bva) * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
                }
            }
        } else {   // RG_EPI_CTC: fold this 32-column tile into the running (max, argmax, sum exp) of my row
            // transposed C layout: lane&31 = output row, column-in-tile = (r&3) + 8(r>>2) + 4*fh
            float v[16];
            float m = -INFINITY;
            int mi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = wrow_of(t) + (r & 3) + 8 * (r >> 2) + 4 * fh;
                v[r] = c < p.N ? acc[r] + p.bias[min(c, p.N - 1)] : -INFINITY;
                if (v[r] > m) { m = v[r]; mi = c; }              // ascending c: first maximum wins
            }
            {
                const float om = __shfl_xor(m, 32, 64);
                const int oi = __shfl_xor(mi, 32, 64);
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
            }
            const float mnew = fmaxf(cm, m);
            const float mref = (mnew == -INFINITY) ? 0.f : mnew;   // stripe without a valid column yet
            float e = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) e += __expf(v[r] - mref);  // -inf -> 0
            e += __shfl_xor(e, 32, 64);
            cs = cs * __expf(cm - mref) + e;
            if (m > cm || (m == cm && mi < ci)) ci = mi;
            cm = mnew;
        }
    }

    if (EPI == RG_EPI_CTC) {
        // combine the 8 waves (column stripes) per row
        if (fh == 0) {
            float* d = red + (wave * RG_BM + frow) * 3;
            d[0] = cm;
            d[1] = cs;
            d[2] = __int_as_float(ci);
        }
        __syncthreads();
        if (tid < RG_BM && row0 + tid < p.M) {
            float M = -INFINITY;
            int I = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float* d = red + (w * RG_BM + tid) * 3;
                const int wi = __float_as_int(d[2]);
                if (d[0] > M || (d[0] == M && wi < I)) { M = d[0]; I = wi; }
            }
            float S = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float* d = red + (w * RG_BM + tid) * 3;
                S += d[1] * __expf(d[0] - M);
            }
            p.out_idx[row0 + tid] = I;
            p.out_maxp[row0 + tid] = 1.0f / S;       // softmax probability of the argmax = exp(0) / sum
        }
    }
}

template <int PRO, int EPI>
static void launch_rg(const RowGemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(RG_BM * RG_ALD + 8 * 2 * RG_WSLAB + (EPI == RG_EPI_CHAIN ? RG_BM * RG_ALD : 8 * RG_BM * 3)) * sizeof(float);
    constexpr bool can_split = EPI == RG_EPI_STORE || EPI == RG_EPI_RESID;
    static LdsAttr attr0, attr1;
    ensure_dynamic_lds(reinterpret_cast<const void*>(rowgemm_kernel<PRO, EPI, 0>), lds, attr0);
    if (can_split) ensure_dynamic_lds(reinterpret_cast<const void*>(rowgemm_kernel<PRO, EPI, can_split ? 1 : 0>), lds, attr1);
    const int rowblocks = (a.M + RG_BM - 1) / RG_BM, ngroups = (a.N + 255) / 256;
    if (EPI != RG_EPI_CTC && a.N % 256 != 0) {
        fprintf(stderr, "rowgemm: N must be a multiple of 256 for this epilogue\n");
        abort();
    }
    constexpr bool can_pack = true;      // every epilogue of the row-block kernel (the column-group split of few row blocks keeps the slabs)
    if (can_split && ngroups > 1 && rowblocks < 64) {
        hipLaunchKernelGGL((rowgemm_kernel<PRO, EPI, can_split ? 1 : 0>), dim3(rowblocks, ngroups), dim3(512), lds, s, a);
    } else if (can_pack && a.Wp) {
        static LdsAttr attrp;
        ensure_dynamic_lds(reinterpret_cast<const void*>(rowgemm_kernel<PRO, EPI, 0, can_pack ? 1 : 0>), lds, attrp);
        hipLaunchKernelGGL((rowgemm_kernel<PRO, EPI, 0, can_pack ? 1 : 0>), dim3(rowblocks), dim3(512), lds, s, a);
    } else {
        hipLaunchKernelGGL((rowgemm_kernel<PRO, EPI, 0>), dim3(rowblocks), dim3(512), lds, s, a);
    }
}

static void launch_rowgemm_big(const RowGemmArgs& a, int pro, int epi, hipStream_t s) {
    if (pro == RG_PRO_LN && epi == RG_EPI_STORE) launch_rg<RG_PRO_LN, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_STORE) launch_rg<RG_PRO_PLAIN, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_RESID) launch_rg<RG_PRO_PLAIN, RG_EPI_RESID>(a, s);
    else if (pro == RG_PRO_LN_PAD && epi == RG_EPI_GLU) launch_rg<RG_PRO_LN_PAD, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_GLU) launch_rg<RG_PRO_PLAIN, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_AFFINE && epi == RG_EPI_STORE) launch_rg<RG_PRO_AFFINE, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_AFFINE && epi == RG_EPI_GLU) launch_rg<RG_PRO_AFFINE, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_CTC) launch_rg<RG_PRO_PLAIN, RG_EPI_CTC>(a, s);
    else if (pro == RG_PRO_LN && epi == RG_EPI_CTC) launch_rg<RG_PRO_LN, RG_EPI_CTC>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_CHAIN) launch_rg<RG_PRO_PLAIN, RG_EPI_CHAIN>(a, s);
}

static int g_small = 1;
void set_rowgemm_small(int on) { g_small = on; }

// false: nothing was launched (the fused streaming prologues HIST / DWCONV exist in the small-M kernel only; the caller then
// runs the unfused sequence)
bool launch_rowgemm(const RowGemmArgs& a, int pro, int epi, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return true;
    // few row blocks (streaming chunk steps, short utterances): K-split kernel with 4x more, 4x shorter workgroups
    if (g_small && launch_rowgemm_small(a, pro, epi, s)) return true;
    if (pro == RG_PRO_HIST || pro == RG_PRO_DWCONV) return false;
    launch_rowgemm_big(a, pro, epi, s);
    // the big kernel stores all QKV columns to C; the streams' cache append is then its own launch
    if (a.kv_seqs && epi == RG_EPI_STORE) launch_kv_append(a.kv_seqs, a.C, a.M / a.kv_tq, a.kv_tq, s);
    return true;
}

}  // namespace masr
