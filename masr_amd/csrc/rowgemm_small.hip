// Small-M variant of the K = 256 row-block GEMM (rowgemm.hip) for streaming chunk steps, where M = n_streams * 16 rows gives
// only M/32 = 8..15 row blocks: a 32-row x N-column workgroup with one 32x32 tile per wave then runs 128 dependent MFMAs
// (~4 us) on a handful of the 256 CUs and its latency IS the kernel time (13-21 us per projection, 48 projections per
// chunk step).  Here the work of one row block is cut into more, shorter pieces:
//     workgroup = 32 rows x 64 columns;  wave (ct, kq) = column tile ct (32 columns) x K quarter kq (64 of the 256 k)
//     -> 32 MFMAs per wave, 4x more workgroups, no LDS staging: every lane loads its own A / W fragments straight
//        from global memory (all loads issued up front, one latency exposure), the four K-quarter partial tiles are
//        summed through LDS in a fixed order, and each wave finishes 4 of the 16 accumulator registers.
// Prologues: plain rows (optionally from a per-sequence padded buffer) | LayerNorm(256) | per-channel affine |
//            streaming conv module, pointwise_conv1 side (HIST): row (i, tp) of the padded layout = cnn-cache row tp of stream i
//            or LayerNorm of the new frame tp - pad; the column-block-0 workgroups also write the new cache (replaces the
//            conv_hist launch and the lnpad buffer) |
//            streaming conv module, pointwise_conv2 side (DWCONV): row = SiLU(LayerNorm(causal depthwise conv of the GLU
//            output)), computed for the 32 rows in the prologue (replaces the dwconv_ln_silu launch).
//            The last two stage the A tile in LDS (each wave prepares 4 rows); every column block repeats that work, which
//            is ~100 kFMA against a saved kernel launch (~7 us on the dependent chain of a chunk step).
// Epilogues: store (fused QKV; optionally the K|V columns go straight to the streams' key/value caches, which replaces
//            the separate append launch) | residual + alpha * (.) | GLU (value tile ct = 0, gate tile ct = 1).
// Same arithmetic as rowgemm.hip (v_mfma_f32_32x32x2_f32, fp32 throughout); only the order of the K summation differs.
// References: conformer/attention.py:53-79 (QKV), encoder.py:123-145 (out-projection / residual),
//             convolution.py:117-119,128 (pointwise_conv1 + GLU, pointwise_conv2), encoder.py:404-419 (cache append).
#include <algorithm>

#include "common.h"

namespace masr {

__device__ __forceinline__ float rs_wsum(float v) {
    return wave_sum_dpp(v);
}

template <int PRO, int EPI>
__global__ __launch_bounds__(512) void rowgemm_small_kernel(RowGemmArgs p) {
    constexpr bool STAGED = PRO == RG_PRO_HIST || PRO == RG_PRO_DWCONV;
    constexpr int ALD = 256 + 4;
    __shared__ __align__(16) float xch[4 * 2 * 16 * 64];     // [kq][ct][acc register][lane]
    __shared__ float stat[32 * 2];                            // LayerNorm mean / rstd of the 32 rows
    __shared__ __align__(16) float at[STAGED ? 32 * ALD : 4]; // staged A tile (HIST / DWCONV prologues)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, kq = wave >> 1;
    const int frow = lane & 31, fh = lane >> 5;
    // blockIdx.x = column block (fastest varying): consecutive workgroups go to consecutive XCDs, so all row blocks of one
    // column block share an XCD and its weight rows are fetched from HBM once, not once per XCD (L2 is per XCD)
    const int row0 = blockIdx.y * 32;
    // weight rows of my tile: STORE / RESID: 64 consecutive output columns per workgroup; GLU: 32 channels (value | gate)
    const int col0 = (EPI == RG_EPI_GLU) ? (int)blockIdx.x * 32 + ct * 256 : (int)blockIdx.x * 64 + ct * 32;
    const int kbase = kq * 64 + 4 * fh;

    // ---- issue every global load of the kernel -----------------------------------------------------------------
    const int arow = min(row0 + frow, p.M - 1);
    size_t asrc = arow;
    if (PRO == RG_PRO_PLAIN && p.a_seq_t > 0) {
        const int b = arow / p.a_seq_t;
        asrc = (size_t)b * p.a_seq_stride + (arow - b * p.a_seq_t);
    }
    const float* ap = p.A + asrc * p.lda + kbase;
    const float* wp = p.W + (size_t)(col0 + frow) * 256 + kbase;
    f32x4 a[8], b[8];
    // rows for the LayerNorm statistics first: vmcnt retires in order, so their wait leaves the later loads in flight
    f32x4 srow[4];
    constexpr bool LNORM = PRO == RG_PRO_LN || PRO == RG_PRO_LN_PAD;     // LN_PAD (offline pointwise_conv1, pad == 0): LN + pad mask
    if (LNORM) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = min(row0 + wave * 4 + rr, p.M - 1);
            srow[rr] = *reinterpret_cast<const f32x4*>(p.A + (size_t)row * p.lda + lane * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep them first (the scheduler otherwise mixes them into the later loads)
    if (!STAGED) {
#pragma unroll
        for (int g = 0; g < 8; ++g) a[g] = *reinterpret_cast<const f32x4*>(ap + 8 * g);
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) b[g] = *reinterpret_cast<const f32x4*>(wp + 8 * g);
    f32x4 gw[8], gb[8];
    if (LNORM || PRO == RG_PRO_AFFINE) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            gw[g] = *reinterpret_cast<const f32x4*>(p.lnw + kbase + 8 * g);
            gb[g] = *reinterpret_cast<const f32x4*>(p.lnb + kbase + 8 * g);
        }
    }
    // epilogue operands of the 4 (GLU: 2) accumulator registers this wave finishes
    constexpr int NFIN = (EPI == RG_EPI_GLU) ? 2 : 4;
    const int r0 = (EPI == RG_EPI_GLU) ? 4 * kq + 2 * ct : 4 * kq;
    const int ocol = (EPI == RG_EPI_GLU) ? (int)blockIdx.x * 32 + frow : col0 + frow;      // GLU: channel
    float res[NFIN];
    if (EPI == RG_EPI_RESID) {
#pragma unroll
        for (int i = 0; i < NFIN; ++i) {
            const int r = r0 + i;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * fh, p.M - 1);
            res[i] = p.R[(size_t)row * p.ldr + ocol];
        }
    }
    // fused cache append: destination rows of my NFIN output rows (pointer + counters fetched with everything else; a load in the
    // epilogue would be a second, serial memory round trip per register)
    const float* kvk[NFIN];      // loads only up here: arithmetic on the loaded values would force a wait per register
    int kvnq[NFIN], kvnk[NFIN], kvt[NFIN];
    const bool to_cache = EPI == RG_EPI_STORE && p.kv_seqs != nullptr && ocol >= 256;
    if (EPI == RG_EPI_STORE && p.kv_seqs != nullptr) {
#pragma unroll
        for (int i = 0; i < NFIN; ++i) {
            const int r = r0 + i;
            const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * fh, p.M - 1);
            const int bq = row / p.kv_tq;
            kvt[i] = row - bq * p.kv_tq;
            const AttSeq* sq = p.kv_seqs + bq;
            kvk[i] = sq->k;
            kvnq[i] = sq->nq;
            kvnk[i] = sq->nk;
        }
    }
    // LN_PAD: frames behind an utterance's own length enter the convolution as zeros (convolution.py:91-92 masked_fill)
    int a_len = 0x7fffffff, a_t = 0;
    if (PRO == RG_PRO_LN_PAD && p.lens) {
        const int bq = arow / p.seq_t;
        a_t = arow - bq * p.seq_t;
        a_len = p.lens[bq];
    }
    float bv = 0.f, bg = 0.f;
    if (EPI == RG_EPI_GLU) {
        bv = p.bias[ocol];
        bg = p.bias[256 + ocol];
    } else if (p.bias) {
        bv = p.bias[ocol];
    }
    __builtin_amdgcn_sched_barrier(0);      // all loads above are in flight before anything waits (hipcc otherwise sinks them
                                            // next to their use: eight exposed latencies instead of one)

    // ---- staged prologues: wave w prepares rows 4w .. 4w+3 of the A tile in LDS ----------------------------------------------
    if (PRO == RG_PRO_HIST) {
        // padded row (i, tp), per = seq_t + pad rows per stream: tp < pad -> cnn cache row tp (already normalised when it was
        // new), else LayerNorm(x[i * seq_t + tp - pad])  (convolution.py:98-108; the cache keeps the last pad rows)
        const int per = p.seq_t + p.pad;
        const f32x4 lw = *reinterpret_cast<const f32x4*>(p.lnw + lane * 4);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(p.lnb + lane * 4);
        f32x4 v4[4];
        int tp4[4], i4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = min(row0 + wave * 4 + rr, p.M - 1);
            i4[rr] = row / per;
            tp4[rr] = row - i4[rr] * per;
            const float* src = tp4[rr] < p.pad ? p.cache_rd[i4[rr]] + (size_t)tp4[rr] * 256
                                               : p.A + ((size_t)i4[rr] * p.seq_t + tp4[rr] - p.pad) * 256;
            v4[rr] = *reinterpret_cast<const f32x4*>(src + lane * 4);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            f32x4 o = v4[rr];
            if (tp4[rr] >= p.pad && p.hist_affine) {          // Squeezeformer: adaptive scale / bias instead of the LayerNorm
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = lw[k] * o[k] + lb[k];
            } else if (tp4[rr] >= p.pad) {
                const f32x4 v = v4[rr];
                const float mean = rs_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = rs_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + p.eps);
                o[0] = d0 * rstd * lw[0] + lb[0];
                o[1] = d1 * rstd * lw[1] + lb[1];
                o[2] = d2 * rstd * lw[2] + lb[2];
                o[3] = d3 * rstd * lw[3] + lb[3];
            }
            const bool live = row0 + lr < p.M;
            if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&at[lr * ALD + lane * 4]) = o;
            // new cache = last pad rows of [cache | new rows] -- written to the OTHER half of the double buffer, once
            if (live && blockIdx.x == 0 && tp4[rr] >= p.seq_t)
                *reinterpret_cast<f32x4*>(p.cache_wr[i4[rr]] + (size_t)(tp4[rr] - p.seq_t) * 256 + lane * 4) = o;
        }
        __syncthreads();
    } else if (PRO == RG_PRO_DWCONV) {
        // row (i, t): out[c] = SiLU(LayerNorm_c(b[c] + sum_j w[j][c] * gpad[i][t + j][c])), gpad = [pad history | seq_t new] GLU
        // rows (convolution.py:120-127).  seq_t % 4 == 0 (checked by the launcher): the 4 rows of a wave share one stream and a
        // window of KT + 3 input rows.  Same arithmetic order as dwconv_ln_silu_kernel.
        constexpr int KMAX = 15;
        const int KT = p.pad + 1;
        const int row = min(row0 + wave * 4, p.M - 4);
        const int i = row / p.seq_t, t = row - i * p.seq_t;
        const float* gin = p.A + ((size_t)i * (p.pad + p.seq_t) + t) * 256 + lane * 4;
        f32x4 win[KMAX + 3], w[KMAX];
        f32x4 gc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.gconst) gc = *reinterpret_cast<const f32x4*>(p.gconst + lane * 4);
        // every window row is requested unconditionally, all loads in flight before the first use (a branch around a load makes
        // hipcc wait for it on the spot: the chunk steps' launch went 11.4 -> 17 us with the select written as an if / else)
#pragma unroll
        for (int j = 0; j < KMAX + 3; ++j)
            if (j < KT + 3) win[j] = *reinterpret_cast<const f32x4*>(gin + (size_t)j * 256);
        if (p.gconst) {
            // offline causal conv: padded rows t + j < pad are the unmaterialised history = the constant glu(bias) row (the
            // buffer rows exist, whatever they hold is replaced)
#pragma unroll
            for (int j = 0; j < KMAX + 3; ++j)
                if (j < KT + 3 && t + j < p.pad) win[j] = gc;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < KT) w[j] = *reinterpret_cast<const f32x4*>(p.dw_w + j * 256 + lane * 4);
        const f32x4 cb = *reinterpret_cast<const f32x4*>(p.dw_b + lane * 4);
        const f32x4 lw = *reinterpret_cast<const f32x4*>(p.lnw + lane * 4);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(p.lnb + lane * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            f32x4 v = cb;
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < KT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaf(w[j][k], win[rr + j][k], v[k]);
                }
            const float mean = rs_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = rs_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + p.eps);
            f32x4 o;
            o[0] = d0 * rstd * lw[0] + lb[0];
            o[1] = d1 * rstd * lw[1] + lb[1];
            o[2] = d2 * rstd * lw[2] + lb[2];
            o[3] = d3 * rstd * lw[3] + lb[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] / (1.0f + expf(-o[k]));
            const int lr = row - row0 + rr;                                 // (the clamped last group re-writes valid rows)
            if (lr >= 0) *reinterpret_cast<f32x4*>(&at[lr * ALD + lane * 4]) = o;
        }
        __syncthreads();
    }
    if (STAGED) {
#pragma unroll
        for (int g = 0; g < 8; ++g) a[g] = *reinterpret_cast<const f32x4*>(&at[frow * ALD + kbase + 8 * g]);
    }

    // ---- prologue on the A fragments (registers) ------------------------------------------------------------------
    if (LNORM) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const f32x4 v = srow[rr];
            const float mean = rs_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = rs_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            if (lane == 0) {
                stat[(wave * 4 + rr) * 2] = mean;
                stat[(wave * 4 + rr) * 2 + 1] = 1.0f / sqrtf(var + p.eps);
            }
        }
        __syncthreads();
        const float mean = stat[frow * 2], rstd = stat[frow * 2 + 1];
        const bool dead = PRO == RG_PRO_LN_PAD && p.mstride * a_t >= a_len;
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[g][q] = dead ? 0.f : (a[g][q] - mean) * rstd * gw[g][q] + gb[g][q];
    } else if (PRO == RG_PRO_AFFINE) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[g][q] = gw[g][q] * a[g][q] + gb[g][q];
    }

    // ---- 32 MFMAs: my 32x32 tile over my K quarter ------------------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g][q], b[g][q], acc, 0, 0, 0);

    // ---- K-quarter reduction through LDS (ascending kq: deterministic) ------------------------------------------------
    float* mine = xch + ((kq * 2 + ct) * 16) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
    __syncthreads();

    if (EPI == RG_EPI_GLU) {
#pragma unroll
        for (int i = 0; i < NFIN; ++i) {
            const int r = r0 + i;
            float sv = 0.f, sg = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sv += xch[((k * 2 + 0) * 16 + r) * 64 + lane];
                sg += xch[((k * 2 + 1) * 16 + r) * 64 + lane];
            }
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (row >= p.M) continue;
            size_t crow = row;
            if (p.out_seq_t > 0) {
                const int bq = row / p.out_seq_t, t = row - bq * p.out_seq_t;
                crow = (size_t)bq * (p.out_seq_t + p.out_pad_tot) + p.out_pad_l + t;
            }
            const float gt = sg + bg;
            p.C[crow * p.ldc + ocol] = (sv + bv) * __builtin_amdgcn_rcpf(1.0f + __expf(-gt));
        }
    } else {
#pragma unroll
        for (int i = 0; i < NFIN; ++i) {
            const int r = r0 + i;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += xch[((k * 2 + ct) * 16 + r) * 64 + lane];
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (row >= p.M) continue;
            float v = sum + bv;
            if (EPI == RG_EPI_RESID) {
                if (p.mask_tp > 0) {
                    const int bq = row / p.mask_tp, tt = row - bq * p.mask_tp;
                    if (p.mstride * tt >= p.lens[bq]) v = 0.f;
                }
                p.C[(size_t)row * p.ldc + ocol] = res[i] + p.alpha * v;
            } else {
                if (to_cache) {
                    // k | v columns of the fused QKV projection -> the stream's cache rows nk - nq .. nk - 1 ([k(256) | v(256)] per row)
                    const_cast<float*>(kvk[i])[(size_t)(kvnk[i] - kvnq[i] + kvt[i]) * 512 + (ocol - 256)] = v;
                    continue;
                }
                size_t crow = row;
                int ccol = ocol;
                float* cb = p.C;
                if (p.out_seq_t > 0) {
                    const int bq = row / p.out_seq_t, t = row - bq * p.out_seq_t;
                    crow = (size_t)bq * (p.out_seq_t + p.out_pad_tot) + p.out_pad_l + t;
                }
                if (p.plane_cols > 0) {
                    cb += (size_t)(ocol / p.plane_cols) * p.plane_stride;
                    ccol = ocol % p.plane_cols;
                }
                cb[crow * p.ldc + ccol] = v;
            }
        }
    }
}

template <int PRO, int EPI>
static void launch_rs(const RowGemmArgs& a, hipStream_t s) {
    const int rowblocks = (a.M + 31) / 32;
    const int ny = (EPI == RG_EPI_GLU) ? 8 : a.N / 64;
    hipLaunchKernelGGL((rowgemm_small_kernel<PRO, EPI>), dim3(ny, rowblocks), dim3(512), 0, s, a);
}

// true when the small-M kernel took the launch
// row blocks (of 32 rows) below which the K-split kernel takes the projection: 4x more, 4x shorter workgroups fill the chip
// where a 32-row x N workgroup per row block leaves CUs idle (128 lock-step streams = 64 row blocks: chunk call 5.75 -> 5.34 ms;
// tools/chunk_step_ab.py).  At 124 row blocks (16 x 10 s; the Efficient Conformer's half-rate layers at 32 x 10 s) the row-block
// kernel is ahead again: 4.56 -> 4.42 ms per forward, 6.11 -> 6.04 ms per Efficient-Conformer pass (tools/offline_size_ab.py,
// tools/efficient_size_ab.py); 93 row blocks are indifferent
static int g_small_blocks = 112;
void set_rowgemm_small_blocks(int n) { g_small_blocks = n; }
int rowgemm_small_blocks() { return g_small_blocks; }
bool launch_rowgemm_small(const RowGemmArgs& a, int pro, int epi, hipStream_t s) {
    // (the streaming conv module's fused prologues exist only here: where this kernel declines, the caller runs an extra launch in
    //  front of the row-block kernel -- 128 streams = 120 padded row blocks of pointwise_conv1: 3.26 -> 3.18 ms per chunk call with
    //  the limit at 160 for those two prologues, tools/chunk_lat.py MASR_AB=12:112,12:128)
    const int limit = (pro == RG_PRO_HIST || pro == RG_PRO_DWCONV) ? std::max(g_small_blocks, 160) : g_small_blocks;
    if (a.M <= 0 || a.M >= limit * 32) return false;
    if (epi == RG_EPI_GLU ? a.N != 512 : (a.N % 64) != 0) return false;
    if (pro == RG_PRO_AFFINE && a.lens && a.seq_t > 0) return false;       // pad masking in the prologue: big kernel only
    if (pro == RG_PRO_LN && epi == RG_EPI_STORE) launch_rs<RG_PRO_LN, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_AFFINE && epi == RG_EPI_STORE) launch_rs<RG_PRO_AFFINE, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_STORE) launch_rs<RG_PRO_PLAIN, RG_EPI_STORE>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_RESID) launch_rs<RG_PRO_PLAIN, RG_EPI_RESID>(a, s);
    else if (pro == RG_PRO_PLAIN && epi == RG_EPI_GLU) launch_rs<RG_PRO_PLAIN, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_AFFINE && epi == RG_EPI_GLU) launch_rs<RG_PRO_AFFINE, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_LN_PAD && epi == RG_EPI_GLU) {
        if (a.pad != 0 || a.seq_t <= 0) return false;      // (history rows through the GEMM: row-block kernel only)
        launch_rs<RG_PRO_LN_PAD, RG_EPI_GLU>(a, s);
    }
    else if (pro == RG_PRO_HIST && epi == RG_EPI_GLU) launch_rs<RG_PRO_HIST, RG_EPI_GLU>(a, s);
    else if (pro == RG_PRO_DWCONV && epi == RG_EPI_RESID) {
        // groups of 4 rows share one sequence's window: frames per sequence a multiple of 4 -- or ONE sequence (its last group is
        // clamped to the last 4 rows and re-writes valid rows)
        if (a.pad + 1 > 15 || a.M < 4) return false;
        if ((a.seq_t % 4 || a.M % 4) && a.M != a.seq_t) return false;
        launch_rs<RG_PRO_DWCONV, RG_EPI_RESID>(a, s);
    }
    else return false;
    return true;
}

}  // namespace masr
