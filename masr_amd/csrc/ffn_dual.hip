// Fused position-wise feed-forward block, producer / consumer waves with TWO INDEPENDENT ACCUMULATOR CHAINS per wave
// (reference conformer/positionwise.py:30-37, conformer/encoder.py:113-121,150-158; same block as ffn_pc.hip):
//     x <- x + scale * ( W2 . silu( W1 . LayerNorm(x) + b1 ) + b2 )
// ffn_pc.hip gives every wave ONE 32x32 accumulator at a time: 128 (producer) or 16 (consumer, per slab) MFMAs in a row that
// each wait for the one before, with the weight loads, the SiLU epilogue and the LDS stores of the side work sitting between
// dependent instructions.  Whenever a wave is alone on its SIMD's matrix pipe (the other role has reached the phase barrier,
// and in the fill / drain phases) every such slot is exposed.  Here d_ff is consumed in chunks of 256 hidden units:
//   producer p (waves 0..3): hidden tiles h[32 rows, 64 units] = two 32x32 accumulators that share every A fragment
//   consumer c (waves 4..7): output tiles acc2[32 rows, 64 cols], both accumulators advanced MFMA by MFMA over K = 256
// so consecutive MFMAs of a wave never depend on each other, each A fragment read from LDS feeds two MFMAs instead of one,
// and there are 8 + 2 phase barriers instead of 16 + 2.  Every accumulator still sums its k in ascending order: the results are
// BIT-IDENTICAL to ffn_pc.hip (tests/test_gpu_ffn_packed.py).  Weights: packed copies in the order the waves consume them
// ([chunk][idx][k-slab j][tile n][group g][lane][4], pack_ffn_dual_kernel), one coalesced 16-byte-per-lane load per four MFMAs
// straight into operand registers, ring of two k-slabs (64 MFMAs ahead).
// TAIL (LN + fused QKV projection on the finished rows): the three 256-column tiles of a wave advance together (three chains).
// HEADK (depthwise conv + LN + SiLU + pointwise_conv2 + residual in front of the block): as in ffn_pc.hip, one tile per wave.
#include "common.h"

namespace masr {

static constexpr int DU_BM = 32;
static constexpr int DU_D = 256;
static constexpr int DU_CH = 256;          // hidden units per chunk
static constexpr int DU_XLD = DU_D + 4;    // 260
static constexpr int DU_HLD = DU_CH + 4;   // 260

__device__ __forceinline__ float du_wsum(float v) { return wave_sum_dpp(v); }
// packed fragments through raw buffer loads: descriptor in SGPRs, constant per-lane offset, wave-uniform scalar offset per fragment
__device__ __forceinline__ f32x4 du_bufld(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, float_index * 4u, 0));
}

template <int AFFINE, int TAIL, int HEADK>
__global__ __launch_bounds__(512) void ffn_dual_kernel(float* x, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       const float* __restrict__ p1, const float* __restrict__ b1,
                                                       const float* __restrict__ p2, const float* __restrict__ b2, int M,
                                                       int dff, float eps, float scale, FfnTail tail, FfnHead head) {
    extern __shared__ __align__(16) float sm[];
    float* xn = sm;                              // [32][260]    LayerNorm(x) tile (A operand of GEMM1)
    float* hs = xn + DU_BM * DU_XLD;             // [2][32][260] hidden tile (A operand of GEMM2), double-buffered

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, idx = wave & 3;  // waves w and w+4 share a SIMD: producer idx and consumer idx
    const int row0 = blockIdx.x * DU_BM;
    const int frow = lane & 31, fh = lane >> 5;
    const unsigned lane16 = lane * 16;

    f32x4 gw_head = {0.f, 0.f, 0.f, 0.f}, gb_head = gw_head;
    if (HEADK > 0) {
        // ---- head 1: depthwise conv of my 32 rows -> xn tile (thread = (channel, 16-row half), sliding window) ----------------
        {
            constexpr int KT = HEADK > 0 ? HEADK : 1, pad = KT - 1;
            const int c = tid & 255, half = tid >> 8;
            float w[KT], win[KT], nw[16];
#pragma unroll
            for (int j = 0; j < KT; ++j) w[j] = head.dw_w[j * 256 + c];
            const float bv = head.dw_b[c];
            const float gc = head.gconst ? head.gconst[c] : 0.f;
            const bool has_gc = head.gconst != nullptr;
            const int hb0 = row0 / head.seq_t, ht0 = row0 - hb0 * head.seq_t, lr_last = M - 1 - row0;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const SeqRow q = seq_row(hb0, ht0, head.seq_t, min(half * 16 + rr, lr_last));
                nw[rr] = head.glu[((size_t)q.b * (pad + head.seq_t) + q.t + pad) * 256 + c];
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) win[j] = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int lr = half * 16 + rr;
                const SeqRow q = seq_row(hb0, ht0, head.seq_t, min(lr, lr_last));
                const int b = q.b, t = q.t;
                if (rr == 0 || t == 0 || row0 + lr >= M) {
                    const float* gin = head.glu + ((size_t)b * (pad + head.seq_t) + t) * 256 + c;     // padded rows t .. t + pad
#pragma unroll
                    for (int j = 0; j < pad; ++j) win[j + 1] = (has_gc && t + j < pad) ? gc : gin[(size_t)j * 256];
                }
#pragma unroll
                for (int j = 0; j < pad; ++j) win[j] = win[j + 1];
                win[pad] = nw[rr];
                float acc = bv;                       // out[t] = b + sum_j w[j] * gpad[t + j]
#pragma unroll
                for (int j = 0; j < KT; ++j) acc = fmaf(w[j], win[j], acc);
                xn[lr * DU_XLD + c] = acc;
            }
        }
        // ---- head 2: LayerNorm + SiLU per row (wave w: rows 4w .. 4w+3), in place: the A tile of pointwise_conv2 ------------
        {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(head.lnw + lane * 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(head.lnb + lane * 4);
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int lr = wave * 4 + rr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&xn[lr * DU_XLD + lane * 4]);
                const float mean = du_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = du_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                f32x4 o;
                o[0] = d0 * rstd * ww[0] + bb[0];
                o[1] = d1 * rstd * ww[1] + bb[1];
                o[2] = d2 * rstd * ww[2] + bb[2];
                o[3] = d3 * rstd * ww[3] + bb[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = o[i] / (1.0f + expf(-o[i]));
                *reinterpret_cast<f32x4*>(&xn[lr * DU_XLD + lane * 4]) = o;
            }
        }
        // ---- head 3: pointwise_conv2 on all 8 waves (wave w: output channels 32w .. 32w+31), + bias, pad mask, residual -> x
        // and, raw, back into the xn tile.  head.W: the packed copy [wave][slab j][group g][lane][4] of pack_rows_pc_kernel.
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(head.W), 0, DU_D * DU_D * 4, 0x00020000);
        auto hld = [&](int j, int g) -> f32x4 { return du_bufld(hrs, lane16, (unsigned)((wave * 8 + j) * 4 + g) * 256u); };
        f32x4 hp[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) hp[k][g] = hld(k, g);
        __syncthreads();                                  // A tile complete
        {
            const float* xa = xn + frow * DU_XLD + 4 * fh;
            const int col = wave * 32 + frow;
            const float bv = head.bias[col];
            float res[16];
            unsigned padded = 0;                          // bit r: row r of this lane is a padded frame (pad mask of the conv module)
            const int mb0 = row0 / head.seq_t, mt0 = row0 - mb0 * head.seq_t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrc = min((r & 3) + 8 * (r >> 2) + 4 * fh, M - 1 - row0);
                res[r] = x[(size_t)(row0 + lrc) * DU_D + col];
                if (head.lens) {
                    const SeqRow q = seq_row(mb0, mt0, head.seq_t, lrc);
                    if (head.mstride * q.t >= head.lens[q.b]) padded |= 1u << r;
                }
            }
            gw_head = *reinterpret_cast<const f32x4*>(lnw + lane * 4);      // the FFN's own LayerNorm, for the prologue below
            gb_head = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 a[2];
                a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], hp[j & 3][g][q], acc, 0, 0, 0);
                        if (q == 3 && j + 4 < 8) hp[j & 3][g] = hld(j + 4, g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();                              // every wave has read its A fragments: the tile may be overwritten
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int row = row0 + lr;
                float v = acc[r] + bv;
                if (padded & (1u << r)) v = 0.f;
                v = res[r] + v;
                if (row < M) x[(size_t)row * DU_D + col] = v;
                xn[lr * DU_XLD + col] = v;
            }
        }
        __syncthreads();                                  // the updated rows are in the xn tile (and on their way to x)
    }

    // ---- this wave's packed weight stream: slab s = chunk * 8 + j, fragment (n, g) at ((s_of_wave * 2 + n) * 4 + g) * 256 ------
    const int nchunk = dff / DU_CH;
    const int nlast = nchunk - 1;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(role == 0 ? p1 : p2), 0, dff * DU_D * 4, 0x00020000);
    auto pld = [&](int chunk, int j, int n, int g) -> f32x4 {
        return du_bufld(wrs, lane16, (unsigned)(((((chunk * 4 + idx) * 8 + j) * 2 + n) * 4 + g)) * 256u);
    };
    f32x4 pre[2][2][4];                                   // [k-slab parity][tile n][group g]
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][n][g] = pld(0, k, n, g);

    // ---- LayerNorm prologue: wave w normalises rows 4w..4w+3 ----------------------------------------
    {
        const f32x4 gw = HEADK > 0 ? gw_head : *reinterpret_cast<const f32x4*>(lnw + lane * 4);
        const f32x4 gb = HEADK > 0 ? gb_head : *reinterpret_cast<const f32x4*>(lnb + lane * 4);
        f32x4 v4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = min(row0 + wave * 4 + rr, M - 1);
            v4[rr] = HEADK > 0 ? *reinterpret_cast<const f32x4*>(&xn[(wave * 4 + rr) * DU_XLD + lane * 4])
                               : *reinterpret_cast<const f32x4*>(x + (size_t)row * DU_D + lane * 4);
        }
        if (TAIL && tail.pre_lnw) {
            // the previous layer's closing LayerNorm on my rows, written back as the new residual stream
            const f32x4 pw = *reinterpret_cast<const f32x4*>(tail.pre_lnw + lane * 4);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(tail.pre_lnb + lane * 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 v = v4[rr];
                const float mean = du_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = du_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                f32x4 o;
                o[0] = d0 * rstd * pw[0] + pb[0];
                o[1] = d1 * rstd * pw[1] + pb[1];
                o[2] = d2 * rstd * pw[2] + pb[2];
                o[3] = d3 * rstd * pw[3] + pb[3];
                v4[rr] = o;
                const int row = row0 + wave * 4 + rr;
                if (row < M) *reinterpret_cast<f32x4*>(x + (size_t)row * DU_D + lane * 4) = o;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            const f32x4 v = v4[rr];
            f32x4 o;
            if (AFFINE) {          // Squeezeformer: ada_scale * x + ada_bias (positionwise.py:57-58), no LayerNorm
                o[0] = gw[0] * v[0] + gb[0];
                o[1] = gw[1] * v[1] + gb[1];
                o[2] = gw[2] * v[2] + gb[2];
                o[3] = gw[3] * v[3] + gb[3];
            } else {
                const float mean = du_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = du_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                o[0] = d0 * rstd * gw[0] + gb[0];
                o[1] = d1 * rstd * gw[1] + gb[1];
                o[2] = d2 * rstd * gw[2] + gb[2];
                o[3] = d3 * rstd * gw[3] + gb[3];
            }
            if (row0 + lr >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&xn[lr * DU_XLD + lane * 4]) = o;
        }
    }
    __syncthreads();                                 // xn tile complete

    // refill of fragment (n, g) of the ring entry slab (chunk, j) just finished with: slab (chunk, j) + 2, clamped past the end
    // (re-fetches land in registers nobody reads any more)
    auto refill = [&](int chunk, int j, int n, int g) {
        pre[j & 1][n][g] = pld(min(chunk + (j + 2) / 8, nlast), (j + 2) & 7, n, g);
    };

    // Phases p = 0 .. nchunk+1, one workgroup barrier at the end of each:
    //   producer: MFMAs of chunk p (p < nchunk); bias + SiLU + LDS store of chunk p-1 in the issue slots between them -> hs[(p-1) & 1]
    //   consumer: chunk p-2 from hs[(p-2) & 1]
    if (role == 0) {
        const float* xa = xn + frow * DU_XLD + 4 * fh;
        f32x16 accp[2];                    // raw sums of the previous chunk
        float bvp[2] = {0.f, 0.f};
        f32x4 resx[8];                     // drain phases: this wave's 8 residual rows on their way into the xn tile
#pragma unroll
        for (int r = 0; r < 16; ++r) { accp[0][r] = 0.f; accp[1][r] = 0.f; }
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            float* hprev = hs + ((phase + 1) & 1) * DU_BM * DU_HLD + idx * 64 + frow;     // buffer (phase-1) & 1
            auto finish = [&](int e) {     // bias + SiLU of element e = 16 n + r of the previous chunk (C layout: row = (r&3)+8(r>>2)+4fh)
                const int n = e >> 4, r = e & 15;
                const float v = accp[n][r] + bvp[n];
                hprev[((r & 3) + 8 * (r >> 2) + 4 * fh) * DU_HLD + 32 * n] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            };
            if (phase < nchunk) {
                const int chunk = phase;
                const float bv0 = b1[chunk * DU_CH + idx * 64 + frow], bv1 = b1[chunk * DU_CH + idx * 64 + 32 + frow];
                f32x16 acc[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 a[2];
                    a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j & 1][0][g][q], acc[0], 0, 0, 0);
                            if (q == 3) refill(chunk, j, 0, g);
                            if (phase > 0 && q == 0) finish(4 * j + g);          // 32 elements over the 32 (j, g) groups
                            __builtin_amdgcn_sched_barrier(0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j & 1][1][g][q], acc[1], 0, 0, 0);
                            if (q == 3) refill(chunk, j, 1, g);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { accp[0][r] = acc[0][r]; accp[1][r] = acc[1][r]; }
                bvp[0] = bv0;
                bvp[1] = bv1;
            } else if (phase == nchunk) {
#pragma unroll
                for (int e = 0; e < 32; ++e) finish(e);
                // drain: the LayerNorm tile has had its last read (barrier of phase nchunk - 1); the producers, idle from here on,
                // bring the raw residual rows back into it while the consumers multiply their last two chunks
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = min(row0 + idx * 8 + i, M - 1);
                    resx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * DU_D + lane * 4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&xn[(idx * 8 + i) * DU_XLD + lane * 4]) = resx[i];
            }
            __syncthreads();
        }
    } else {
        f32x16 acc2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
        const float bv2n[2] = {b2[idx * 64 + frow], b2[idx * 64 + 32 + frow]};
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            if (phase >= 2) {
                const int chunk = phase - 2;
                const float* ha = hs + (phase & 1) * DU_BM * DU_HLD + frow * DU_HLD + 4 * fh;       // buffer (phase-2) & 1
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 a[2];
                    a[0] = *reinterpret_cast<const f32x4*>(ha + j * 32);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(ha + j * 32 + 8 * (g + 1));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j & 1][0][g][q], acc2[0], 0, 0, 0);
                            if (q == 3) refill(chunk, j, 0, g);
                            __builtin_amdgcn_sched_barrier(0);
                            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j & 1][1][g][q], acc2[1], 0, 0, 0);
                            if (q == 3) refill(chunk, j, 1, g);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- epilogue (consumers): x <- x + scale * (acc2 + b2) ---------------------------------------------------
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = idx * 64 + n * 32 + frow;
            const float bv2 = bv2n[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float v = xn[lr * DU_XLD + col] + scale * (acc2[n][r] + bv2);       // residual row: put there by the producers
                if (row0 + lr < M) x[(size_t)(row0 + lr) * DU_D + col] = v;
                if (TAIL) xn[lr * DU_XLD + col] = v;          // the LayerNorm tile is free: every producer passed its last read
            }
        }
    }
    if (!TAIL) return;

    // ---- tail stage: out[32 rows, 768] = LayerNorm_tail(x_new) . Wt^T + bt, all 8 waves, three column tiles per wave together ----
    // tail.W: the packed copy [wave][slab j][tile t][group g][lane][4] of pack_rows_dual_kernel (N = 768)
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tail.W), 0, 768 * DU_D * 4, 0x00020000);
    auto tld = [&](int j, int t, int g) -> f32x4 { return du_bufld(trs, lane16, (unsigned)((((wave * 8 + j) * 3 + t) * 4 + g)) * 256u); };
    f32x4 tp[2][3][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) tp[k][t][g] = tld(k, t, g);
    {
        const f32x4 gw = *reinterpret_cast<const f32x4*>(tail.lnw + lane * 4);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(tail.lnb + lane * 4);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&xn[lr * DU_XLD + lane * 4]);
            const float mean = du_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = du_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            f32x4 o;
            o[0] = d0 * rstd * gw[0] + gb[0];
            o[1] = d1 * rstd * gw[1] + gb[1];
            o[2] = d2 * rstd * gw[2] + gb[2];
            o[3] = d3 * rstd * gw[3] + gb[3];
            if (row0 + lr >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&xn[lr * DU_XLD + lane * 4]) = o;
        }
    }
    __syncthreads();                                  // normalised tile complete
    {
        const float* xa = xn + frow * DU_XLD + 4 * fh;
        float bv[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[t] = tail.bias[t * 256 + wave * 32 + frow];      // requested before the MFMAs
        f32x16 acc[3];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 a[2];
            a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], tp[j & 1][t][g][q], acc[t], 0, 0, 0);
                        if (q == 3 && j + 2 < 8) tp[j & 1][t][g] = tld(j + 2, t, g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int col = t * 256 + wave * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < M) tail.out[(size_t)row * tail.ldo + col] = acc[t][r] + bv[t];
            }
        }
    }
}

// W1 [dff, 256], W2 [256, dff] -> [chunk (256 units)][idx][k-slab j][tile n][group g][lane][4]:
//   p1: W1[chunk*256 + 64 idx + 32 n + (lane & 31)][32 j + 8 g + 4 (lane >> 5) + q]
//   p2: W2[64 idx + 32 n + (lane & 31)][chunk*256 + 32 j + 8 g + 4 (lane >> 5) + q]
__global__ __launch_bounds__(256) void pack_ffn_dual_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                            float* __restrict__ p1, float* __restrict__ p2, int dff) {
    const size_t n_el = (size_t)dff * DU_D;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * n_el) return;
    const bool second = t >= n_el;
    const size_t e = second ? t - n_el : t;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), g = (int)((e >> 8) & 3), n = (int)((e >> 10) & 1),
              j = (int)((e >> 11) & 7), idx = (int)((e >> 14) & 3), chunk = (int)(e >> 16);
    const int frow = lane & 31, fh = lane >> 5;
    const int k = 32 * j + 8 * g + 4 * fh + q;
    if (!second) p1[e] = w1[(size_t)(chunk * DU_CH + 64 * idx + 32 * n + frow) * DU_D + k];
    else p2[e] = w2[(size_t)(64 * idx + 32 * n + frow) * dff + chunk * DU_CH + k];
}
// W [768, 256] (fused QKV weights) -> [wave][slab j][tile t][group g][lane][4]:
//   P = W[t*256 + 32 wave + (lane & 31)][32 j + 8 g + 4 (lane >> 5) + q]
__global__ __launch_bounds__(256) void pack_rows_dual_kernel(const float* __restrict__ w, float* __restrict__ p) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)768 * DU_D) return;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), g = (int)((e >> 8) & 3);
    const int rest = (int)(e >> 10);                     // (wave * 8 + j) * 3 + t
    const int t = rest % 3, j = (rest / 3) & 7, wave = rest / 24;
    p[e] = w[(size_t)(t * 256 + 32 * wave + (lane & 31)) * DU_D + 32 * j + 8 * g + 4 * (lane >> 5) + q];
}
void launch_pack_ffn_dual(const float* w1, const float* w2, float* p1, float* p2, int dff, hipStream_t s) {
    const size_t n = (size_t)2 * dff * DU_D;
    hipLaunchKernelGGL(pack_ffn_dual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w1, w2, p1, p2, dff);
}
void launch_pack_rows_dual(const float* w, float* p, hipStream_t s) {
    hipLaunchKernelGGL(pack_rows_dual_kernel, dim3((unsigned)(768 * DU_D / 256)), dim3(256), 0, s, w, p);
}

template <int AFFINE, int TAIL, int HEADK>
static void launch_dual_t(float* x, const float* lnw, const float* lnb, const float* p1, const float* b1, const float* p2,
                          const float* b2, int M, int dff, float eps, float scale, hipStream_t s, const FfnTail& tail,
                          const FfnHead& head) {
    const size_t lds = (size_t)(DU_BM * DU_XLD + 2 * DU_BM * DU_HLD) * sizeof(float);
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_dual_kernel<AFFINE, TAIL, HEADK>), lds, attr);
    hipLaunchKernelGGL((ffn_dual_kernel<AFFINE, TAIL, HEADK>), dim3((M + DU_BM - 1) / DU_BM), dim3(512), lds, s, x, lnw, lnb, p1, b1,
                       p2, b2, M, dff, eps, scale, tail, head);
}

// p1 / p2: packed copies of launch_pack_ffn_dual; tail->W: launch_pack_rows_dual (N = 768 only); head->W: launch_pack_rows_pc.
// Returns 0 (block only), 2 (tail stage done), 4 (head stage done), -1 (sizes not covered: the caller uses ffn_pc.hip)
int launch_ffn_dual(float* x, const float* lnw, const float* lnb, const float* p1, const float* b1, const float* p2,
                    const float* b2, int M, int dff, float eps, float scale, int affine_prologue, hipStream_t s,
                    const FfnTail* tail, const FfnHead* head) {
    if (M <= 0) return 0;
    if (dff % DU_CH != 0 || dff < 2 * DU_CH) return -1;
    if (affine_prologue) {
        if (tail || head) return -1;
        launch_dual_t<1, 0, 0>(x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale, s, FfnTail{}, FfnHead{});
        return 0;
    }
    if (head && head->glu) {
        if (tail) return -1;
        if (head->ktaps == 15) launch_dual_t<0, 0, 15>(x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale, s, FfnTail{}, *head);
        else if (head->ktaps == 7) launch_dual_t<0, 0, 7>(x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale, s, FfnTail{}, *head);
        else return -1;
        return 4;
    }
    if (tail && tail->out) {
        if (tail->N != 768) return -1;
        launch_dual_t<0, 1, 0>(x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale, s, *tail, FfnHead{});
        return 2;
    }
    launch_dual_t<0, 0, 0>(x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale, s, FfnTail{}, FfnHead{});
    return 0;
}

}  // namespace masr
