// Squeezeformer encoder layer (post-LN) in THREE launches per layer: attention, and the two kernels of this file
// (reference masr/model_utils/squeezeformer/encoder.py:412-463, attention.py:112-115, convolution.py:92-148,
// positionwise.py:57-58):
//
//     x = LN1(x + Wo . attention(...))                       | STAGE 0 ("mid"), head
//     x = LN2(x + FFN1(ada_f1 * x + adb_f1))                 | STAGE 0, block + post-LN
//     glu = GLU(pw1(mask(ada_cv * x + adb_cv)))              | STAGE 0, tail          -> padded GLU buffer
//     ------------------------------------------------------- kernel boundary: the depthwise conv needs the neighbours' rows
//     x = LN3(x + mask(pw2(SiLU(BN(dwconv(glu))))))          | STAGE 1 ("end"), head
//     x = LN4(x + FFN2(ada_f2 * x + adb_f2))                 | STAGE 1, block + post-LN
//     qkv = Wqkv . (ada_att' * x + adb_att') + bqkv          | STAGE 1, tail (the NEXT layer's projection) -> qkv buffer
//     ------------------------------------------------------- kernel boundary: attention needs every key of the sequence
//
// A workgroup (8 waves, 2 per SIMD) owns 32 rows for the whole stage: a post-LayerNorm needs the complete 256-column row, and
// that row never leaves the CU between the residual add and the next GEMM.  The FFN in the middle is ffn_pc.hip's
// producer / consumer kernel (packed weight fragments through raw buffer loads, one workgroup barrier per chunk of 128 hidden
// units); the row-local GEMMs around it (K = 256; N = 256 / 512 / 768) run on all eight waves, wave w owning columns
// 32w .. 32w+31 of every 256-column tile, from the same packed layout (pack_rows_pc_kernel).
//
// Every stage performs the operations of the launches it replaces (rowgemm PRO_PLAIN / PRO_AFFINE with EPI_RESID / EPI_GLU /
// EPI_STORE, layernorm256_kernel, ffn_pc_kernel<AFFINE>, dwconv_ln_silu_kernel<KT, 1>) in the same order on the same operands:
// the encoder output is bit-identical to the unfused path (masr_debug_set key 36 = 0; tests/test_gpu_parity.py).
#include "common.h"

namespace masr {

static constexpr int SQ_BM = 32;
static constexpr int SQ_D = 256;
static constexpr int SQ_CH = 128;          // hidden units per FFN chunk
static constexpr int SQ_XLD = SQ_D + 4;    // 260
static constexpr int SQ_HLD = SQ_CH + 4;   // 132
static constexpr int SQ_NSET = 4;

__device__ __forceinline__ f32x4 sq_bufld(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, float_index * 4u, 0));
}

// LayerNorm of one 256-wide row held as a float4 per lane (the arithmetic of layernorm256_kernel / the GEMM prologues)
__device__ __forceinline__ f32x4 sq_layernorm(const f32x4 v, const f32x4 gw, const f32x4 gb, float eps) {
    const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    const float var = wave_sum_dpp(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + eps);
    f32x4 o;
    o[0] = d0 * rstd * gw[0] + gb[0];
    o[1] = d1 * rstd * gw[1] + gb[1];
    o[2] = d2 * rstd * gw[2] + gb[2];
    o[3] = d3 * rstd * gw[3] + gb[3];
    return o;
}
__device__ __forceinline__ f32x4 sq_affine(const f32x4 v, const f32x4 gw, const f32x4 gb) {
    f32x4 o;
    o[0] = gw[0] * v[0] + gb[0];
    o[1] = gw[1] * v[1] + gb[1];
    o[2] = gw[2] * v[2] + gb[2];
    o[3] = gw[3] * v[3] + gb[3];
    return o;
}

// One 32 x 32 output tile of this wave: acc = A[32, 256] (LDS tile, row stride SQ_XLD) . W_tile^T, the B fragments coming out of
// the ring `pre` (slab j in pre[j & 3], refilled with slab j + 4 through `next(j + 4, g)` as soon as group g's last MFMA has
// issued -- the loader decides what lies behind slab 7: the next tile of the stage, or a clamped re-fetch).
// MORE = false: nothing follows slab 7 (the ring is not refilled past the end).
template <bool MORE, class NEXT>
__device__ __forceinline__ void sq_tile(const float* xa, f32x4 (&pre)[SQ_NSET][4], f32x16& acc, NEXT next) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f32x4 a[2];
        a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j % SQ_NSET][g][q], acc, 0, 0, 0);
                if (q == 3 && (MORE || j + SQ_NSET < 8)) pre[j % SQ_NSET][g] = next(j + SQ_NSET, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// STAGE 0 = mid (out-projection head, FFN1, GLU tail), STAGE 1 = end (conv-module head with KT taps, FFN2, optional QKV tail)
template <int STAGE, int KT>
__global__ __launch_bounds__(512) void sqz_stage_kernel(SqzStageArgs p) {
    extern __shared__ __align__(16) float sm[];
    float* xn = sm;                              // [32][260]   the block's rows: A tile of whatever GEMM comes next
    float* hs = xn + SQ_BM * SQ_XLD;             // [2][32][132] hidden tile of the FFN, double-buffered

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, idx = wave & 3;  // waves w and w+4 share a SIMD: producer idx and consumer idx of the FFN
    const int row0 = blockIdx.x * SQ_BM;
    const int frow = lane & 31, fh = lane >> 5;
    const int M = p.M, dff = p.dff;
    const float eps = p.eps;
    float* x = p.x;
    const unsigned lane16 = lane * 16;
    const float* xa = xn + frow * SQ_XLD + 4 * fh;
    const int sb0 = row0 / p.seq_t, st0 = row0 - sb0 * p.seq_t;       // (sequence, frame) of the block's first row
    // a block of padded frames only: nothing a valid frame reads comes from it (its GLU rows stay zero from the per-resolution
    // memset, attention stops at the valid length) -- whole workgroup, before any barrier
    // (the encoder OUTPUT is the exception: its padded frames are written as zeros, by skipped and computed blocks alike, so that
    //  what a caller sees there does not depend on what the buffer held)
    const bool zero_pad_out = STAGE == 1 && p.skip_pad && p.lens && p.out != p.x;
    if (p.skip_pad && p.lens && st0 + (min(row0 + SQ_BM, M) - 1 - row0) < p.seq_t && p.mstride * st0 >= p.lens[sb0]) {
        if (zero_pad_out)
            for (int i = tid; i < SQ_BM * 64; i += 512)
                if (row0 + (i >> 6) < M) *reinterpret_cast<f32x4*>(p.out + (size_t)(row0 + (i >> 6)) * SQ_D + (i & 63) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    f32x4 pre[SQ_NSET][4];

    // =========================================================================================================================
    // head: 32 rows of a K = 256 projection + residual -> raw rows in the xn tile
    // =========================================================================================================================
    {
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.head_w), 0, SQ_D * SQ_D * 4, 0x00020000);
        auto hld = [&](int j, int g) -> f32x4 {            // fragment (slab j, group g) of this wave's 32 weight rows
            return sq_bufld(hrs, lane16, (unsigned)((wave * 8 + j) * 4 + g) * 256u);
        };
#pragma unroll
        for (int k = 0; k < SQ_NSET; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][g] = hld(k, g);
        if (STAGE == 0) {
            // A tile = the attention output rows (rowgemm PRO_PLAIN): wave w brings rows 4w .. 4w+3
            f32x4 v4[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = min(row0 + wave * 4 + rr, M - 1);
                v4[rr] = *reinterpret_cast<const f32x4*>(p.att + (size_t)row * SQ_D + lane * 4);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int lr = wave * 4 + rr;
                *reinterpret_cast<f32x4*>(&xn[lr * SQ_XLD + lane * 4]) = row0 + lr < M ? v4[rr] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            // A tile = SiLU(BatchNorm(depthwise_conv(glu))) of my 32 rows (dwconv_ln_silu_kernel<KT, 1>): thread = (channel c,
            // 16-row half); the window slides down the rows and is reloaded where a new sequence starts
            constexpr int pad = KT - 1;
            const int c = tid & 255, half = tid >> 8;
            float w[KT], win[KT], nw[16];
#pragma unroll
            for (int j = 0; j < KT; ++j) w[j] = p.dw_w[j * 256 + c];
            const float bv = p.dw_b[c];
            const float bsc = p.bn_scale[c], bsh = p.bn_shift[c];
            const float gc = p.gconst ? p.gconst[c] : 0.f;
            const bool has_gc = p.gconst != nullptr;
            // skip_pad: the row blocks of padded frames only never wrote their GLU rows; a padded frame inside the batch's padded
            // length holds glu(pointwise_conv1 bias) in the reference (its input is masked to zero BEFORE pointwise_conv1,
            // convolution.py:105-119) and the symmetric window of the last valid frames reaches it: substituted here (gpad =
            // the constant glu_const_kernel computes with the mid stage's own epilogue arithmetic: identical bits)
            const bool sub = p.skip_pad && p.gpad != nullptr && p.lens != nullptr;
            const float gp = sub ? p.gpad[c] : 0.f;
            const int lr_last = M - 1 - row0;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {              // the newest window element of every row, all 16 loads in flight
                const SeqRow q = seq_row(sb0, st0, p.seq_t, min(half * 16 + rr, lr_last));
                nw[rr] = p.glu[((size_t)q.b * (pad + p.seq_t) + q.t + pad) * 256 + c];
                if (sub) {
                    const int f = q.t + pad - p.glu_pad_l;
                    if (f < p.seq_t && p.mstride * f >= p.lens[q.b]) nw[rr] = gp;
                }
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) win[j] = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int lr = half * 16 + rr;
                const SeqRow q = seq_row(sb0, st0, p.seq_t, min(lr, lr_last));
                if (rr == 0 || q.t == 0 || row0 + lr >= M) {
                    const float* gin = p.glu + ((size_t)q.b * (pad + p.seq_t) + q.t) * 256 + c;     // padded rows t .. t + pad
#pragma unroll
                    for (int j = 0; j < pad; ++j) {
                        win[j + 1] = (has_gc && q.t + j < pad) ? gc : gin[(size_t)j * 256];
                        if (sub) {
                            const int f = q.t + j - p.glu_pad_l;
                            if (f >= 0 && f < p.seq_t && p.mstride * f >= p.lens[q.b]) win[j + 1] = gp;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < pad; ++j) win[j] = win[j + 1];
                win[pad] = nw[rr];
                float acc = bv;                            // out[t] = b + sum_j w[j] * gpad[t + j]
#pragma unroll
                for (int j = 0; j < KT; ++j) acc = fmaf(w[j], win[j], acc);
                float o = acc * bsc + bsh;                 // eval-mode BatchNorm1d, folded (convolution.py:62-67,137-141)
                o = o / (1.0f + expf(-o));
                xn[lr * SQ_XLD + c] = o;
            }
        }
        // the epilogue's bias, residual rows and pad flags are requested before the MFMAs (x is not written before them)
        const int col = wave * 32 + frow;
        const float hb = p.head_b[col];
        float res[16];
        unsigned padded = 0;                               // bit r: row r of this lane is a padded frame
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrc = min((r & 3) + 8 * (r >> 2) + 4 * fh, M - 1 - row0);
            res[r] = x[(size_t)(row0 + lrc) * SQ_D + col];
            if (STAGE == 1 && p.lens) {
                const SeqRow q = seq_row(sb0, st0, p.seq_t, lrc);
                if (p.mstride * q.t >= p.lens[q.b]) padded |= 1u << r;
            }
        }
        __syncthreads();                                   // A tile complete
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sq_tile<false>(xa, pre, acc, hld);
        __syncthreads();                                   // every wave has read its A fragments: the tile may be overwritten
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
            float v = acc[r] + hb;
            if (STAGE == 1 && (padded & (1u << r))) v = 0.f;       // conv module: padded frames are zeroed before the residual
            v = res[r] + 1.0f * v;
            xn[lr * SQ_XLD + col] = v;
        }
        __syncthreads();
    }

    // ---- post-LN of the head (LN1 / LN3) -> x; the FFN's adaptive scale / bias -> A tile (wave w: rows 4w .. 4w+3) ------------
    const int nchunk = dff / SQ_CH;
    const float* wpk = role == 0 ? p.w1 : p.w2;            // packed [chunk][idx][slab j][group g][lane][4] (pack_ffn_pc_kernel)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, dff * SQ_D * 4, 0x00020000);
    auto pld = [&](int chunk, int j, int g) -> f32x4 {
        return sq_bufld(wrs, lane16, (unsigned)((((chunk * 4 + idx) * 8 + j) * 4 + g)) * 256u);
    };
    {
        const f32x4 lw = *reinterpret_cast<const f32x4*>(p.ln_a_w + lane * 4);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(p.ln_a_b + lane * 4);
        const f32x4 aw = *reinterpret_cast<const f32x4*>(p.ffn_s + lane * 4);
        const f32x4 ab = *reinterpret_cast<const f32x4*>(p.ffn_b + lane * 4);
#pragma unroll
        for (int k = 0; k < SQ_NSET; ++k)                  // the FFN's first weight fragments travel under the LayerNorms
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][g] = pld(0, k, g);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr, row = row0 + lr;
            const f32x4 y = sq_layernorm(*reinterpret_cast<const f32x4*>(&xn[lr * SQ_XLD + lane * 4]), lw, lb, eps);
            if (row < M) *reinterpret_cast<f32x4*>(x + (size_t)row * SQ_D + lane * 4) = y;
            f32x4 o = sq_affine(y, aw, ab);
            if (row >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&xn[lr * SQ_XLD + lane * 4]) = o;
        }
    }
    __syncthreads();                                       // A tile of the FFN complete

    // =========================================================================================================================
    // FFN: x_raw = x + (W2 . silu(W1 . a + b1) + b2), producer / consumer waves (ffn_pc.hip, VAR == 2), raw rows -> xn tile
    // =========================================================================================================================
    const int nlast = nchunk - 1;
    auto side_work = [&](int chunk, int j, int slot) {
        if ((slot & 3) == 3) pre[j % SQ_NSET][slot >> 2] = pld(min(chunk + (j + SQ_NSET) / 8, nlast), (j + SQ_NSET) & 7, slot >> 2);
    };
    if (role == 0) {
        f32x16 accp;                       // raw sums of the previous chunk
        float bvp = 0.f;
        f32x4 resx[8];                     // drain phases: this wave's 8 residual rows on their way into the xn tile
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[r] = 0.f;
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            float* hprev = hs + ((phase + 1) & 1) * SQ_BM * SQ_HLD + idx * 32 + frow;     // buffer (phase-1) & 1
            auto finish = [&](int r) {     // bias + SiLU of element r of the previous chunk
                const float v = accp[r] + bvp;
                hprev[((r & 3) + 8 * (r >> 2) + 4 * fh) * SQ_HLD] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            };
            if (phase < nchunk) {
                const int chunk = phase;
                const float bv1 = p.b1[chunk * SQ_CH + idx * 32 + frow];
                f32x16 acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 a[2];
                    a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j % SQ_NSET][g][q], acc1, 0, 0, 0);
                            side_work(chunk, j, g * 4 + q);
                            if (phase > 0 && g * 4 + q == 1) finish(2 * j);
                            if (phase > 0 && g * 4 + q == 9) finish(2 * j + 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[r] = acc1[r];
                bvp = bv1;
            } else if (phase == nchunk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) finish(r);
                // drain: the A tile has had its last read; the idle producers bring the residual rows (the post-LN rows this
                // workgroup wrote to x above) back into it while the consumers multiply their last two chunks
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = min(row0 + idx * 8 + i, M - 1);
                    resx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * SQ_D + lane * 4);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&xn[(idx * 8 + i) * SQ_XLD + lane * 4]) = resx[i];
            }
            __syncthreads();
        }
    } else {
        f32x16 acc2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
        const float bv2n[2] = {p.b2[idx * 64 + frow], p.b2[idx * 64 + 32 + frow]};
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            if (phase >= 2) {
                const int chunk = phase - 2;
                const float* ha = hs + (phase & 1) * SQ_BM * SQ_HLD + frow * SQ_HLD + 4 * fh;       // buffer (phase-2) & 1
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 a[2];
                    a[0] = *reinterpret_cast<const f32x4*>(ha + (j >> 1) * 32);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(ha + (j >> 1) * 32 + 8 * (g + 1));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc2[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], pre[j % SQ_NSET][g][q], acc2[j & 1], 0, 0, 0);
                            side_work(chunk, j, g * 4 + q);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- epilogue (consumers): raw rows x + 1.0 * (acc2 + b2) into the xn tile (each element read and written by its lane)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = idx * 64 + n * 32 + frow;
            const float bv2 = bv2n[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                xn[lr * SQ_XLD + col] = xn[lr * SQ_XLD + col] + 1.0f * (acc2[n][r] + bv2);
            }
        }
    }

    // ---- post-LN of the FFN (LN2 / LN4) -> out rows; the tail's adaptive scale / bias (+ pad mask) -> A tile -------------------
    const bool has_tail = p.tail_w != nullptr;             // (STAGE 1 at a resolution change / the last layer: no tail)
    const int ntile = p.tail_n / 256;                      // 2 (GLU pair) or 3 (q | k | v)
    const __amdgpu_buffer_rsrc_t trs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_tail ? p.tail_w : p.head_w), 0, (has_tail ? p.tail_n : SQ_D) * SQ_D * 4, 0x00020000);
    auto tld_abs = [&](int s, int g) -> f32x4 {            // slab s = tile * 8 + j of this wave's stream over the tail's tiles
        const int t = min(s >> 3, ntile - 1), j = s & 7;
        return sq_bufld(trs, lane16, (unsigned)(((t * 8 + wave) * 8 + j) * 4 + g) * 256u);
    };
    {
        const f32x4 lw = *reinterpret_cast<const f32x4*>(p.ln_b_w + lane * 4);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(p.ln_b_b + lane * 4);
        f32x4 aw = {1.f, 1.f, 1.f, 1.f}, ab = {0.f, 0.f, 0.f, 0.f};
        if (has_tail) {
            aw = *reinterpret_cast<const f32x4*>(p.tail_s + lane * 4);
            ab = *reinterpret_cast<const f32x4*>(p.tail_sb + lane * 4);
#pragma unroll
            for (int k = 0; k < SQ_NSET; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[k][g] = tld_abs(k, g);
        }
        __syncthreads();                                   // raw rows complete
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr, row = row0 + lr;
            f32x4 y = sq_layernorm(*reinterpret_cast<const f32x4*>(&xn[lr * SQ_XLD + lane * 4]), lw, lb, eps);
            if (zero_pad_out && row < M) {
                const SeqRow q = seq_row(sb0, st0, p.seq_t, lr);
                if (p.mstride * q.t >= p.lens[q.b]) y = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (row < M) *reinterpret_cast<f32x4*>(p.out + (size_t)row * SQ_D + lane * 4) = y;
            if (has_tail) {
                bool live = row < M;
                if (STAGE == 0 && live && p.lens) {        // conv module: padded frames are zeroed AFTER the adaptive scale / bias
                    const SeqRow q = seq_row(sb0, st0, p.seq_t, lr);
                    live = p.mstride * q.t < p.lens[q.b];
                }
                f32x4 o = sq_affine(y, aw, ab);
                if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&xn[lr * SQ_XLD + lane * 4]) = o;
            }
        }
    }
    if (!has_tail) return;
    __syncthreads();                                       // A tile of the tail complete

    // =========================================================================================================================
    // tail: [32, tail_n] = A . Wt^T + bt on all 8 waves; STAGE 0: (value, gate) tiles -> GLU -> padded buffer; STAGE 1: q | k | v
    // =========================================================================================================================
    if (STAGE == 0) {
        const int ch = wave * 32 + frow;
        const float bva = p.tail_b[ch], bvg = p.tail_b[256 + ch];
        f32x16 accv, accg;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accv[r] = 0.f; accg[r] = 0.f; }
        sq_tile<true>(xa, pre, accv, [&](int s, int g) { return tld_abs(s, g); });
        sq_tile<false>(xa, pre, accg, [&](int s, int g) { return tld_abs(8 + s, g); });
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (row0 + lr >= M) continue;
            const SeqRow q = seq_row(sb0, st0, p.seq_t, lr);
            const size_t crow = (size_t)q.b * (p.seq_t + p.glu_pad_tot) + p.glu_pad_l + q.t;
            const float g = accg[r] + bvg;
            p.glu_out[crow * SQ_D + ch] = (accv[r] + bva) * __builtin_amdgcn_rcpf(1.0f + __expf(-g));
        }
    } else {
        for (int t = 0; t < ntile; ++t) {
            const int col = t * 256 + wave * 32 + frow;
            const float bv = p.tail_b[col];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            sq_tile<true>(xa, pre, acc, [&](int s, int g) { return tld_abs(t * 8 + s, g); });
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < M) p.tail_out[(size_t)row * p.tail_n + col] = acc[r] + bv;
            }
        }
    }
}

template <int STAGE, int KT>
static void launch_sq_t(const SqzStageArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(SQ_BM * SQ_XLD + 2 * SQ_BM * SQ_HLD) * sizeof(float);
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(sqz_stage_kernel<STAGE, KT>), lds, attr);
    hipLaunchKernelGGL((sqz_stage_kernel<STAGE, KT>), dim3((a.M + SQ_BM - 1) / SQ_BM), dim3(512), lds, s, a);
}

// false: the sizes are not covered (the caller runs the unfused launches)
bool launch_sqz_stage(const SqzStageArgs& a, int stage, hipStream_t s) {
    if (a.M <= 0) return true;
    if (a.dff % SQ_CH != 0 || a.dff < 2 * SQ_CH || a.seq_t <= 0) return false;
    if (a.tail_w && a.tail_n != (stage == 0 ? 512 : 768)) return false;
    if (stage == 0) {
        launch_sq_t<0, 1>(a, s);
        return true;
    }
    if (a.ktaps == 31) launch_sq_t<1, 31>(a, s);
    else if (a.ktaps == 15) launch_sq_t<1, 15>(a, s);
    else return false;
    return true;
}

}  // namespace masr
