// Fused position-wise feed-forward block, producer/consumer wave specialisation (reference conformer/positionwise.py:30-37,
// conformer/encoder.py:113-121,150-158):
//     x <- x + scale * ( W2 . silu( W1 . LayerNorm(x) + b1 ) + b2 )
// One workgroup = 32 rows, 8 waves = 2 per SIMD.  The two waves of a SIMD have DIFFERENT jobs:
//   producer  wave p (0..3): hidden tile  h[32 rows, 32 units]  = silu(xn . W1[chunk*128 + 32p .. +31, 0..255]^T + b1)
//                            (128 MFMA 32x32x2 per chunk, K = 256 from the LayerNorm tile in LDS) -> hs[phase & 1]
//   consumer  wave c (0..3): output tiles acc2[32 rows, 64 cols] += h(prev chunk) . W2[64c .. 64c+63, chunk]^T
//                            (128 MFMA per chunk, K = 128 from hs[(phase-1) & 1])
// In phase k the producers work on chunk k while the consumers work on chunk k-1: the bias + SiLU + LDS traffic of one
// wave overlaps with the matrix work of the other wave on the same SIMD, and there is ONE workgroup barrier per chunk
// (round 1's first version split K of the first GEMM over the waves and exchanged partial sums through LDS, two barriers per
// chunk; that kernel is gone).  Weights (VAR == 2, every launch since round 3): packed copies in the order the waves consume
// them, fetched with raw buffer loads straight into MFMA operand registers (ring of 16 fragments = 64 MFMAs ahead) -- no LDS and
// no barrier on the weight path.  VAR == 0 keeps the earlier scheme for A/B runs (masr_debug_set key 23): wave-private,
// double-buffered LDS slabs [32][36] filled by the consuming wave itself with a 4-deep register prefetch rotation.
// TAIL = 1 (first macaron FFN of an offline Conformer layer): the 32 finished rows do not leave the CU before the next
// row-local stage -- they are put back into the LayerNorm tile, normalised with the attention block's LayerNorm, and all 8 waves
// run the fused QKV projection on them ([768, 256] weights through the same wave-private slab stream, 3 output tiles per
// wave; conformer/attention.py:53-79, encoder.py:123-131).  This replaces a separate launch whose ramp, row reload +
// LayerNorm prologue and store tail cost about as much as its 20 us of matrix work.
#include "common.h"

namespace masr {

static constexpr int PC_BM = 32;
static constexpr int PC_D = 256;
static constexpr int PC_CH = 128;          // hidden units per chunk
static constexpr int PC_XLD = PC_D + 4;    // 260
static constexpr int PC_HLD = PC_CH + 4;   // 132
static constexpr int PC_WLD = 32 + 4;
static constexpr int PC_WSLAB = 32 * PC_WLD;
static constexpr int PC_NSET = 4;
#ifdef PC_NOFENCE_PACKED
#define PC_FENCE(var) do { if ((var) != 2) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PC_FENCE(var) __builtin_amdgcn_sched_barrier(0)
#endif

// VAR == 2: the packed fragments are fetched with raw buffer loads -- a descriptor in SGPRs, ONE constant per-lane byte offset
// (16 * lane) and a wave-uniform scalar offset per fragment: no per-load VALU address arithmetic between the MFMAs and half
// the address registers of a 64-bit flat address (tools/mfma_study.hip: an MFMA stream that pulls 1 KB per four MFMAs and
// wave runs at 138 TF with buffer loads, 132 TF with global loads; without any loads 151 TF).  PC_BUFLOAD 0 keeps global loads.
#ifndef PC_BUFLOAD
#define PC_BUFLOAD 1
#endif
__device__ __forceinline__ f32x4 pc_bufld(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, float_index * 4u, 0));
}

__device__ __forceinline__ float pc_wsum(float v) {
    return wave_sum_dpp(v);
}

// HEADK > 0 (second FFN of an offline Conformer layer): the conv module's depthwise conv (HEADK taps) + LayerNorm + SiLU +
// pointwise_conv2 + residual run first, on the same 32 rows (FfnHead, common.h): the depthwise conv reads its HEADK - 1 earlier
// GLU rows from the padded buffer the previous kernel wrote, everything behind it is row-local.  Replaces two launches
// (dwconv_ln_silu_kernel, rowgemm PRO_PLAIN / EPI_RESID) with the same arithmetic in the same order (bit-identical).
template <int AFFINE, int SPLIT, int VAR, int TAIL, int HEADK>
__global__ __launch_bounds__(512) void ffn_pc_kernel(float* x, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2, int M,
                                                     int dff, float eps, float scale, float* partial,
                                                     int chunks_per_block, FfnTail tail, FfnHead head) {
    extern __shared__ __align__(16) float sm[];
    float* xn = sm;                              // [32][260]   LayerNorm(x) tile (A operand of GEMM1)
    float* hs = xn + PC_BM * PC_XLD;             // [2][32][132] hidden tile (A operand of GEMM2), double-buffered
    float* wpv = hs + 2 * PC_BM * PC_HLD;        // [8 waves][2][32][36] wave-private weight slabs

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2, idx = wave & 3;  // waves w and w+4 share a SIMD: producer idx and consumer idx
    const int row0 = blockIdx.x * PC_BM;
    const int frow = lane & 31, fh = lane >> 5;
    float* wmine = wpv + wave * 2 * PC_WSLAB;
    const int lr8 = lane >> 3, lc4 = (lane & 7) * 4;
    f32x4 pre[PC_NSET][4];
    auto dst_of = [&](int s, int i) -> float* { return wmine + (s & 1) * PC_WSLAB + (lr8 + 8 * i) * PC_WLD + lc4; };
    const float* wfrag = wmine + frow * PC_WLD + 4 * fh;

    f32x4 gw_head = {0.f, 0.f, 0.f, 0.f}, gb_head = gw_head;
    if (HEADK > 0) {
        // ---- head 1: depthwise conv of my 32 rows -> xn tile.  Thread = (channel c, 16-row half); the window slides down the
        // rows like dwconv_ln_silu_kernel's and is reloaded where a new sequence starts (rows are (b, t) = divmod(row, seq_t)).
        {
            constexpr int KT = HEADK > 0 ? HEADK : 1, pad = KT - 1;      // (KT: no zero-length arrays in the HEADK = 0 kernels)
            const int c = tid & 255, half = tid >> 8;
            float w[KT], win[KT], nw[16];
#pragma unroll
            for (int j = 0; j < KT; ++j) w[j] = head.dw_w[j * 256 + c];
            const float bv = head.dw_b[c];
            const float gc = head.gconst ? head.gconst[c] : 0.f;
            const bool has_gc = head.gconst != nullptr;
            // the newest window element of every row (padded row t + pad) is requested up front: 16 loads in flight instead of
            // one exposed memory round trip per row of the sliding loop
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
                xn[lr * PC_XLD + c] = acc;
            }
        }
        // ---- head 2: LayerNorm + SiLU per row (wave w: rows 4w .. 4w+3), in place: the A tile of pointwise_conv2 ------------
        {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(head.lnw + lane * 4);       // (requested before the barrier)
            const f32x4 bb = *reinterpret_cast<const f32x4*>(head.lnb + lane * 4);
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int lr = wave * 4 + rr;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&xn[lr * PC_XLD + lane * 4]);
                const float mean = pc_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = pc_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                f32x4 o;
                o[0] = d0 * rstd * ww[0] + bb[0];
                o[1] = d1 * rstd * ww[1] + bb[1];
                o[2] = d2 * rstd * ww[2] + bb[2];
                o[3] = d3 * rstd * ww[3] + bb[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = o[i] / (1.0f + expf(-o[i]));
                *reinterpret_cast<f32x4*>(&xn[lr * PC_XLD + lane * 4]) = o;
            }
        }
        // ---- head 3: pointwise_conv2 on all 8 waves (wave w: output channels 32w .. 32w+31, K = 256 in 8 slabs through the
        // wave-private slab pipeline), + bias, pad mask, residual -> x (global) and, raw, back into the xn tile ----------------
        // (VAR == 2: head.W is the packed copy [wave][slab j][group g][lane][4] -- fragments straight into registers, as in the
        //  main loops; otherwise the rows go through the wave-private slabs)
        const float* hl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hl[i] = head.W + (size_t)(wave * 32 + lr8 + 8 * i) * PC_D + lc4;
        const float* hpk = head.W + (size_t)wave * 8 * 4 * 256 + (size_t)lane * 4;
        const unsigned lane16h = lane * 16;
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(head.W), 0, PC_D * PC_D * 4, 0x00020000);
        auto hld = [&](int j, int g) -> f32x4 {            // fragment (slab j, group g) of this wave's pointwise_conv2 rows
            if (PC_BUFLOAD) return pc_bufld(hrs, lane16h, (unsigned)((wave * 8 + j) * 4 + g) * 256u);
            return *reinterpret_cast<const f32x4*>(hpk + (size_t)(j * 4 + g) * 256);
        };
        if (VAR == 2) {
#pragma unroll
            for (int k = 0; k < PC_NSET; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[k][g] = hld(k, g);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[0][i] = *reinterpret_cast<const f32x4*>(hl[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dst_of(0, i)) = pre[0][i];
#pragma unroll
            for (int k = 1; k <= PC_NSET; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) pre[k % PC_NSET][i] = *reinterpret_cast<const f32x4*>(hl[i] + k * 32);
        }
        __syncthreads();                                  // A tile complete
        {
            const float* xa = xn + frow * PC_XLD + 4 * fh;
            // the epilogue's bias and residual rows are requested first and land under the MFMAs (x is not written before them)
            const int col = wave * 32 + frow;
            const float bv = head.bias[col];
            float res[16];
            unsigned padded = 0;                          // bit r: row r of this lane is a padded frame (pad mask of the conv module)
            const int mb0 = row0 / head.seq_t, mt0 = row0 - mb0 * head.seq_t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrc = min((r & 3) + 8 * (r >> 2) + 4 * fh, M - 1 - row0);
                res[r] = x[(size_t)(row0 + lrc) * PC_D + col];
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
                const float* wp = wfrag + (j & 1) * PC_WSLAB;
                f32x4 a[2], b[2];
                a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
                if (VAR != 2) b[0] = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g + 1 < 4) {
                        a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
                        if (VAR != 2) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + 8 * (g + 1));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], VAR == 2 ? pre[j % PC_NSET][g][q] : b[g & 1][q], acc, 0, 0, 0);
                        const int slot = g * 4 + q, pset = (j + 1) % PC_NSET;
                        if (VAR == 2) {
                            if ((slot & 3) == 3 && j + PC_NSET < 8) pre[j % PC_NSET][g] = hld(j + PC_NSET, g);
                        } else if (slot < 8) {
                            if ((slot & 1) == 0 && j + 1 < 8) *reinterpret_cast<f32x4*>(dst_of(j + 1, slot >> 1)) = pre[pset][slot >> 1];
                        } else if ((slot & 1) == 0 && j + 1 + PC_NSET < 8) {
                            pre[pset][(slot - 8) >> 1] = *reinterpret_cast<const f32x4*>(hl[(slot - 8) >> 1] + (j + 1 + PC_NSET) * 32);
                        }
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
                if (SPLIT) {          // all d_ff slices of the row block computed the same rows from the same old x: slice 0 publishes
                    if (blockIdx.y == 0 && row < M) head.xout[(size_t)row * PC_D + col] = v;
                } else if (row < M) {
                    x[(size_t)row * PC_D + col] = v;
                }
                xn[lr * PC_XLD + col] = v;
            }
        }
        __syncthreads();                                  // the updated rows are in the xn tile (and on their way to x)
    }

    // ---- LayerNorm prologue: wave w normalises rows 4w..4w+3 ----------------------------------------
    {
        const f32x4 gw = HEADK > 0 ? gw_head : *reinterpret_cast<const f32x4*>(lnw + lane * 4);
        const f32x4 gb = HEADK > 0 ? gb_head : *reinterpret_cast<const f32x4*>(lnb + lane * 4);
        f32x4 v4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = min(row0 + wave * 4 + rr, M - 1);
            v4[rr] = HEADK > 0 ? *reinterpret_cast<const f32x4*>(&xn[(wave * 4 + rr) * PC_XLD + lane * 4])
                               : *reinterpret_cast<const f32x4*>(x + (size_t)row * PC_D + lane * 4);
        }
        if (TAIL && tail.pre_lnw) {
            // the previous layer's closing LayerNorm on my rows, written back as the new residual stream (read again by the
            // epilogue of this workgroup only, after workgroup barriers)
            const f32x4 pw = *reinterpret_cast<const f32x4*>(tail.pre_lnw + lane * 4);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(tail.pre_lnb + lane * 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 v = v4[rr];
                const float mean = pc_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = pc_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                f32x4 o;
                o[0] = d0 * rstd * pw[0] + pb[0];
                o[1] = d1 * rstd * pw[1] + pb[1];
                o[2] = d2 * rstd * pw[2] + pb[2];
                o[3] = d3 * rstd * pw[3] + pb[3];
                v4[rr] = o;
                const int row = row0 + wave * 4 + rr;
                if (row < M) *reinterpret_cast<f32x4*>(x + (size_t)row * PC_D + lane * 4) = o;
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
                const float mean = pc_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float var = pc_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
                const float rstd = 1.0f / sqrtf(var + eps);
                o[0] = d0 * rstd * gw[0] + gb[0];
                o[1] = d1 * rstd * gw[1] + gb[1];
                o[2] = d2 * rstd * gw[2] + gb[2];
                o[3] = d3 * rstd * gw[3] + gb[3];
            }
            if (row0 + lr >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&xn[lr * PC_XLD + lane * 4]) = o;
        }
    }

    // ---- this wave's weight slab stream: 8 slabs (32 rows x 32 k) per chunk -----------------------------------------
    //   producer p, slab j : W1[chunk*128 + 32p + r][32j .. 32j+31]
    //   consumer c, slab j : W2[64c + 32(j&1) + r][chunk*128 + 32(j>>1) .. +31]     (tile n = j&1, k-slab j>>1)
    const int chunk_lo = SPLIT ? blockIdx.y * chunks_per_block : 0;
    const int chunk_hi = SPLIT ? min(chunk_lo + chunks_per_block, dff / PC_CH) : dff / PC_CH;
    const int nchunk = chunk_hi - chunk_lo;
    // branch-free addressing: role-dependent strides are wave-uniform scalars
    const float* wbase = role == 0 ? w1 + (size_t)(idx * 32) * PC_D : w2 + (size_t)(idx * 64) * dff;
    const size_t rs = role == 0 ? (size_t)PC_D : (size_t)dff;            // row stride of my weight matrix
    const size_t cs = role == 0 ? (size_t)PC_CH * PC_D : (size_t)PC_CH;  // offset of one chunk
    const float* wlane = wbase + (size_t)lr8 * rs + lc4;
    auto src_of = [&](int chunk, int j, int i) -> const float* {   // j, i are compile-time constants at every call
        const size_t joff = role == 0 ? (size_t)(j * 32) : (size_t)((j & 1) * 32) * dff + (size_t)((j >> 1) * 32);
        return wlane + (size_t)(8 * i) * rs + (size_t)chunk * cs + joff;
    };
    const int nlast = chunk_hi - 1;
    // VAR == 2: w1 / w2 point to PACKED copies (pack_ffn_pc_kernel below) laid out in the order the waves consume them --
    // [chunk][wave idx][slab j][group g][lane][4] -- so that the B fragment of four MFMAs is ONE coalesced 16-byte-per-lane
    // load straight into operand layout: the wave-private LDS slabs and their ds_write / ds_read pairs drop out of the main
    // loops (same operand values in the same MFMA order: bit-identical results).  `pre[s & 3][g]` then holds the fragments of
    // slab s, refilled with slab s + 4 as soon as the group's last MFMA has issued.
    const float* pbase = (role == 0 ? w1 : w2) + (size_t)lane * 4;
    auto psrc = [&](int chunk, int j, int g) -> const float* {
        return pbase + ((((size_t)chunk * 4 + idx) * 8 + j) * 4 + g) * 256;
    };
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(role == 0 ? w1 : w2), 0, (VAR == 2 ? dff : 0) * PC_D * 4, 0x00020000);
    auto pld = [&](int chunk, int j, int g) -> f32x4 {
        if (PC_BUFLOAD) return pc_bufld(wrs, lane16, (unsigned)((((chunk * 4 + idx) * 8 + j) * 4 + g)) * 256u);
        return *reinterpret_cast<const f32x4*>(psrc(chunk, j, g));
    };
    // side work in the MFMA issue slots of slab (c, j): store slab s+1 (set (j+1)%NSET) to LDS, refill that set with slab
    // s+1+NSET (chunk index clamped past the end: re-fetches land in buffers nobody reads any more)
    auto side_work = [&](int chunk, int j, int slot) {
        if (VAR == 2) {
            if ((slot & 3) == 3) pre[j % PC_NSET][slot >> 2] = pld(min(chunk + (j + PC_NSET) / 8, nlast), (j + PC_NSET) & 7, slot >> 2);
            return;
        }
        const int p = (j + 1) % PC_NSET;
        if (slot < 8) {
            if ((slot & 1) == 0) *reinterpret_cast<f32x4*>(dst_of(j + 1, slot >> 1)) = pre[p][slot >> 1];
        } else if ((slot & 1) == 0 && VAR != 1) {
            pre[p][(slot - 8) >> 1] = *reinterpret_cast<const f32x4*>(
                src_of(min(chunk + (j + 1 + PC_NSET) / 8, nlast), (j + 1 + PC_NSET) & 7, (slot - 8) >> 1));
        }
    };

    if (VAR == 2) {
#pragma unroll
        for (int k = 0; k < PC_NSET; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][g] = pld(chunk_lo, k, g);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[0][i] = *reinterpret_cast<const f32x4*>(src_of(chunk_lo, 0, i));
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dst_of(0, i)) = pre[0][i];
#pragma unroll
        for (int k = 1; k <= PC_NSET; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[k % PC_NSET][i] = *reinterpret_cast<const f32x4*>(src_of(chunk_lo, k, i));
    }
    __syncthreads();                                 // xn tile complete

    // Phases p = 0 .. nchunk+1, one workgroup barrier at the end of each:
    //   producer: MFMAs of chunk p (p < nchunk) with the bias + SiLU + LDS store of chunk p-1 spread over the free issue
    //             slots between them (the raw sums of chunk p-1 wait in accp) -> hs[(p-1) & 1]
    //   consumer: chunk p-2 from hs[(p-2) & 1]
    // The two roles run separate loops (wave-uniform branch, same number of barriers), so each keeps only its own
    // accumulators live.
    if (role == 0) {
        const float* xa = xn + frow * PC_XLD + 4 * fh;
        f32x16 accp;                       // raw sums of the previous chunk
        float bvp = 0.f;
        f32x4 resx[8];                     // drain phases: this wave's 8 residual rows on their way into the xn tile
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[r] = 0.f;
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            float* hprev = hs + ((phase + 1) & 1) * PC_BM * PC_HLD + idx * 32 + frow;     // buffer (phase-1) & 1
            auto finish = [&](int r) {     // bias + SiLU of element r of the previous chunk (C layout: row = (r&3)+8(r>>2)+4fh)
                const float v = accp[r] + bvp;
                hprev[((r & 3) + 8 * (r >> 2) + 4 * fh) * PC_HLD] = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
            };
            if (phase < nchunk) {
                const int chunk = chunk_lo + phase;
                const float bv1 = b1[chunk * PC_CH + idx * 32 + frow];
                f32x16 acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* wp = wfrag + (j & 1) * PC_WSLAB;
                    f32x4 a[2], b[2];
                    a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
                    if (VAR != 2) b[0] = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) {
                            a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
                            if (VAR != 2) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + 8 * (g + 1));
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], VAR == 2 ? pre[j % PC_NSET][g][q] : b[g & 1][q],
                                                                        acc1, 0, 0, 0);
                            side_work(chunk, j, g * 4 + q);
                            // odd slots 1 and 9 of every slab: one element of the previous chunk's epilogue each
                            if (phase > 0 && g * 4 + q == 1) finish(2 * j);
                            if (phase > 0 && g * 4 + q == 9) finish(2 * j + 1);
                            PC_FENCE(VAR);   // keep the written MFMA / load / LDS-store interleave
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[r] = acc1[r];
                bvp = bv1;
            } else if (phase == nchunk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) finish(r);
                // drain: the LayerNorm tile has had its last read (barrier of phase nchunk - 1).  The producers, idle from here
                // on, bring the raw residual rows back into it (row 8 idx + i, one 1 KB row per load) while the consumers
                // multiply their last two chunks; the epilogue then adds the residual from LDS instead of waiting for memory.
                if (!SPLIT) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = min(row0 + idx * 8 + i, M - 1);
                        resx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * PC_D + lane * 4);
                    }
                }
            } else if (!SPLIT) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&xn[(idx * 8 + i) * PC_XLD + lane * 4]) = resx[i];
            }
            __syncthreads();
        }
    } else {
        f32x16 acc2[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
        float bv2n[2] = {0.f, 0.f};
        if (!SPLIT) { bv2n[0] = b2[idx * 64 + frow]; bv2n[1] = b2[idx * 64 + 32 + frow]; }
        for (int phase = 0; phase <= nchunk + 1; ++phase) {
            if (phase >= 2) {
                const int chunk = chunk_lo + phase - 2;
                const float* ha = hs + (phase & 1) * PC_BM * PC_HLD + frow * PC_HLD + 4 * fh;       // buffer (phase-2) & 1
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float* wp = wfrag + (j & 1) * PC_WSLAB;
                    f32x4 a[2], b[2];
                    a[0] = *reinterpret_cast<const f32x4*>(ha + (j >> 1) * 32);
                    if (VAR != 2) b[0] = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g + 1 < 4) {
                            a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(ha + (j >> 1) * 32 + 8 * (g + 1));
                            if (VAR != 2) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + 8 * (g + 1));
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc2[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], VAR == 2 ? pre[j % PC_NSET][g][q] : b[g & 1][q],
                                                                               acc2[j & 1], 0, 0, 0);
                            side_work(chunk, j, g * 4 + q);
                            PC_FENCE(VAR);
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
            if (SPLIT) {
                float* pp = partial + (size_t)blockIdx.y * M * PC_D;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (row < M) pp[(size_t)row * PC_D + col] = acc2[n][r];
                }
            } else {
                const float bv2 = bv2n[n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = xn[lr * PC_XLD + col] + scale * (acc2[n][r] + bv2);       // residual row: put there by the producers
                    if (row0 + lr < M) x[(size_t)(row0 + lr) * PC_D + col] = v;
                    if (TAIL) xn[lr * PC_XLD + col] = v;          // the LayerNorm tile is free: every producer passed its last read
                }
            }
        }
    }
    if (!TAIL) return;

    // ---- tail stage: out[32 rows, tail.N] = LayerNorm_tail(x_new) . Wt^T + bt, all 8 waves -----------------------------------
    {
        const f32x4 gw = *reinterpret_cast<const f32x4*>(tail.lnw + lane * 4);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(tail.lnb + lane * 4);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&xn[lr * PC_XLD + lane * 4]);
            const float mean = pc_wsum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = pc_wsum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            f32x4 o;
            o[0] = d0 * rstd * gw[0] + gb[0];
            o[1] = d1 * rstd * gw[1] + gb[1];
            o[2] = d2 * rstd * gw[2] + gb[2];
            o[3] = d3 * rstd * gw[3] + gb[3];
            if (row0 + lr >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&xn[lr * PC_XLD + lane * 4]) = o;
        }
    }
    // weight tile t of this wave: rows t * 256 + 32 * wave .. +31 of Wt [N, 256]; slab j = k range 32j .. 32j+31
    const int ntile = (tail.N + 255) / 256;
    const float* tl[4];                                // N is a multiple of 256 (launcher): per-lane bases + wave-uniform tile offset
#pragma unroll
    for (int i = 0; i < 4; ++i) tl[i] = tail.W + (size_t)(wave * 32 + lr8 + 8 * i) * PC_D + lc4;
    auto tsrc = [&](int t, int j, int i) -> const float* {
        return tl[i] + (size_t)(min(t, ntile - 1) * 256) * PC_D + j * 32;
    };
    // (VAR == 2: tail.W is the packed copy [tile t][wave][slab j][group g][lane][4])
    const float* tpk = tail.W + (size_t)wave * 8 * 4 * 256 + (size_t)lane * 4;
    auto tpsrc = [&](int t, int j, int g) -> const float* {
        return tpk + ((size_t)min(t, ntile - 1) * 8 * 8 * 4 + (size_t)(j * 4 + g)) * 256;
    };
    const __amdgpu_buffer_rsrc_t trs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tail.W), 0, (VAR == 2 ? ntile : 0) * 256 * PC_D * 4, 0x00020000);
    auto tld = [&](int t, int j, int g) -> f32x4 {
        if (PC_BUFLOAD) return pc_bufld(trs, lane16, (unsigned)(((min(t, ntile - 1) * 8 + wave) * 8 + j) * 4 + g) * 256u);
        return *reinterpret_cast<const f32x4*>(tpsrc(t, j, g));
    };
    if (VAR == 2) {
#pragma unroll
        for (int k = 0; k < PC_NSET; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[k][g] = tld(0, k, g);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[0][i] = *reinterpret_cast<const f32x4*>(tsrc(0, 0, i));
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(dst_of(0, i)) = pre[0][i];
#pragma unroll
        for (int k = 1; k <= PC_NSET; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[k % PC_NSET][i] = *reinterpret_cast<const f32x4*>(tsrc(0, k, i));
    }
    __syncthreads();                                  // normalised tile complete
    {
        const float* xa = xn + frow * PC_XLD + 4 * fh;
        for (int t = 0; t < ntile; ++t) {
            const int col = t * 256 + wave * 32 + frow;
            const float bv = tail.bias[min(col, tail.N - 1)];      // requested before the tile's MFMAs
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* wp = wfrag + (j & 1) * PC_WSLAB;
                f32x4 a[2], b[2];
                a[0] = *reinterpret_cast<const f32x4*>(xa + j * 32);
                if (VAR != 2) b[0] = *reinterpret_cast<const f32x4*>(wp);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g + 1 < 4) {
                        a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xa + j * 32 + 8 * (g + 1));
                        if (VAR != 2) b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(wp + 8 * (g + 1));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], VAR == 2 ? pre[j % PC_NSET][g][q] : b[g & 1][q], acc, 0, 0, 0);
                        const int slot = g * 4 + q, pset = (j + 1) % PC_NSET;
                        if (VAR == 2) {
                            if ((slot & 3) == 3) pre[j % PC_NSET][g] = tld(t + (j + PC_NSET) / 8, (j + PC_NSET) & 7, g);
                        } else if (slot < 8) {
                            if ((slot & 1) == 0) *reinterpret_cast<f32x4*>(dst_of(j + 1, slot >> 1)) = pre[pset][slot >> 1];
                        } else if ((slot & 1) == 0) {
                            pre[pset][(slot - 8) >> 1] = *reinterpret_cast<const f32x4*>(
                                tsrc(t + (j + 1 + PC_NSET) / 8, (j + 1 + PC_NSET) & 7, (slot - 8) >> 1));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (col < tail.N && tail.plane_stride > 0) {
                // planar q | k | v with per-sequence time padding (grouped attention): tile t IS plane t
                float* plane = tail.out + (size_t)t * tail.plane_stride + wave * 32 + frow;
                const int pb0 = row0 / tail.seq_t, pt0 = row0 - pb0 * tail.seq_t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const SeqRow q = seq_row(pb0, pt0, tail.seq_t, lr);
                    if (row0 + lr < M) plane[((size_t)q.b * (tail.seq_t + tail.pad_t) + q.t) * 256] = acc[r] + bv;
                }
            } else if (col < tail.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (row < M) tail.out[(size_t)row * tail.ldo + col] = acc[r] + bv;
                }
            }
        }
    }
}


// W1 [dff, 256], W2 [256, dff] -> the VAR == 2 layout [chunk][idx][slab j][group g][lane][4]:
//   p1: W1[chunk*128 + 32 idx + (lane & 31)][32 j + 8 g + 4 (lane >> 5) + q]
//   p2: W2[64 idx + 32 (j & 1) + (lane & 31)][chunk*128 + 32 (j >> 1) + 8 g + 4 (lane >> 5) + q]
__global__ __launch_bounds__(256) void pack_ffn_pc_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                          float* __restrict__ p1, float* __restrict__ p2, int dff) {
    const size_t n = (size_t)dff * PC_D;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 2 * n) return;
    const bool second = t >= n;
    const size_t e = second ? t - n : t;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), g = (int)((e >> 8) & 3), j = (int)((e >> 10) & 7),
              idx = (int)((e >> 13) & 3), chunk = (int)(e >> 15);
    const int frow = lane & 31, fh = lane >> 5;
    if (!second) p1[e] = w1[(size_t)(chunk * PC_CH + 32 * idx + frow) * PC_D + 32 * j + 8 * g + 4 * fh + q];
    else p2[e] = w2[(size_t)(64 * idx + 32 * (j & 1) + frow) * dff + chunk * PC_CH + 32 * (j >> 1) + 8 * g + 4 * fh + q];
}
// W [N, 256] (N a multiple of 256: the fused QKV weights, pointwise_conv2) -> [tile t][wave][slab j][group g][lane][4]:
//   P = W[t*256 + 32 wave + (lane & 31)][32 j + 8 g + 4 (lane >> 5) + q]         (tail / head stages with VAR == 2)
// (rows >= n_src -- the padding of a vocabulary that is no multiple of 256 -- are packed as zeros)
__global__ __launch_bounds__(256) void pack_rows_pc_kernel(const float* __restrict__ w, float* __restrict__ p, int N, int n_src) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)N * PC_D) return;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), g = (int)((e >> 8) & 3), j = (int)((e >> 10) & 7),
              wave = (int)((e >> 13) & 7), t = (int)(e >> 16);
    const int row = t * 256 + 32 * wave + (lane & 31);
    p[e] = row < n_src ? w[(size_t)row * PC_D + 32 * j + 8 * g + 4 * (lane >> 5) + q] : 0.f;
}
void launch_pack_rows_pc(const float* w, float* p, int N, hipStream_t s, int n_src) {
    const size_t n = (size_t)N * PC_D;
    hipLaunchKernelGGL(pack_rows_pc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, p, N, n_src < 0 ? N : n_src);
}
void launch_pack_ffn_pc(const float* w1, const float* w2, float* p1, float* p2, int dff, hipStream_t s) {
    const size_t n = (size_t)2 * dff * PC_D;
    hipLaunchKernelGGL(pack_ffn_pc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w1, w2, p1, p2, dff);
}

template <int HEADK, int VAR>
static void launch_head_t(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                          const float* b2, int M, int dff, float eps, float scale, hipStream_t s, const FfnHead& head, size_t lds) {
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<0, 0, VAR, 0, HEADK>), lds, attr);
    hipLaunchKernelGGL((ffn_pc_kernel<0, 0, VAR, 0, HEADK>), dim3((M + PC_BM - 1) / PC_BM), dim3(512), lds, s, x, lnw, lnb, w1, b1, w2,
                       b2, M, dff, eps, scale, (float*)nullptr, 0, FfnTail{}, head);
}

// VAR: 0 production slab pipeline | 1 without weight loads (floor measurement) | 2 packed weights straight into registers
// (w1 / w2 are then the packed copies; full kernels only -- the d_ff-split launches of small M keep the slab pipeline)
template <int AFFINE, int VAR>
static int launch_pc_t(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                       const float* b2, int M, int dff, float eps, float scale, float* partial, int nsplit, hipStream_t s,
                       const FfnPostLn* post, const FfnTail* tail, const FfnHead* head) {
    const size_t lds = (size_t)(PC_BM * PC_XLD + 2 * PC_BM * PC_HLD + 8 * 2 * PC_WSLAB) * sizeof(float);
    constexpr int TV = VAR == 1 ? 0 : VAR;               // variant of the tail / head kernels (never built for VAR == 1)
    static LdsAttr attr_full, attr_split, attr_tail;
    ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<AFFINE, 0, VAR, 0, 0>), lds, attr_full);
    ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<AFFINE, 1, 0, 0, 0>), lds, attr_split);
    ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<0, 0, TV, 1, 0>), lds, attr_tail);
    const int nchunk = dff / PC_CH;
    if (partial && nsplit > 1) {
        constexpr int SV = VAR == 2 ? 2 : 0;              // split launches: slab pipeline or packed weights (never the no-load variant)
        static LdsAttr attr_split_v;
        ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<AFFINE, 1, SV, 0, 0>), lds, attr_split_v);
        const int cpb = (nchunk + nsplit - 1) / nsplit;
        const int ny = (nchunk + cpb - 1) / cpb;          // every blockIdx.y owns at least one chunk
        // head stage on the split launch (few rows): every slice of a row block runs the conv module's [depthwise conv -> LN ->
        // SiLU -> pointwise_conv2 + residual] on the block's rows before its share of the FFN; slice 0 writes the updated rows to
        // head->xout and the reduction continues from there
#if MASR_EXPERIMENTS
        if (head && head->glu && head->xout && head->ktaps == 15 && !AFFINE && SV == 2) {
            static LdsAttr attr_split_h;
            ensure_dynamic_lds(reinterpret_cast<const void*>(ffn_pc_kernel<0, 1, 2, 0, 15>), lds, attr_split_h);
            hipLaunchKernelGGL((ffn_pc_kernel<0, 1, 2, 0, 15>), dim3((M + PC_BM - 1) / PC_BM, ny), dim3(512), lds, s, x, lnw, lnb, w1,
                               b1, w2, b2, M, dff, eps, scale, partial, cpb, FfnTail{}, *head);
            launch_ffn_reduce(x, partial, b2, M, ny, scale, s, post, head->xout);
            return (post && post->y ? 1 : 0) | 4;
        }
#endif
        hipLaunchKernelGGL((ffn_pc_kernel<AFFINE, 1, SV, 0, 0>), dim3((M + PC_BM - 1) / PC_BM, ny), dim3(512), lds, s, x, lnw, lnb, w1,
                           b1, w2, b2, M, dff, eps, scale, partial, cpb, FfnTail{}, FfnHead{});
        launch_ffn_reduce(x, partial, b2, M, ny, scale, s, post);
        return post && post->y ? 1 : 0;
    } else if (head && head->glu && (head->ktaps == 15 || head->ktaps == 7) && !AFFINE && VAR != 1 && !(tail && tail->out)) {
        if (head->ktaps == 15) launch_head_t<15, TV>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, s, *head, lds);
        else launch_head_t<7, TV>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, s, *head, lds);
        return 4;                                         // head stage done
    } else if (tail && tail->out && tail->N % 256 == 0 && !AFFINE && VAR != 1) {
        hipLaunchKernelGGL((ffn_pc_kernel<0, 0, TV, 1, 0>), dim3((M + PC_BM - 1) / PC_BM), dim3(512), lds, s, x, lnw, lnb, w1, b1, w2,
                           b2, M, dff, eps, scale, (float*)nullptr, 0, *tail, FfnHead{});
        return 2;                                         // tail stage done
    } else {
        hipLaunchKernelGGL((ffn_pc_kernel<AFFINE, 0, VAR, 0, 0>), dim3((M + PC_BM - 1) / PC_BM), dim3(512), lds, s, x, lnw, lnb, w1,
                           b1, w2, b2, M, dff, eps, scale, (float*)nullptr, 0, FfnTail{}, FfnHead{});
    }
    return 0;
}

// returns 1 when the post LayerNorm was applied (split-d_ff path), 2 when the tail stage ran (full kernel with `tail`), 4 when
// the head stage ran (full kernel with `head`), 0 when the caller still has to run whichever it asked for
int launch_ffn_pc(float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2,
                  const float* b2, int M, int dff, float eps, float scale, int affine_prologue, float* partial, int nsplit,
                  hipStream_t s, int variant, const FfnPostLn* post, const FfnTail* tail, const FfnHead* head) {
    if (M <= 0) return 0;
    if (affine_prologue && variant == 2) return launch_pc_t<1, 2>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, partial, nsplit, s, post, nullptr, nullptr);
    if (affine_prologue) return launch_pc_t<1, 0>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, partial, nsplit, s, post, nullptr, nullptr);
    if (variant == 1) return launch_pc_t<0, 1>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, partial, nsplit, s, post, nullptr, nullptr);
    if (variant == 2) return launch_pc_t<0, 2>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, partial, nsplit, s, post, tail, head);
    return launch_pc_t<0, 0>(x, lnw, lnb, w1, b1, w2, b2, M, dff, eps, scale, partial, nsplit, s, post, tail, head);
}

}  // namespace masr
