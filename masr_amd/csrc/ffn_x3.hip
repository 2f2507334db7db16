// EXPLORATORY precision mode (masr_debug_set key 20, bit 2; never the contract path): the position-wise feed-forward block
//     x <- x + scale * ( W2 . silu( W1 . LayerNorm(x) + b1 ) + b2 )                     (conformer/positionwise.py:30-37)
// fused like ffn_pc.hip, but on the bf16 matrix pipe with split operands (gemm_bf16x3.hip explains the arithmetic: every fp32
// operand a = a_hi + a_lo in bf16, a product = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation).
//
// One workgroup = 32 rows, 8 waves.  d_ff is walked in 16 chunks of 128 hidden units; in phase k the four PRODUCER waves
// compute the hidden tile of chunk k (wave p: units 32p .. 32p+31, K = 256: 16 k-steps x 3 MFMAs) while the four CONSUMER waves
// accumulate chunk k - 1 into the output (wave q: output columns 64q .. 64q+63, K = 128: 8 k-steps x 2 tiles x 3 MFMAs); one
// workgroup barrier per chunk, the hidden tensor never leaves the CU (bf16 hi / lo tiles in LDS, double-buffered).
//   * LayerNorm(x) is constant for the whole kernel: every producer wave keeps its A fragments (16 k-steps x hi, lo = 128 VGPRs)
//     in registers -- no LDS traffic on the producers' A side at all.
//   * The weights are PRE-PACKED once (pack_ffn_x3_kernel, cached per FFN by the engine): split into bf16 hi / lo and laid out
//     in exactly the order the waves consume them -- [chunk][tile][k-step][hi | lo][lane][8] -- so every B fragment is ONE
//     fully coalesced global_load_dwordx4 per lane (1 KB per wave instruction) straight into MFMA operand layout: no LDS staging,
//     no conversion, no barrier on the weight path; an 8-deep (producers) / 4-deep (consumers) register ring hides the latency.
//   * What bounds the kernel (measured 72 us per block at B = 32 x 10 s; the fp32 kernel: 144 us, of which 116 us are matrix-pipe
//     time): each workgroup pulls the whole 4 MB of (hi, lo) weights through its own CU -- 57 GB/s per CU, the same with 248, 124
//     or 62 workgroups in flight, i.e. a per-CU L2 -> L1 streaming rate, not the L2s' aggregate bandwidth; the matrix pipe needs
//     21 us.  Tried: the whole next chunk's weights in flight (16 / 8-deep rings, LayerNorm tile in LDS instead of registers):
//     77 us -- latency is not what the stream waits for.  More rows per workgroup would halve the bytes per flop but leaves half
//     the CUs idle at this batch size.
#include "common.h"

namespace masr {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int FX_D = 256;
constexpr int FX_CH = 128;                 // hidden units per chunk
constexpr int FX_XLD = FX_D + 4;           // fp32 staging row of the LayerNorm tile
constexpr int FX_HLD = FX_CH + 8;          // bf16 row of a hidden tile (272 B: 16-byte reads of 16 rows tile all banks)

__device__ __forceinline__ unsigned cvt2(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const unsigned*>(&h);
}
__device__ __forceinline__ unsigned short bf16_1(float a) { return (unsigned short)(cvt2(a, 0.f) & 0xFFFFu); }
// eight consecutive fp32 -> hi and lo fragments (8 bf16 each)
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
    u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = cvt2(v[2 * i], v[2 * i + 1]);
        l[i] = cvt2(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xFFFF0000u));
    }
    hi = *reinterpret_cast<const bf16x8*>(&h);
    lo = *reinterpret_cast<const bf16x8*>(&l);
}

// ---- weight packing ------------------------------------------------------------------------------------------------------
// W1 [dff, 256] -> p1[c][p][s][piece][lane][8]:  piece(W1[c*128 + p*32 + (lane & 31)][16 s + 8 (lane >> 5) + i])
// W2 [256, dff] -> p2[c][t][s][piece][lane][8]:  piece(W2[t*32 + (lane & 31)][c*128 + 16 s + 8 (lane >> 5) + i])
__global__ __launch_bounds__(256) void pack_ffn_x3_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                          unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                          int dff) {
    const size_t n1 = (size_t)dff * FX_D;       // elements per piece plane... (each source element yields a hi and a lo)
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= 2 * n1) return;
    const bool second = g >= n1;
    const size_t e = second ? g - n1 : g;       // index of (c, tile, s, lane, i) without the piece dimension
    const int i = (int)(e & 7), lane = (int)((e >> 3) & 63);
    float v;
    size_t dst;
    if (!second) {
        const int s = (int)((e >> 9) & 15), p = (int)((e >> 13) & 3), c = (int)(e >> 15);
        v = w1[(size_t)(c * FX_CH + p * 32 + (lane & 31)) * FX_D + 16 * s + 8 * (lane >> 5) + i];
        dst = ((((size_t)(c * 4 + p) * 16 + s) * 2) * 64 + lane) * 8 + i;
    } else {
        const int s = (int)((e >> 9) & 7), t = (int)((e >> 12) & 7), c = (int)(e >> 15);
        v = w2[(size_t)(t * 32 + (lane & 31)) * dff + c * FX_CH + 16 * s + 8 * (lane >> 5) + i];
        dst = ((((size_t)(c * 8 + t) * 8 + s) * 2) * 64 + lane) * 8 + i;
    }
    const unsigned short hi = bf16_1(v);
    const unsigned short lo = bf16_1(v - __uint_as_float((unsigned)hi << 16));
    unsigned short* out = second ? p2 : p1;
    out[dst] = hi;
    out[dst + 512] = lo;                         // the lo piece follows the hi piece of the same (c, tile, s): + 64 lanes x 8
}

// ---- the block -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void ffn_x3_kernel(float* __restrict__ x, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, const unsigned short* __restrict__ p1,
                                                     const float* __restrict__ b1, const unsigned short* __restrict__ p2,
                                                     const float* __restrict__ b2, int M, int dff, float eps, float scale,
                                                     int rot_on) {
    // one LDS area, two lives: the fp32 LayerNorm tile xs [32][260] during the prologue (33 280 B), then the hidden tiles
    // hh / hl [2 buffers][32][136] bf16 hi and lo pieces (34 816 B) -- a barrier separates the two uses
    __shared__ __align__(16) unsigned char fx_smem[4 * 32 * FX_HLD * 2];
    static_assert(4 * 32 * FX_HLD * 2 >= 32 * FX_XLD * 4, "hidden tiles must cover the LayerNorm tile");
    float* xs = reinterpret_cast<float*>(fx_smem);
    unsigned short* hh = reinterpret_cast<unsigned short*>(fx_smem);                 // [2][32 * FX_HLD]
    unsigned short* hl = hh + 2 * 32 * FX_HLD;                                       // [2][32 * FX_HLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 32;
    const int frow = lane & 31, kb = lane >> 5;
    const int nchunk = dff / FX_CH;
    // every workgroup walks the chunks in its own rotation: 248 workgroups asking the L2s for the SAME weight lines in the same
    // microsecond serialise on the channels that hold them (no rotation: 84 us per block, with it 72 us).  Workgroup b runs on
    // XCD b % 8, so b / 8 gives the ~31 workgroups that share an L2 different rotations.
    const int rot = rot_on ? (int)((blockIdx.x / 8) % (unsigned)nchunk) : 0;

    // ---- prologue: LayerNorm of the 32 rows -> xs (wave w: rows 4w .. 4w+3, lane: 4 channels) ------------------------------
    {
        const f32x4 gw = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int lr = wave * 4 + rr;
            const int row = min(row0 + lr, M - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * FX_D + lane * 4);
            const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = wave_sum_dpp(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            f32x4 o;
            o[0] = d0 * rstd * gw[0] + gb[0];
            o[1] = d1 * rstd * gw[1] + gb[1];
            o[2] = d2 * rstd * gw[2] + gb[2];
            o[3] = d3 * rstd * gw[3] + gb[3];
            *reinterpret_cast<f32x4*>(&xs[lr * FX_XLD + lane * 4]) = o;
        }
    }
    __syncthreads();

    if (wave < 4) {
        // =========================== producers: hidden tile of chunk k ===========================
        const int p = wave;
        bf16x8 ah[16], al[16];                  // LayerNorm(x) fragments of all 16 k-steps: resident for the whole kernel
#pragma unroll
        for (int s = 0; s < 16; ++s) split8(&xs[frow * FX_XLD + 16 * s + 8 * kb], ah[s], al[s]);
        __syncthreads();                        // xs is dead from here on: the hidden tiles take its place
        // weight stream of this wave: (c, p, s) blocks of 2 KB (hi 1 KB | lo 1 KB), lane's 16 bytes inside each
        const unsigned short* wbase = p1 + (size_t)lane * 8;
        auto wptr = [&](int c, int s) {
            int cr = c + rot;
            if (cr >= nchunk) cr -= nchunk;
            return wbase + (((size_t)(cr * 4 + p) * 16 + s) * 2) * 512;
        };
        bf16x8 rh[8], rl[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            rh[s] = *reinterpret_cast<const bf16x8*>(wptr(0, s));
            rl[s] = *reinterpret_cast<const bf16x8*>(wptr(0, s) + 512);
        }
        for (int c = 0; c <= nchunk; ++c) {
            if (c < nchunk) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float bias = b1[((c + rot) % nchunk) * FX_CH + p * 32 + frow];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const bf16x8 wh = rh[s & 7], wl = rl[s & 7];
                    // refill the ring slot: k-step s + 8 of this chunk, or s - 8 of the next
                    const int cn = s < 8 ? c : c + 1, sn = s < 8 ? s + 8 : s - 8;
                    if (cn < nchunk) {
                        rh[s & 7] = *reinterpret_cast<const bf16x8*>(wptr(cn, sn));
                        rl[s & 7] = *reinterpret_cast<const bf16x8*>(wptr(cn, sn) + 512);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], wh, acc, 0, 0, 0);
                }
                // bias + SiLU -> hi / lo pieces -> hidden tile of buffer c & 1 (C layout: unit = lane & 31, row from the register)
                unsigned short* dh = hh + (c & 1) * 32 * FX_HLD + p * 32 + frow;
                unsigned short* dl = hl + (c & 1) * 32 * FX_HLD + p * 32 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                    float v = acc[r] + bias;
                    v = v / (1.0f + __expf(-v));
                    const unsigned short hi = bf16_1(v);
                    dh[row * FX_HLD] = hi;
                    dl[row * FX_HLD] = bf16_1(v - __uint_as_float((unsigned)hi << 16));
                }
            }
            __syncthreads();
        }
    } else {
        // =========================== consumers: chunk k - 1 into the output ===========================
        const int q = wave - 4;
        __syncthreads();                        // (pairs with the producers' barrier after their fragment load)
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const unsigned short* wbase = p2 + (size_t)lane * 8;
        auto wptr = [&](int c, int t, int s) {
            int cr = c + rot;
            if (cr >= nchunk) cr -= nchunk;
            return wbase + (((size_t)(cr * 8 + 2 * q + t) * 8 + s) * 2) * 512;
        };
        bf16x8 rh[4][2], rl[4][2];               // ring over k-steps (4 deep), two column tiles each
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                rh[s][t] = *reinterpret_cast<const bf16x8*>(wptr(0, t, s));
                rl[s][t] = *reinterpret_cast<const bf16x8*>(wptr(0, t, s) + 512);
            }
        for (int c = 0; c <= nchunk; ++c) {
            if (c >= 1) {
                const int cc = c - 1;            // the chunk whose hidden tile is complete
                const unsigned short* sh = hh + (cc & 1) * 32 * FX_HLD + frow * FX_HLD + 8 * kb;
                const unsigned short* sl = hl + (cc & 1) * 32 * FX_HLD + frow * FX_HLD + 8 * kb;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(sh + 16 * s);
                    const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(sl + 16 * s);
                    bf16x8 wh[2], wl[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) { wh[t] = rh[s & 3][t]; wl[t] = rl[s & 3][t]; }
                    const int cn = s < 4 ? cc : cc + 1, sn = s < 4 ? s + 4 : s - 4;
                    if (cn < nchunk) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            rh[s & 3][t] = *reinterpret_cast<const bf16x8*>(wptr(cn, t, sn));
                            rl[s & 3][t] = *reinterpret_cast<const bf16x8*>(wptr(cn, t, sn) + 512);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, wh[t], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, wl[t], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, wh[t], acc[t], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        // ---- epilogue: x <- x + scale * (acc + b2) ------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 64 * q + 32 * t + frow;
            const float bias = b2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                if (row < M) {
                    float* xp = x + (size_t)row * FX_D + col;
                    *xp = *xp + scale * (acc[t][r] + bias);
                }
            }
        }
    }
}

}  // namespace

static int g_x3_rot = 1;
void set_ffn_x3_rotation(int on) { g_x3_rot = on; }

size_t ffn_x3_packed_elems(int dff) { return (size_t)2 * dff * FX_D; }      // bf16 elements per packed matrix (hi + lo)

void launch_pack_ffn_x3(const float* w1, const float* w2, unsigned short* p1, unsigned short* p2, int dff, hipStream_t s) {
    const size_t n = (size_t)2 * dff * FX_D;
    hipLaunchKernelGGL(pack_ffn_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w1, w2, p1, p2, dff);
}

// d_model 256, d_ff a multiple of 128 (the shipped YAMLs); false = not taken
bool launch_ffn_x3(float* x, const float* lnw, const float* lnb, const unsigned short* p1, const float* b1,
                   const unsigned short* p2, const float* b2, int M, int dff, float eps, float scale, hipStream_t s) {
    if (M <= 0 || dff % FX_CH != 0 || dff / FX_CH < 1) return false;
    hipLaunchKernelGGL(ffn_x3_kernel, dim3((M + 31) / 32), dim3(512), 0, s, x, lnw, lnb, p1, b1, p2, b2, M, dff, eps, scale,
                       g_x3_rot);
    return true;
}

}  // namespace masr
