// Batched feature front-end: int16 PCM -> (RMS-dB normalise -> int16 truncation) -> Kaldi fbank.
//
// Follows AudioFeaturizer.featurize (reference masr/data_utils/featurizer/audio_featurizer.py:37-69):
//   AudioSegment float32 = pcm / 32768                         (data_utils/audio.py:532-546)
//   normalize(target_db): gain = target - 10*log10(mean(x^2)); x *= 10^(gain/20)   (audio.py:287-304,256-264,519-529)
//   to('int16'): x*32768, clip, C truncation                   (audio.py:549-574)
//   torchaudio.compliance.kaldi.fbank(num_mel_bins=80, frame_length=25, frame_shift=10, dither=0)
//       frames of 400 @ hop 160 (snip_edges) -> remove DC -> pre-emphasis 0.97 -> povey window ->
//       zero-pad to 512 -> |rFFT|^2 -> 80 mel bins -> log(max(., eps))    (audio_featurizer.py:120-138)
//
// MI355X mapping: one wave64 per frame, 4 frames per workgroup.  The 400 samples of a frame are read
// coalesced (int16), windowed in registers, and the 512-point real FFT is done as a 256-point complex
// radix-4 Stockham FFT with four elements per lane in registers (three LDS exchanges) plus the real-split
// post-pass.  The mel filterbank is applied from the LDS-resident power spectrum with a transposed
// [tap][filter] weight table.
// Frames past an utterance's length are written as zeros (collate_fn zero padding, collate_fn.py:8-42).
#include <algorithm>

#include "common.h"

namespace masr {

// ---- per-utterance gain: s = float32(10 ** ((target - 10*log10(mean(x^2))) / 20)) -----------------
// sample formats: int16 PCM (AudioSegment scales it by 1/32768, audio.py:532-546) or float32 samples
__device__ __forceinline__ float to_unit(int16_t v) { return (float)v * (1.0f / 32768.0f); }
__device__ __forceinline__ float to_unit(float v) { return v; }

// mean(samples ** 2) exactly as numpy computes it for a contiguous float32 array (audio.py:519-529,
// np.mean -> np.add.reduce): the reduction runs over buffered chunks of 8192 elements, accumulated
// sequentially in float32, and every chunk is summed by numpy's pairwise routine
// (numpy/core/src/umath/loops_utils.h.src: leaves of <= 128 elements with 8 strided accumulators
// combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), split point n/2 rounded down to a multiple of 8).
// Reproducing that order makes the gain -- and with it the int16 normalisation -- bit-identical
// to the reference (pinned by oracle/fbank.py against numpy itself, tests/test_oracle_golden.py).
// one rounding per operation, whatever the compiler's contraction mode: the product and the sum below carry no `contract` flag, so
// LLVM cannot fuse them with a neighbour (HIP's own mul_rn / add_rn intrinsics -- spelled with the two leading underscores and an
// f -- are plain ``*`` / ``+`` compiled in contract-fast mode unless OCML_BASIC_ROUNDED_OPERATIONS is defined: they DO fuse)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}

// NO FLOATING-POINT CONTRACTION in anything that feeds the mean square: hipcc's default (-ffp-contract=fast) turns
// ``r + f * f`` into one fused multiply-add -- a single rounding where numpy rounds the square and the sum separately.  Rounds 1-4
// shipped 96 v_pk_fma_f32 in rms_partial_kernel: the mean squares of ~10 % of utterances were one ulp off numpy's (found by the
// B = 32 route test of round 5; tools/mean_square_locate.py, tools/mean_square_probe.py).
template <class ST>
__device__ __forceinline__ float sq_unit(const ST* x, int i) {
#pragma clang fp contract(off)
    const float f = to_unit(x[i]);
    return mul_rn(f, f);               // samples ** 2: rounded to float32 before any add
}

template <class ST>
__device__ float pw_leaf(const ST* x, int off, int len) {
#pragma clang fp contract(off)
    if (len < 8) {
        float r = 0.f;
        for (int i = 0; i < len; ++i) r = add_rn(r, sq_unit(x, off + i));
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = sq_unit(x, off + j);
    int i = 8;
    const int m = len - (len % 8);
#pragma unroll 4
    for (; i < m; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = add_rn(r[j], sq_unit(x, off + i + j));
    }
    float res = add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])),
                          add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
    for (; i < len; ++i) res = add_rn(res, sq_unit(x, off + i));
    return res;
}

// the 128-element leaf of a full chunk, same add order as pw_leaf, but the 8 strided accumulators are fed by one
// 16-byte (int16) / two 16-byte (float) vector loads per step instead of 8 scalar loads (leaf starts are 256-B aligned
// relative to the utterance, and utterance rows are 16-B aligned when n_max is a multiple of 8 -- checked by the caller)
__device__ __forceinline__ float pw_leaf128(const int16_t* x, int off) {
#pragma clang fp contract(off)
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8* p = reinterpret_cast<const s16x8*>(x + off);
    s16x8 v[16];                           // all 16 loads in flight before the (ordered) add chain starts
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = p[i];
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = (float)v[0][j] * (1.0f / 32768.0f); r[j] = mul_rn(f, f); }
#pragma unroll
    for (int i = 1; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)v[i][j] * (1.0f / 32768.0f); r[j] = add_rn(r[j], mul_rn(f, f)); }
    return add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])),
                     add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
}
__device__ __forceinline__ float pw_leaf128(const float* x, int off) {
#pragma clang fp contract(off)
    const f32x4* p = reinterpret_cast<const f32x4*>(x + off);
    float r[8];
    {
        const f32x4 a = p[0], b = p[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = mul_rn(a[j], a[j]); r[4 + j] = mul_rn(b[j], b[j]); }
    }
#pragma unroll 5
    for (int i = 1; i < 16; ++i) {
        const f32x4 a = p[2 * i], b = p[2 * i + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = add_rn(r[j], mul_rn(a[j], a[j]));
            r[4 + j] = add_rn(r[4 + j], mul_rn(b[j], b[j]));
        }
    }
    return add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])),
                     add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
}

static constexpr int NP_BUF = 8192;       // numpy's default ufunc buffer size (elements)
static constexpr int MAX_CHUNKS = 2048;   // 16.7 M samples (17 min @ 16 kHz) per utterance

// pass 1: one wave per 8192-sample chunk (grid.x blocks of 4 waves, grid.y = utterance); the block with
// blockIdx.x == gridDim.x - 1 additionally sums the tail chunk (< 8192 samples) with the general recursion
template <class ST>
__global__ __launch_bounds__(256) void rms_partial_kernel(const ST* __restrict__ pcm, const int* __restrict__ nsamp,
                                                          int n_max, float* __restrict__ chunk_sum /*[B][MAX_CHUNKS+1]*/) {
#pragma clang fp contract(off)
    __shared__ int lvl_child[8][64], lvl_split[8][64], tmp_off[64], tmp_len[64];
    __shared__ float tmp_val[64];
    const int b = blockIdx.y;
    const int n = min(nsamp[b], NP_BUF * MAX_CHUNKS);
    const ST* x = pcm + (size_t)b * n_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = n / NP_BUF, tail = n - nc * NP_BUF;
    float* out = chunk_sum + (size_t)b * (MAX_CHUNKS + 1);
    // full chunk: 64 leaves of 128 elements, balanced tree == xor-butterfly (fp add is commutative)
    const int c = blockIdx.x * 4 + wave;
    if (c < nc) {
        float v = (n_max & 7) == 0 ? pw_leaf128(x, c * NP_BUF + lane * 128) : pw_leaf(x, c * NP_BUF + lane * 128, 128);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v = add_rn(v, __shfl_xor(v, o, 64));
        if (lane == 0) out[c] = v;
    }
    if (blockIdx.x != gridDim.x - 1 || wave != 0) return;
    // tail chunk (< 8192 samples) on wave 0: numpy's pairwise recursion (split at n/2 rounded down to a multiple of 8, leaves of
    // <= 128 elements) unrolled level by level -- one node per lane: a node that splits gets two slots in the next level, a leaf
    // is carried down unchanged; the sums then fold back up level by level in the same tree order.  A subtree of <= 4103 elements
    // has at most 33 nodes per level, a whole tail up to 65 (441 of the 8191 tail lengths: rounds 1-4 overflowed the 64 lanes
    // there), so the tail's two top-level halves are walked one after the other and added.
    if (tail == 0) {
        if (lane == 0) out[MAX_CHUNKS] = 0.f;
        return;
    }
    auto subtree = [&](int off0, int len0) -> float {
        int off = off0, len = len0;
        bool have = lane == 0;
        int depth = 0;
        while (true) {
            const bool split = have && len > 128;
            lvl_child[depth][lane] = 0;
            lvl_split[depth][lane] = split ? 1 : 0;
            if (!__ballot(split)) break;
            const int width = have ? (split ? 2 : 1) : 0;
            int incl = width;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            const int pos = incl - width;
            const int total = __shfl(incl, 63, 64);
            lvl_child[depth][lane] = pos;
            if (have) {
                if (split) {
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    tmp_off[pos] = off; tmp_len[pos] = n2;
                    tmp_off[pos + 1] = off + n2; tmp_len[pos + 1] = len - n2;
                } else {
                    tmp_off[pos] = off; tmp_len[pos] = len;
                }
            }
            __builtin_amdgcn_wave_barrier();
            have = lane < total;
            off = have ? tmp_off[lane] : 0;
            len = have ? tmp_len[lane] : 0;
            __builtin_amdgcn_wave_barrier();
            ++depth;
        }
        float val = have ? pw_leaf(x, off, len) : 0.f;
        for (int d = depth - 1; d >= 0; --d) {
            tmp_val[lane] = val;
            __builtin_amdgcn_wave_barrier();
            const int ch = lvl_child[d][lane];
            val = lvl_split[d][lane] ? add_rn(tmp_val[min(ch, 63)], tmp_val[min(ch + 1, 63)]) : tmp_val[min(ch, 63)];
            __builtin_amdgcn_wave_barrier();
        }
        return val;                                 // lane 0 holds the subtree's sum
    };
    float val;
    if (tail <= 128) {
        val = subtree(nc * NP_BUF, tail);
    } else {
        int n2 = tail / 2;
        n2 -= n2 % 8;
        const float left = subtree(nc * NP_BUF, n2);
        __builtin_amdgcn_wave_barrier();
        const float right = subtree(nc * NP_BUF + n2, tail - n2);
        val = add_rn(left, right);
    }
    if (lane == 0) out[MAX_CHUNKS] = val;
}

// pass 2: sequential float32 accumulation of the chunk sums (numpy's buffered reduction) + the gain
// ms_out (optional): the float32 mean square itself == np.mean(samples ** 2) of audio.py:524, bit for bit -- lets the host
// evaluate the reference's scalar numpy expressions (log10 / power in float32, not correctly rounded and machine dependent)
// on its own numpy and hand the gain back (use_db == 2)
__global__ void rms_final_kernel(const int* __restrict__ nsamp, int B, float target_db,
                                 const float* __restrict__ chunk_sum, float* __restrict__ gain, float* __restrict__ ms_out) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = min(nsamp[b], NP_BUF * MAX_CHUNKS);
    const int nc = n / NP_BUF, tail = n - nc * NP_BUF;
    const float* cs = chunk_sum + (size_t)b * (MAX_CHUNKS + 1);
    float tot = 0.f;
    for (int c = 0; c < nc; ++c) tot = add_rn(tot, cs[c]);
    if (tail > 0) tot = add_rn(tot, cs[MAX_CHUNKS]);
    // numpy's mean divides the float32 sum by an np.intp count: evaluated in float64 and rounded to float32 (core/_methods.py
    // _mean: ``ret.dtype.type(ret / rcount)``) -- the correctly rounded float32 quotient
    float ms = n > 0 ? (float)((double)tot / (double)n) : 0.f;
    if (ms_out) ms_out[b] = ms;
    if (ms == 0.f || !(ms == ms)) ms = 1.f;
    // float32 scalar arithmetic of rms_db / normalize / gain_db (numpy >= 2 promotion): each
    // step is evaluated in double and rounded once to float32
    const float lg = (float)log10((double)ms);
    const float rms_db = (float)(10.0 * (double)lg);
    const float g = target_db - rms_db;
    const float g20 = g / 20.0f;
    gain[b] = (float)pow(10.0, (double)g20);
}

template <class ST>
__device__ __forceinline__ float norm_sample(ST v, float s, int use_db) {
#pragma clang fp contract(off)
    float f = to_unit(v);
    if (use_db) f = mul_rn(f, s);
    f = f * 32768.0f;
    f = fminf(fmaxf(f, -32768.0f), 32767.0f);
    return truncf(f);
}

// The same value with fewer operations, for the feature kernel's int16 input: rn((v * 2^-15) * s) * 2^15 == rn(v * s) -- scaling by
// a power of two commutes with rounding while nothing leaves the normal range (|v * s * 2^-15| >= 2^-15 * s and a gain is
// >= 0.1 for int16 audio; zero, infinities and NaN behave alike) -- so the two exact scalings drop out.  float32 input keeps
// norm_sample's operation sequence.
__device__ __forceinline__ float norm_sample_fast(int16_t v, float s, int use_db) {
    float f = (float)v;
    if (use_db) f = mul_rn(f, s);
    return truncf(fminf(fmaxf(f, -32768.0f), 32767.0f));
}
__device__ __forceinline__ float norm_sample_fast(float v, float s, int use_db) { return norm_sample(v, s, use_db); }

template <class ST>
__global__ __launch_bounds__(256) void norm_int16_kernel(const ST* __restrict__ pcm, const int* __restrict__ nsamp,
                                                         int n_max, const float* __restrict__ gain, int use_db,
                                                         int16_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_max) return;
    const float s = use_db ? gain[b] : 1.f;
    out[(size_t)b * n_max + i] = i < nsamp[b] ? (int16_t)norm_sample(pcm[(size_t)b * n_max + i], s, use_db) : 0;
}

// ---- fbank ------------------------------------------------------------------------------------------
static constexpr int WIN = 400, HOP = 160, NFFT = 512, NBIN = 257, NMEL = 80;
static constexpr int FB_TAPS = 16;        // widest mel filter of the 80-bin 20 Hz - 8 kHz bank (build_fbank_tables checks it)

// ---- fbank kernel --------------------------------------------------------------------------------------------------------------
// One wave per frame.  The 256-point complex FFT is a Stockham radix-4 transform with FOUR elements per lane in registers: four
// in-register radix-4 butterflies with three exchanges through a 2 KB per-wave LDS buffer (24 LDS instructions per lane).  Rounds
// 1-4 ran a radix-2 FFT entirely in LDS (eight LDS-synchronised stages, 128 LDS instructions and 32 table loads per lane): 57.6 us
// per 32 x 10 s launch against 30.5-34.8 us for this kernel on the same box (profiles/r05_fbank_ab.txt; the old kernel is in the
// history, commit 'fbank: register radix-4 Stockham FFT kernel').  Lane j holds z[j + 64 r], r = 0..3 (z[n] = y[2n] + i y[2n+1], y = windowed frame zero-padded to 512): the
// first pass needs no exchange at all and the last one leaves Z[j + 64 q] in the lane's registers.  Complex arithmetic on float2
// values (packed fp32 instructions).  The left neighbour of the pre-emphasis comes over one DPP wave shift instead of an LDS
// round trip; the twiddles of the three later passes are one table row per lane (9 coalesced 8-byte loads, issued before the
// samples arrive); the mel filterbank reads a TRANSPOSED weight table [tap][filter] (coalesced; a filter has at most 16 taps):
// filters 0..63 one per lane, filters 64..79 four lanes each.
typedef float f2 __attribute__((ext_vector_type(2)));
static constexpr int FB_LDS = 272;                  // complex slots per wave: index c lives at c + (c >> 4) (bank-conflict padding)
__device__ __forceinline__ int fb_slot(int c) { return c + (c >> 4); }
__device__ __forceinline__ f2 fb_cmul(f2 a, f2 w) {                   // a * w
    const f2 t = f2{a.x, a.x} * w;
    return f2{a.y, a.y} * f2{-w.y, w.x} + t;
}
__device__ __forceinline__ void fb_radix4(f2 (&v)[4]) {              // forward DFT-4 in place: X_q = sum_r v_r (-i)^(r q)
    const f2 a0 = v[0] + v[2], a1 = v[0] - v[2], a2 = v[1] + v[3], a3 = v[1] - v[3];
    const f2 ia3 = f2{a3.y, -a3.x};                                     // -i * a3
    v[0] = a0 + a2;
    v[1] = a1 + ia3;
    v[2] = a0 - a2;
    v[3] = a1 - ia3;
}

template <class ST>
__global__ __launch_bounds__(256) void fbank_kernel(const ST* __restrict__ pcm, const int* __restrict__ nsamp, int n_max,
                                                       int use_db, const float* __restrict__ gain, FbankTables tb,
                                                       float* __restrict__ feats, int T_max) {
    __shared__ f2 lds[4][FB_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 4 + wv;
    if (t >= T_max) return;                        // (wave-uniform; only wave barriers below)
    const int n = nsamp[b];
    const int T = n >= WIN ? 1 + (n - WIN) / HOP : 0;
    float* dst = feats + ((size_t)b * T_max + t) * NMEL;
    if (t >= T) {                                  // frames past the utterance: zeros (collate_fn zero padding)
        dst[lane] = 0.f;
        if (lane < NMEL - 64) dst[64 + lane] = 0.f;
        return;
    }
    f2* buf = lds[wv];
    float* pw = reinterpret_cast<float*>(buf);     // the power spectrum reuses the buffer once the FFT is done
    const float s = use_db ? gain[b] : 1.f;

    // ---- table rows of this lane (independent of the samples: requested first) ----------------------------------------
    f2 tw[3][3];                                   // [pass 1..3][r = 1..3]
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int r = 0; r < 3; ++r) tw[p][r] = *reinterpret_cast<const f2*>(tb.twr4 + ((p * 3 + r) * 64 + lane) * 2);
    f2 win[4], tws[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s0 = min(2 * lane + 128 * r, WIN - 2);
        win[r] = *reinterpret_cast<const f2*>(tb.window + s0);
        tws[r] = *reinterpret_cast<const f2*>(tb.tw512 + 2 * (lane + 64 * r));
    }

    // ---- samples 2j + 128 r (+ 1), normalised like AudioSegment (gain, int16 truncation) -----------------------------------
    const ST* src = pcm + (size_t)b * n_max + (size_t)t * HOP;
    float x0[4], x1[4];
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s0 = 2 * lane + 128 * r;
        const bool in = s0 < WIN;                  // WIN is even: s0 and s0 + 1 are inside or outside together
        const int sc = in ? s0 : 0;
        x0[r] = in ? norm_sample_fast(src[sc], s, use_db) : 0.f;
        x1[r] = in ? norm_sample_fast(src[sc + 1], s, use_db) : 0.f;
        part += x0[r] + x1[r];
    }
    const float mean = wave_sum_dpp(part) / (float)WIN;

    // ---- DC removal, pre-emphasis (left neighbour: replicate at the frame start), povey window -> z[j + 64 r] ---------------
    f2 v[4];
    {
        float d0[4], d1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { d0[r] = x0[r] - mean; d1[r] = x1[r] - mean; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // sample 2j + 128 r - 1 = the second sample of lane j - 1 (same r); lane 0: of lane 63 one r earlier
            const float edge = r == 0 ? d0[0]
                                      : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d1[r - 1]), 63));
            const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                                     __builtin_bit_cast(int, d1[r]), 0x138, 0xf, 0xf, false));   // wave_shr:1
            const bool in = 2 * lane + 128 * r < WIN;
            v[r].x = in ? (d0[r] - 0.97f * prev) * win[r].x : 0.f;
            v[r].y = in ? (d1[r] - 0.97f * d0[r]) * win[r].y : 0.f;
        }
    }

    // ---- 256-point complex FFT: Stockham radix-4, passes Ns = 1, 4, 16, 64 ------------------------------------------------
    fb_radix4(v);                                  // pass 0 (Ns = 1): no twiddles, inputs already in registers
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[fb_slot(4 * lane + q)] = v[q];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 1; p < 4; ++p) {
        const int Ns = p == 1 ? 4 : p == 2 ? 16 : 64;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = buf[fb_slot(lane + 64 * r)];
#pragma unroll
        for (int r = 1; r < 4; ++r) v[r] = fb_cmul(v[r], tw[p - 1][r - 1]);
        fb_radix4(v);
        __builtin_amdgcn_wave_barrier();           // (a wave's LDS traffic is in order: every lane has read before anyone writes)
        if (p < 3) {
            const int k = lane & (Ns - 1), j0 = (lane - k) * 4 + k;
#pragma unroll
            for (int q = 0; q < 4; ++q) buf[fb_slot(j0 + q * Ns)] = v[q];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // v[q] = Z[lane + 64 q]

    // ---- real-split post-pass: X[k] = (Z[k] + conj(Z[256-k]))/2 - i/2 W512^k (Z[k] - conj(Z[256-k])),  power spectrum -------
#pragma unroll
    for (int q = 0; q < 4; ++q) buf[fb_slot(lane + 64 * q)] = v[q];
    __builtin_amdgcn_wave_barrier();
    f2 zc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zc[q] = buf[fb_slot((256 - lane - 64 * q) & 255)];
    const f2 z0 = buf[fb_slot(0)];
    __builtin_amdgcn_wave_barrier();               // all reads of the complex buffer are done: it becomes the power spectrum
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float zr = v[q].x, zi = v[q].y, yr = zc[q].x, yi = -zc[q].y;
        const float er = 0.5f * (zr + yr), ei = 0.5f * (zi + yi);
        const float dr = 0.5f * (zr - yr), di = 0.5f * (zi - yi);
        const float wr = tws[q].x, wi = tws[q].y;
        const float pr = wr * dr - wi * di, pi = wr * di + wi * dr;
        const float xr = er + pi, xi = ei - pr;
        pw[lane + 64 * q] = xr * xr + xi * xi;
    }
    if (lane == 0) pw[256] = (z0.x - z0.y) * (z0.x - z0.y);            // k = 256: X = Re Z[0] - Im Z[0]
    if (lane < FB_TAPS) pw[257 + lane] = 0.f;                          // taps past a filter's last bin carry weight 0: finite operands
    __builtin_amdgcn_wave_barrier();

    // ---- mel filterbank + log: filters 0..63 one per lane, filters 64..79 on four lanes each --------------------------------
    {
        const int lo = tb.mel_lo[lane];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < FB_TAPS; ++i) acc = fmaf(tb.melwt[i * NMEL + lane], pw[lo + i], acc);
        dst[lane] = logf(fmaxf(acc, 1.1920928955078125e-07f));
        const int m2 = 64 + (lane >> 2), i0 = 4 * (lane & 3);
        const int lo2 = tb.mel_lo[m2];
        float a2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) a2 = fmaf(tb.melwt[(i0 + u) * NMEL + m2], pw[lo2 + i0 + u], a2);
        a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
        if ((lane & 3) == 0) dst[m2] = logf(fmaxf(a2, 1.1920928955078125e-07f));
    }
}

template <class ST>
static void launch_fbank_t(const ST* pcm, const int* nsamp, int B, int n_max, int use_db, float target_db,
                           const FbankTables& tb, float* feats, int T_max, float* gain_scratch, int16_t* norm_out, hipStream_t s) {
    if (use_db == 1) {       // (use_db == 2: the caller has already put its own gains into gain_scratch[0 .. B))
        // gain_scratch: [B] gains followed by [B][MAX_CHUNKS + 1] chunk sums
        float* chunk_sum = gain_scratch + B;
        const int nblk = (std::min(n_max / NP_BUF, MAX_CHUNKS) + 3) / 4 + 1;
        hipLaunchKernelGGL(rms_partial_kernel<ST>, dim3(nblk, B), dim3(256), 0, s, pcm, nsamp, n_max, chunk_sum);
        hipLaunchKernelGGL(rms_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, nsamp, B, target_db, chunk_sum,
                           gain_scratch, (float*)nullptr);
    }
    if (norm_out)
        hipLaunchKernelGGL(norm_int16_kernel<ST>, dim3((n_max + 255) / 256, B), dim3(256), 0, s, pcm, nsamp, n_max,
                           gain_scratch, use_db, norm_out);
    if (T_max <= 0) return;
    hipLaunchKernelGGL(fbank_kernel<ST>, dim3((T_max + 3) / 4, B), dim3(256), 0, s, pcm, nsamp, n_max, use_db, gain_scratch, tb, feats,
                       T_max);
}

// ms_out[b] = float32 np.mean(samples ** 2) of utterance b (numpy's summation order); gain_scratch as in launch_fbank
void launch_mean_square(const void* pcm, int sample_format, const int* nsamp, int B, int n_max, float* gain_scratch,
                        float* ms_out, hipStream_t s) {
    if (B <= 0) return;
    float* chunk_sum = gain_scratch + B;
    const int nblk = (std::min(n_max / NP_BUF, MAX_CHUNKS) + 3) / 4 + 1;
    if (sample_format == 0)
        hipLaunchKernelGGL(rms_partial_kernel<int16_t>, dim3(nblk, B), dim3(256), 0, s, (const int16_t*)pcm, nsamp, n_max, chunk_sum);
    else
        hipLaunchKernelGGL(rms_partial_kernel<float>, dim3(nblk, B), dim3(256), 0, s, (const float*)pcm, nsamp, n_max, chunk_sum);
    hipLaunchKernelGGL(rms_final_kernel, dim3((B + 63) / 64), dim3(64), 0, s, nsamp, B, -20.f, chunk_sum, gain_scratch, ms_out);
}

void launch_fbank(const void* pcm, int sample_format, const int* nsamp, int B, int n_max, int use_db, float target_db,
                  const FbankTables& tb, float* feats, int T_max, float* gain_scratch, int16_t* norm_out, hipStream_t s) {
    if (B <= 0) return;
    if (sample_format == 0)
        launch_fbank_t((const int16_t*)pcm, nsamp, B, n_max, use_db, target_db, tb, feats, T_max, gain_scratch, norm_out, s);
    else
        launch_fbank_t((const float*)pcm, nsamp, B, n_max, use_db, target_db, tb, feats, T_max, gain_scratch, norm_out, s);
}


// ------------------------------------------------------------------------------------------------------------------
// MFCC = kaldi.mfcc as called by audio_featurizer.py:98-117: the 80-bin log-mel energies above, right-multiplied by the
// orthonormal DCT-II matrix [80, n_ceps] and scaled by the cepstral lifter (torchaudio: feature.matmul(dct) * lifter).
// One thread per output coefficient; the fbank row is broadcast-read, the table is tiny (L1 resident).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfcc_kernel(const float* __restrict__ fb, long n_out, int n_ceps,
                                                   const float* __restrict__ dct, const float* __restrict__ lifter,
                                                   float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const long r = i / n_ceps;
    const int c = (int)(i - r * n_ceps);
    const float* row = fb + r * NMEL;
    float acc = 0.f;
#pragma unroll 8
    for (int m = 0; m < NMEL; ++m) acc = fmaf(row[m], dct[m * n_ceps + c], acc);
    out[i] = acc * lifter[c];
}
void launch_mfcc(const float* fbank, long rows, int n_ceps, const float* dct, const float* lifter, float* out, hipStream_t s) {
    const long n = rows * n_ceps;
    if (n <= 0) return;
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fbank, n, n_ceps, dct, lifter, out);
}

// ------------------------------------------------------------------------------------------------------------------
// Linear log power spectrogram (audio_featurizer.py:73-95): float32 samples (after the optional dB gain), 20 ms frames
// (320 samples) every 10 ms, symmetric Hann window, 320-point real DFT -> 161 bins, power scaled by
// 2 / (sum(w^2) * fs) (DC and Nyquist: 1 / ...), log(. + 1e-14).  The reference computes it in float64 (numpy promotes
// the float32 frames by the float64 window); so does this kernel: one workgroup per frame, thread k = bin k, a direct DFT
// over the windowed frame in LDS with a 320-entry twiddle table (103 kFMA fp64 per frame -- noise next to the encoder).
// ------------------------------------------------------------------------------------------------------------------
static constexpr int LWIN = 320, LHOP = 160, LBIN = 161;

template <class ST>
__global__ __launch_bounds__(192) void linear_spec_kernel(const ST* __restrict__ pcm, const int* __restrict__ nsamp, int n_max,
                                                          int use_db, const float* __restrict__ gain,
                                                          const double* __restrict__ win, const double* __restrict__ tw,
                                                          double scale, float* __restrict__ feats, int T_max) {
    __shared__ double xw[LWIN], tc[LWIN], ts[LWIN];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int n = nsamp[b];
    const int T = n >= LWIN ? (n - LWIN) / LHOP + 1 : 0;
    float* dst = feats + ((size_t)b * T_max + t) * LBIN;
    if (t >= T) {                       // frames behind the utterance: zero padding
        if (tid < LBIN) dst[tid] = 0.f;
        return;
    }
    const float g = use_db ? gain[b] : 1.0f;
    const ST* src = pcm + (size_t)b * n_max + (size_t)t * LHOP;
    for (int i = tid; i < LWIN; i += 192) {
        float x;
        if (sizeof(ST) == 2) x = (float)src[i] * (1.0f / 32768.0f);      // audio.py:532-546
        else x = (float)src[i];
        if (use_db) x = x * g;                                             // float32 in-place gain (audio.py:256-264)
        xw[i] = (double)x * win[i];
        tc[i] = tw[2 * i];
        ts[i] = tw[2 * i + 1];
    }
    __syncthreads();
    const int k = tid;
    if (k >= LBIN) return;
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int j = 0; j < LWIN; ++j) {
        const double v = xw[j];
        re = fma(v, tc[idx], re);
        im = fma(v, ts[idx], im);
        idx += k;
        if (idx >= LWIN) idx -= LWIN;
    }
    double p = re * re + im * im;
    if (k == 0 || k == LBIN - 1) p = p / scale;
    else p = p * (2.0 / scale);
    dst[k] = (float)log(p + 1e-14);
}

void launch_linear_spec(const void* pcm, int sample_format, const int* nsamp, int B, int n_max, int use_db, const float* gain,
                        const double* win, const double* tw, double scale, float* feats, int T_max, hipStream_t s) {
    if (B <= 0 || T_max <= 0) return;
    if (sample_format == 0)
        hipLaunchKernelGGL(linear_spec_kernel<int16_t>, dim3(T_max, B), dim3(192), 0, s, (const int16_t*)pcm, nsamp, n_max, use_db,
                           gain, win, tw, scale, feats, T_max);
    else
        hipLaunchKernelGGL(linear_spec_kernel<float>, dim3(T_max, B), dim3(192), 0, s, (const float*)pcm, nsamp, n_max, use_db,
                           gain, win, tw, scale, feats, T_max);
}

__global__ void linear_frame_counts_kernel(const int* __restrict__ nsamp, int B, int* __restrict__ nfr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) nfr[b] = nsamp[b] >= LWIN ? (nsamp[b] - LWIN) / LHOP + 1 : 0;
}
void launch_linear_frame_counts(const int* nsamp, int B, int* nfr, hipStream_t s) {
    hipLaunchKernelGGL(linear_frame_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, nsamp, B, nfr);
}

}  // namespace masr

namespace masr {
size_t fbank_gain_scratch_floats(int B) { return (size_t)B * (MAX_CHUNKS + 2); }
}  // namespace masr
