// EXPLORATORY precision mode (masr_debug_set key 20; never the contract path): the GEMMs of gemm_f32.hip on the bf16 matrix
// pipe with SPLIT operands.  Every fp32 operand is cut into two bf16 pieces, a = a_hi + a_lo (a_hi = bf16(a), a_lo = bf16(a - a_hi):
// 16 mantissa bits between them), and a product is taken as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi with fp32 accumulation in
// v_mfma_f32_32x32x16_bf16 -- three instructions of 32 cycles cover K = 16 where the exact-fp32 v_mfma_f32_32x32x2_f32 path
// needs eight of 64 cycles: 5.3x less matrix-pipe time for a result that differs from the fp32 one by ~2e-6 relative
// (tools/mfma_bf16_layout.hip on the box: K = 2304, max error 7.8e-5 against double where the fp32 fma chain itself has 3.7e-5).
// The reference computes this path in fp32, so the contract line (dtype f32) never uses this file; bench.py reports the mode as
// a separate `extra` entry with dtype "bf16x3" and the parity suite runs in it too (tests/test_gpu_bf16x3.py).
//
// Same tiling, staging and epilogue as gemm_f32_kernel: BK = 32 slabs, register-staged global -> LDS with one barrier per slab,
// double-buffered; the split happens in registers between the global load and the LDS store (both operands arrive as fp32: the
// weights keep their fp32 layout, nothing is pre-converted).  LDS rows hold 32 bf16 (64 B) padded to 80 B: the 16-byte fragment
// reads of 16 consecutive rows then tile all 64 banks.  Operand layout of the 32x32x16 bf16 MFMA: lane l holds
// A[i = l & 31][k = 8 * (l >> 5) .. + 7]; C/D as in the f32 forms.
#include "common.h"

namespace masr {

namespace {

constexpr int XBK = 32;
constexpr int XLD = 40;          // padded LDS row, in bf16 elements (80 bytes)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ float xsilu(float x) { return x / (1.0f + expf(-x)); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// two fp32 -> packed bf16 pair (v_cvt_pk_bf16_f32, round to nearest even) as a dword
__device__ __forceinline__ unsigned cvt2(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const unsigned*>(&h);
}
// four consecutive fp32 -> their hi and lo bf16 pieces, packed two per dword (12 VALU instructions)
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
    hi[0] = cvt2(v[0], v[1]);
    hi[1] = cvt2(v[2], v[3]);
    lo[0] = cvt2(v[0] - __uint_as_float(hi[0] << 16), v[1] - __uint_as_float(hi[0] & 0xFFFF0000u));
    lo[1] = cvt2(v[2] - __uint_as_float(hi[1] << 16), v[3] - __uint_as_float(hi[1] & 0xFFFF0000u));
}

template <int BM, int BN, int WM, int WN, int AMODE>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16x3_kernel(GemmArgs p) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    constexpr int NT = 64 * WM * WN;
    constexpr int RPP = NT / 8;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / RPP;
    constexpr int WL = BN / RPP;
    extern __shared__ __align__(16) unsigned short xsm[];
    unsigned short* Ah = xsm;                          // [2][BM][XLD]
    unsigned short* Al = Ah + 2 * BM * XLD;
    unsigned short* Wh = Al + 2 * BM * XLD;            // [2][BN][XLD]
    unsigned short* Wl = Wh + 2 * BN * XLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (as gemm_f32_kernel)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = (bid / nbn) * BM, bn = (bid % nbn) * BN;

    const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
    const float* aptr[AL];
    bool aok[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = bm + lrow + RPP * i;
        aok[i] = m < p.M;
        const int mm = aok[i] ? m : 0;
        if (AMODE == A_PLAIN) {
            aptr[i] = p.A + (size_t)mm * p.lda + lc4;
        } else {
            const int f2 = mm % p.F2, bt = mm / p.F2, t2 = bt % p.T2, b = bt / p.T2;
            aptr[i] = p.A + (((size_t)b * p.T1 + 2 * t2) * p.F1 + 2 * f2) * p.Cc + lc4;
        }
    }
    const float* wptr[WL];
    bool wok[WL];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        const int n = bn + lrow + RPP * i;
        wok[i] = n < p.N;
        wptr[i] = p.W + (size_t)(wok[i] ? n : 0) * p.K + lc4;
    }
    // two register sets: while slab kt is multiplied out of LDS, slab kt + 1 waits in one set for its turn to be split and stored,
    // and slab kt + 2 is on its way from memory into the other -- the multiply of a slab takes ~400 cycles per wave here (a fifth
    // of the fp32 kernel's), less than one memory latency, so a prefetch distance of one slab leaves the loop waiting on loads
    f32x4 ra0[AL], rw0[WL], ra1[AL], rw1[WL];
    auto load_slab = [&](int kt, f32x4 (&areg)[AL], f32x4 (&wreg)[WL]) {
        size_t aoff;
        if (AMODE == A_PLAIN) {
            aoff = (size_t)kt * XBK;
        } else {
            const int cb = kt / 9, pos = kt - 9 * cb, kh = pos / 3, kw = pos - 3 * kh;
            aoff = ((size_t)kh * p.F1 + kw) * p.Cc + (size_t)cb * XBK;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i)
            areg[i] = aok[i] ? *reinterpret_cast<const f32x4*>(aptr[i] + aoff) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WL; ++i)
            wreg[i] = wok[i] ? *reinterpret_cast<const f32x4*>(wptr[i] + (size_t)kt * XBK) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_slab = [&](int buf, const f32x4 (&areg)[AL], const f32x4 (&wreg)[WL]) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            u32x2 hi, lo;
            split4(areg[i], hi, lo);
            const int o = (buf * BM + lrow + RPP * i) * XLD + lc4;
            *reinterpret_cast<u32x2*>(&Ah[o]) = hi;
            *reinterpret_cast<u32x2*>(&Al[o]) = lo;
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            u32x2 hi, lo;
            split4(wreg[i], hi, lo);
            const int o = (buf * BN + lrow + RPP * i) * XLD + lc4;
            *reinterpret_cast<u32x2*>(&Wh[o]) = hi;
            *reinterpret_cast<u32x2*>(&Wl[o]) = lo;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int KT = p.K / XBK;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    auto multiply = [&](int buf) {
        const int ao = (buf * BM + wm * (BM / WM) + frow) * XLD + fk;
        const int wo = (buf * BN + wn * (BN / WN) + frow) * XLD + fk;
#pragma unroll
        for (int s = 0; s < XBK / 16; ++s) {
            bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&Ah[ao + m * 32 * XLD + 16 * s]);
                al[m] = *reinterpret_cast<const bf16x8*>(&Al[ao + m * 32 * XLD + 16 * s]);
            }
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                wh[n] = *reinterpret_cast<const bf16x8*>(&Wh[wo + n * 32 * XLD + 16 * s]);
                wl[n] = *reinterpret_cast<const bf16x8*>(&Wl[wo + n * 32 * XLD + 16 * s]);
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], wh[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wl[n], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wh[n], acc[m][n], 0, 0, 0);
                }
        }
    };
    // one iteration: slab kt out of LDS buffer kt & 1; `ld` receives slab kt + 2, `st` holds slab kt + 1
    auto iteration = [&](int kt, f32x4 (&lda)[AL], f32x4 (&ldw)[WL], const f32x4 (&sta)[AL], const f32x4 (&stw)[WL]) {
        if (kt + 2 < KT) load_slab(kt + 2, lda, ldw);
        __builtin_amdgcn_sched_barrier(0);       // the loads go out first
        multiply(kt & 1);
        __builtin_amdgcn_sched_barrier(0);
        // (measured: the split + store of slab kt + 1 placed BEFORE this slab's multiply, free to interleave, is slower --
        //  634 vs 597 us on conv2; four waves with 64 x 64 per wave, fewer fragment reads per MFMA: 719 us)
        if (kt + 1 < KT) store_slab((kt + 1) & 1, sta, stw);
        __syncthreads();
    };
    load_slab(0, ra0, rw0);
    if (KT > 1) load_slab(1, ra1, rw1);
    store_slab(0, ra0, rw0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
        iteration(kt, ra0, rw0, ra1, rw1);
        if (kt + 1 < KT) iteration(kt + 1, ra1, rw1, ra0, rw0);
    }

    // ---- epilogue (EPI_STD of gemm_f32_kernel) ----------------------------------------------------------------------------
    const int ccol = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = bn + wn * (BN / WN) + n * 32 + ccol;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(bm + wm * (BM / WM) + m * 32 + (r & 3) + 8 * (r >> 2) + rbase, p.M - 1);
                res[r] = p.R ? p.R[(size_t)row * p.ldr + col] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm + wm * (BM / WM) + m * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                float v = acc[m][n][r] + (p.bias_after_alpha ? 0.f : bv);
                if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == ACT_SILU) v = xsilu(v);
                if (p.mask_tp > 0) {
                    const int rc = min(row, p.M - 1);
                    const int b = rc / p.mask_tp, t = rc - b * p.mask_tp;
                    if (4 * t >= p.lens[b]) v = 0.f;
                }
                v = res[r] + v * p.alpha + (p.bias_after_alpha ? bv : 0.f);
                if (row < p.M) p.C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AMODE>
void launch_x(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)2 * 2 * (BM + BN) * XLD * sizeof(unsigned short);
    auto k = gemm_bf16x3_kernel<BM, BN, WM, WN, AMODE>;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, attr);
    hipLaunchKernelGGL(k, dim3(nbm * nbn), dim3(64 * WM * WN), lds, s, a);
}

}  // namespace

static int g_x3_waves = 8;
void set_gemm_bf16x3_waves(int n) { g_x3_waves = n; }

// true when the launch was taken (K a multiple of 32; plain row-major A or the conv2 gather; the standard epilogue)
bool launch_gemm_bf16x3(const GemmArgs& a, int amode, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K % XBK != 0) return false;
    if (amode == A_CONV2) {
        if ((long)((a.M + 127) / 128) * ((a.N + 127) / 128) < 200) launch_x<64, 64, 2, 2, A_CONV2>(a, s);
        else if (g_x3_waves == 4) launch_x<128, 128, 2, 2, A_CONV2>(a, s);     // 64 x 64 per wave: 8 fragment reads per 12 MFMAs
        else launch_x<128, 128, 2, 4, A_CONV2>(a, s);                          // 64 x 32 per wave: 6 per 6
        return true;
    }
    if (amode != A_PLAIN) return false;
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long t64 = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    if (t128 >= 384) launch_x<128, 128, 2, 4, A_PLAIN>(a, s);
    else if (t64 >= 200) launch_x<64, 128, 2, 2, A_PLAIN>(a, s);
    else launch_x<64, 64, 2, 2, A_PLAIN>(a, s);
    return true;
}

}  // namespace masr
