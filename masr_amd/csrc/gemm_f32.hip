// fp32 GEMM on the CDNA4 matrix cores: C[M,N] = epilogue(A[M,K] * W[N,K]^T).
//
// * v_mfma_f32_32x32x2_f32 (exact fp32, == an fmaf chain): the reference computes this path in
//   fp32 (torch Linear / Conv on CPU), north_star asks for 1e-3 fp32 logit parity.
// * 256 threads = 4 wave64 per workgroup, WM x WN wave grid, each wave owns TM x TN tiles of 32x32.
// * K is consumed in BK=32 slabs, register-staged global->LDS with one barrier per slab
//   (double-buffered LDS).  LDS rows are padded to 36 floats so the ds_read_b128 fragment reads
//   (row = lane&31, 16-byte column = lane>>5) are bank-conflict free (row stride 144 B = 9 slots).
// * Fragment trick: MFMA sums over k in any order, so lane half h=(lane>>5) takes the 4 consecutive
//   k values {8g+4h .. 8g+4h+3} of each 8-wide k group with ONE 16-byte LDS read for A and for W.
// * A operand modes: plain row-major, or implicit-GEMM gather for the 3x3/stride-2 subsampling
//   conv over channels-last activations (reference conformer/subsampling.py:86-110), K ordered
//   [32-channel block][kh][kw][channel] so that overlapping window columns are re-read while still cached.
// * Epilogue: bias, ReLU/SiLU, alpha, residual, row masking.  (The K = 256 projections of the layers use
//   rowgemm.hip, the FFN ffn_pc.hip; this kernel serves conv2, the embed projection, the positional-key
//   precompute and the full-probability CTC head.)
#include "common.h"

namespace masr {

static constexpr int BK = 32;
static constexpr int LDP = 36;   // padded LDS row (floats)

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// EPI_SPLITK: blockIdx.y owns a contiguous range of K slabs and stores its raw partial tile to p.C + blockIdx.y*M*ldc
// (p.ksplit slabs per range); splitk_reduce_kernel applies the epilogue.  For M so small that the tile grid cannot fill
// the chip while K is deep (embed projection of a streaming chunk step: M = 256, K = 4864).
template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmArgs p) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
    constexpr int NT = 64 * WM * WN;
    constexpr int RPP = NT / 8;   // slab rows staged per pass of the workgroup (8 threads x float4 per 32-wide row)
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / RPP;  // float4 loads per thread for the A slab
    constexpr int WL = BN / RPP;
    extern __shared__ __align__(16) float smem[];
    float* As = smem;                       // [2][BM][LDP]
    float* Ws = smem + 2 * BM * LDP;        // [2][BN][LDP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: consecutive block ids land on different XCDs (id % 8); give every XCD a
    // contiguous range of M-tiles so that the W panel and neighbouring A rows stay in that XCD's L2.
    const int nbm = (p.M + BM - 1) / BM;
    const int nbn = (p.N + BN - 1) / BN;
    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    if (p.skip_rps <= 0) {          // (tiles are skipped by sequence length: contiguous per-XCD ranges would leave the XCDs of the
                                    //  short sequences idle -- plain order then, every XCD sees every sequence)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = (bid / nbn) * BM;
    const int bn = (bid % nbn) * BN;
    if (p.skip_rps > 0 && p.lens) {         // a tile of padded frames only (whole workgroup, before any barrier)
        const int span = min(bm + BM, p.M) - 1 - bm;
        const int b0 = bm / p.skip_rps, r0 = bm - b0 * p.skip_rps;
        if (r0 + span < p.skip_rps && 4 * (r0 / p.skip_div) >= p.lens[b0]) return;
    }

    // ---- per-thread global source pointers ------------------------------------------------
    const int lrow = tid >> 3;          // 0..RPP-1
    const int lc4 = (tid & 7) * 4;      // float offset inside the 32-wide slab
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
            const int f2 = mm % p.F2;
            const int bt = mm / p.F2;
            const int t2 = bt % p.T2;
            const int b = bt / p.T2;
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

    f32x4 areg[AL], wreg[WL];
    auto load_slab = [&](int kt) {
        size_t aoff;
        if (AMODE == A_PLAIN) {
            aoff = (size_t)kt * BK;
        } else {
            // K is ordered [channel block of 32][kh][kw][32 channels] (weights re-laid out at load): slab kt = window position
            // kt % 9 of channel block kt / 9
            const int cb = kt / 9, pos = kt - 9 * cb;
            const int kh = pos / 3, kw = pos - 3 * kh;
            aoff = ((size_t)kh * p.F1 + kw) * p.Cc + (size_t)cb * BK;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i)
            areg[i] = aok[i] ? *reinterpret_cast<const f32x4*>(aptr[i] + aoff) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WL; ++i)
            wreg[i] = wok[i] ? *reinterpret_cast<const f32x4*>(wptr[i] + (size_t)kt * BK) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + RPP * i) * LDP + lc4]) = areg[i];
#pragma unroll
        for (int i = 0; i < WL; ++i)
            *reinterpret_cast<f32x4*>(&Ws[(buf * BN + lrow + RPP * i) * LDP + lc4]) = wreg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int kt_lo = EPI == EPI_SPLITK ? (int)blockIdx.y * p.ksplit : 0;
    const int KT = EPI == EPI_SPLITK ? min(kt_lo + p.ksplit, p.K / BK) : p.K / BK;
    load_slab(kt_lo);
    store_slab(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fcol = (lane >> 5) * 4;
    for (int kt = kt_lo; kt < KT; ++kt) {
        const int buf = (kt - kt_lo) & 1;
        if (kt + 1 < KT) load_slab(kt + 1);
        __builtin_amdgcn_sched_barrier(0);       // the loads stay ahead of this slab's MFMAs (the scheduler would sink them)
        const float* Ab = &As[(buf * BM + wm * (BM / WM) + frow) * LDP + fcol];
        const float* Wb = &Ws[(buf * BN + wn * (BN / WN) + frow) * LDP + fcol];
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) af[m] = *reinterpret_cast<const f32x4*>(Ab + m * 32 * LDP + g * 8);
#pragma unroll
            for (int n = 0; n < TN; ++n) wf[n] = *reinterpret_cast<const f32x4*>(Wb + n * 32 * LDP + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][s], wf[n][s], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < KT) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const int ccol = lane & 31;
    const int rbase = 4 * (lane >> 5);
    if (EPI == EPI_SPLITK) {
        float* cp = p.C + (size_t)blockIdx.y * p.M * p.ldc;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = bn + wn * (BN / WN) + n * 32 + ccol;
            if (col >= p.N) continue;
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = bm + wm * (BM / WM) + m * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                    if (row < p.M) cp[(size_t)row * p.ldc + col] = acc[m][n][r];
                }
        }
    }
    if (EPI == EPI_STD) {
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = bn + wn * (BN / WN) + n * 32 + ccol;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                float res[16];   // residual loads before the stores (R may alias C)
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
                    else if (p.act == ACT_SILU) v = silu_f(v);
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
}

// waves per workgroup of the conv2 implicit GEMM on 128x128 tiles: 8 (default) or 4 (masr_debug_set key 17).  In one kernel
// trace with both shapes alternating (tools/gemm_waves_trace.py): 1 467 vs 1 486 us on average, 1 437 vs 1 480 us at best.
static int g_gemm_waves = 8;
static int g_conv2_mid_fill = 50;     // 128 streams: 3.26 -> 3.19 ms per chunk call (tools/chunk_lat.py MASR_AB=33:0,33:50)
void set_conv2_mid_fill(int pct) { g_conv2_mid_fill = pct; }
void set_gemm_waves(int n) { g_gemm_waves = n; }

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
static void launch_t(const GemmArgs& a, hipStream_t s) {
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDP * sizeof(float);
    auto k = gemm_f32_kernel<BM, BN, WM, WN, AMODE, EPI>;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, attr);
    hipLaunchKernelGGL(k, dim3(nbm * nbn, EPI == EPI_SPLITK ? a.nsplit : 1), dim3(64 * WM * WN), lds, s, a);
}

// out = R + alpha * act(sum_s partial[s] + bias) [+ bias after alpha]; partials added in ascending s (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p, const float* __restrict__ partial, float* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.M * p.N) return;
    const int row = (int)(i / p.N), col = (int)(i - (size_t)row * p.N);
    float acc = 0.f;
    for (int sp = 0; sp < p.nsplit; ++sp) acc += partial[((size_t)sp * p.M + row) * p.N + col];
    const float bv = p.bias ? p.bias[col] : 0.f;
    float v = acc + (p.bias_after_alpha ? 0.f : bv);
    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == ACT_SILU) v = silu_f(v);
    const float r = p.R ? p.R[(size_t)row * p.ldr + col] : 0.f;
    out[(size_t)row * p.ldc + col] = r + v * p.alpha + (p.bias_after_alpha ? bv : 0.f);
}

void launch_gemm(const GemmArgs& a, int amode, int epi, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return;
    if (amode == A_CONV2 && epi == EPI_SPLITK) {      // caller set a.C = partial buffer, a.nsplit, a.ksplit
        launch_t<64, 64, 2, 2, A_CONV2, EPI_SPLITK>(a, s);
        return;
    }
    if (amode == A_CONV2) {
        // few output rows (streaming chunk steps): 64x64 tiles so that the grid still covers the chip
        const long t128c = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        // mid sizes (the 128-stream chunk step: 608 tiles of 128x128 on 512 resident slots = two rounds, the second at 19 %):
        // 128x64 tiles (four waves, the tile count doubles) when the last round of the 128x128 grid would be under g_conv2_mid_fill
        // percent full (masr_debug_set key 33; 0 = never)
        const long rounds = (t128c + 511) / 512;
        const bool thin_tail = g_conv2_mid_fill > 0 && t128c >= 200 && t128c <= 1024 &&
                               (t128c - (rounds - 1) * 512) * 100 < (long)g_conv2_mid_fill * 512;
        if (t128c < 200) launch_t<64, 64, 2, 2, A_CONV2, EPI_STD>(a, s);
        else if (thin_tail) launch_t<64, 128, 2, 2, A_CONV2, EPI_STD>(a, s);
        // 8 waves (2 x 4 grid, 64 x 32 per wave) on the 128x128 tile: two workgroups per CU = four waves per SIMD cover each
        // other's slab barriers; 1 494 -> 1 457 us at B = 32 x 10 s by HIP events.  (4 x 2 grid: 1 488 us; 16 waves as a 4 x 4 grid: 1 624 us; 128x256 /
        // 256x128 tiles with 8 waves, one workgroup per CU: 1 540 us.)
        else if (g_gemm_waves == 8) launch_t<128, 128, 2, 4, A_CONV2, EPI_STD>(a, s);
        else launch_t<128, 128, 2, 2, A_CONV2, EPI_STD>(a, s);
        return;
    }
    if (epi == EPI_SPLITK) {            // caller set a.C = partial buffer, a.nsplit, a.ksplit
        // many rows (the offline embed projection: 248 tiles of 64x128 = one 4-wave workgroup per CU): the wide tile, so that
        // the split doubles the waves per SIMD instead of the LDS traffic per MFMA
        // (four waves: as a 2 x 4 grid of eight waves, 32 x 32 per wave, this launch is slower -- 183.6 vs 177.4 us in one
        // kernel trace, tools/gemm_waves_trace.py -- the LDS reads per MFMA double)
        if (a.nsplit >= 4 && (long)((a.M + 127) / 128) * ((a.N + 127) / 128) >= 100) launch_t<128, 128, 2, 4, A_PLAIN, EPI_SPLITK>(a, s);
        else if ((long)((a.M + 63) / 64) * ((a.N + 127) / 128) >= 200) launch_t<64, 128, 2, 2, A_PLAIN, EPI_SPLITK>(a, s);
        else launch_t<64, 64, 2, 2, A_PLAIN, EPI_SPLITK>(a, s);
        return;
    }
    // Tile choice: fill >= 256 CUs.  128x128 when that already yields enough workgroups,
    // otherwise 64x128 / 64x64 (N = 256 projections at M = B*T' ~ 8k rows).
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const long t64 = (long)((a.M + 63) / 64) * ((a.N + 127) / 128);
    if (t128 >= 384) launch_t<128, 128, 2, 2, A_PLAIN, EPI_STD>(a, s);
    else if (t64 >= 200) launch_t<64, 128, 2, 2, A_PLAIN, EPI_STD>(a, s);
    else launch_t<64, 64, 2, 2, A_PLAIN, EPI_STD>(a, s);
}

void launch_gemm_splitk(const GemmArgs& a, float* partial, int nsplit, hipStream_t s, int amode) {
    if (a.M <= 0 || a.N <= 0) return;
    GemmArgs b = a;
    const int kts = a.K / BK;
    b.ksplit = (kts + nsplit - 1) / nsplit;
    b.nsplit = (kts + b.ksplit - 1) / b.ksplit;          // every range owns at least one slab
    b.C = partial;
    b.ldc = a.N;
    launch_gemm(b, amode, EPI_SPLITK, s);
    GemmArgs r = a;
    r.nsplit = b.nsplit;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)a.M * a.N + 255) / 256)), dim3(256), 0, s, r, partial, a.C);
}

}  // namespace masr
