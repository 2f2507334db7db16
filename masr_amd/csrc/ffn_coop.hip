// One-chunk feed-forward slice for FEW ROWS (streaming chunk steps of <= 16 streams, one offline utterance): the d_ff-split launch
// in which a workgroup owns 32 rows x ONE chunk of 128 hidden units (reference conformer/positionwise.py:30-37; the partial sums
// are added by ffn_reduce_kernel).
//
// ffn_pc.hip's producer / consumer specialisation is built for 16 chunks per workgroup: four waves multiply the hidden tile of
// chunk k while the other four multiply chunk k - 1 into the output.  With ONE chunk per workgroup the two roles run one after the
// other -- 128 dependent MFMAs on four waves, then 128 on the other four: 2 x 3.4 us of a 14 us launch that sits 24 times on the
// critical path of a chunk step.  Here all eight waves work on both products:
//   GEMM 1   hidden[32 rows, 128 units] = LN(x) . W1c^T      wave w = (row half w & 1, unit group w >> 1): 16 rows x 32 units as two
//            16 x 16 tiles of v_mfma_f32_16x16x4_f32 over K = 256 (128 MFMAs of 32 cycles = the time of 64 of the 32 x 32 x 2 form),
//            no K split and no partial-sum exchange; + bias, SiLU -> LDS
//   GEMM 2   partial[32 rows, 256] = hidden . W2c^T          wave w = output columns 32 w .. 32 w + 31, K = 128: 64 MFMAs 32 x 32 x 2
// so each product costs 1.7 us of matrix pipe per SIMD instead of 3.4.  Every weight fragment of both products (W1c: 32 x 16
// bytes per lane, W2c: 16 x 16 bytes per lane, from packed copies in fragment order) and the rows for the LayerNorm are
// requested before anything waits.  Same fp32 arithmetic; only the order of the K summation of GEMM 1 differs from ffn_pc.hip.
// Round 2 tried a cooperative kernel that split K of GEMM 1 across wave pairs and summed through LDS (slower); this one does not.
#include "common.h"

namespace masr {

static constexpr int FC_XLD = 256 + 4;
static constexpr int FC_HLD = 128 + 4;

__global__ __launch_bounds__(512) void ffn_coop_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2, int M, int dff,
                                                       float eps, int affine, float* __restrict__ partial) {
    __shared__ __align__(16) float xn[32 * FC_XLD];      // LayerNorm(x) tile
    __shared__ __align__(16) float hs[32 * FC_HLD];      // hidden tile of this chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 32, chunk = blockIdx.y;

    // ---- every global load of the kernel ---------------------------------------------------------------------------------------
    f32x4 v4[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = min(row0 + wave * 4 + rr, M - 1);
        v4[rr] = *reinterpret_cast<const f32x4*>(x + (size_t)row * 256 + lane * 4);
    }
    const f32x4 gw = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
    const f32x4 gb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
    __builtin_amdgcn_sched_barrier(0);      // the LayerNorm's rows first: vmcnt retires in order, their wait leaves the rest in flight
    // Both weight operands come from PACKED copies in fragment order -- one contiguous 1 KB per wave and load instruction (read
    // from the row-major matrices in operand layout, the same fragments are 16 / 32 cache lines per instruction: the kernel took
    // 22 us instead of the producer / consumer kernel's 14) -- through raw buffer loads (descriptor + 16 * lane + scalar offset):
    //   GEMM 1 operand B: lane (n = lane & 15, kq = lane >> 4) holds W1[chunk * 128 + 32 ug + 16 tt + n][16 s + 4 kq ..], s = 0..15,
    //                     packed [chunk][ug][tt][s][lane][4] by pack_ffn_coop_w1_kernel
    //   GEMM 2 operand B: lane (n = lane & 31, h = lane >> 5) holds W2[32 wave + n][chunk * 128 + 8 g' + 4 h ..], g' = 0..15 = ffn_pc.hip's
    //                     packed W2 [chunk][idx][slab j][group g][lane][4] with idx = wave >> 1, j = 2 (g' >> 2) + (wave & 1), g = g' & 3
    const int m16 = lane & 15, kq = lane >> 4;
    const int rh = wave & 1, ug = wave >> 1;
    const unsigned lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w1), 0, dff * 256 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w2), 0, dff * 256 * 4, 0x00020000);
    f32x4 wf1[2][16];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            wf1[tt][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        r1, lane16, (unsigned)((((chunk * 4 + ug) * 2 + tt) * 16 + s)) * 1024u, 0));
    float bias1[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) bias1[tt] = b1[chunk * 128 + 32 * ug + 16 * tt + m16];
    const int frow = lane & 31, fh = lane >> 5;
    f32x4 wf2[16];
#pragma unroll
    for (int g = 0; g < 16; ++g)
        wf2[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               r2, lane16, (unsigned)((((chunk * 4 + (wave >> 1)) * 8 + 2 * (g >> 2) + (wave & 1)) * 4 + (g & 3))) * 1024u, 0));
    __builtin_amdgcn_sched_barrier(0);      // all of them in flight before the first wait

    // ---- LayerNorm (or the Squeezeformer's adaptive scale / bias) of my four rows -> xn ------------------------------------------
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int lr = wave * 4 + rr;
        const f32x4 v = v4[rr];
        f32x4 o;
        if (affine) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = gw[k] * v[k] + gb[k];
        } else {
            const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float var = wave_sum_dpp(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + eps);
            o[0] = d0 * rstd * gw[0] + gb[0];
            o[1] = d1 * rstd * gw[1] + gb[1];
            o[2] = d2 * rstd * gw[2] + gb[2];
            o[3] = d3 * rstd * gw[3] + gb[3];
        }
        if (row0 + lr >= M) o = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&xn[lr * FC_XLD + lane * 4]) = o;
    }
    __syncthreads();

    // ---- GEMM 1: 16 rows x 32 units per wave on 16 x 16 x 4 tiles ------------------------------------------------------------------
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const float* xa = xn + (16 * rh + m16) * FC_XLD + 4 * kq;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xa + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wf1[0][s][j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wf1[1][s][j], acc[1], 0, 0, 0);
            }
        }
        // D layout: lane l, register i -> row 4 (l >> 4) + i, column l & 15
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hv = acc[tt][i] + bias1[tt];
                hs[(16 * rh + 4 * kq + i) * FC_HLD + 32 * ug + 16 * tt + m16] = hv * __builtin_amdgcn_rcpf(1.0f + __expf(-hv));      // (ffn_pc.hip's form)
            }
    }
    __syncthreads();

    // ---- GEMM 2: 32 rows x 32 output columns per wave, K = 128 hidden units ---------------------------------------------------------
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    {
        const float* ha = hs + frow * FC_HLD + 4 * fh;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ha + 8 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], wf2[g][q], acc2, 0, 0, 0);
        }
    }
    float* pp = partial + (size_t)chunk * M * 256;
    const int col = 32 * wave + frow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (row < M) pp[(size_t)row * 256 + col] = acc2[r];
    }
}

// W1 [dff, 256] -> [chunk][ug][tt][s][lane][4] = W1[chunk * 128 + 32 ug + 16 tt + (lane & 15)][16 s + 4 (lane >> 4) + q]
__global__ __launch_bounds__(256) void pack_ffn_coop_w1_kernel(const float* __restrict__ w1, float* __restrict__ p, int dff) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)dff * 256) return;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), s = (int)((e >> 8) & 15), tt = (int)((e >> 12) & 1),
              ug = (int)((e >> 13) & 3), chunk = (int)(e >> 15);
    p[e] = w1[(size_t)(chunk * 128 + 32 * ug + 16 * tt + (lane & 15)) * 256 + 16 * s + 4 * (lane >> 4) + q];
}
void launch_pack_ffn_coop_w1(const float* w1, float* p, int dff, hipStream_t s) {
    const size_t n = (size_t)dff * 256;
    hipLaunchKernelGGL(pack_ffn_coop_w1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w1, p, dff);
}

// w1p: launch_pack_ffn_coop_w1's copy of W1; w2p: ffn_pc.hip's packed copy of W2 (launch_pack_ffn_pc); partial [dff / 128][M][256];
// the caller runs the reduction
void launch_ffn_coop(const float* x, const float* lnw, const float* lnb, const float* w1, const float* b1, const float* w2, int M,
                     int dff, float eps, int affine, float* partial, hipStream_t s) {
    hipLaunchKernelGGL(ffn_coop_kernel, dim3((M + 31) / 32, dff / 128), dim3(512), 0, s, x, lnw, lnb, w1, b1, w2, M, dff, eps, affine,
                       partial);
}

}  // namespace masr
