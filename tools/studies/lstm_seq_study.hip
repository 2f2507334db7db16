// Study (round 6), MEASURED AND REJECTED: the DeepSpeech2 recurrence of a layer as ONE cooperative launch -- every workgroup keeps
// its slice of W_hh in registers for all T steps, the workgroups of a direction meet at a barrier in global memory after each
// step -- against the product's one launch per timestep (masr_amd/csrc/lstm.hip).  This file is the part that was added to
// lstm.hip (it uses its sigm() / wave_sum_dpp() and was called from ds2_forward in place of the step loop; outputs bit-identical
// to the step launches for B = 1, 4, 16, 32).  ms per encoder forward (5 layers), one box, step launches -> sequence launch:
//     B = 1,  T' = 208:   6.0 -> 101   (barrier with __threadfence() on both sides: buffer_wbl2 + buffer_inv by every wave, 95 us / step)
//                             ->  18.6 (h through agent-scope (sc1) accesses, ONE counter per direction: 128 - 256 atomics on one address)
//                             ->  10.0 (a flag per workgroup, wave 0 reads all flags: 9.6 us / step against 5.7 us per launched step)
//     B = 32, T' = 248:  25.7 -> 139 -> 25.5 -> 26.4        B = 16: 17.0 -> 112 -> 15.4 -> 16.4        B = 4: 10.5 -> 134 -> 33 -> 16.1
// Why: across XCDs every hand-over is a round trip to the memory side (~2 us): h store + acknowledge, flag store, flag poll, h
// load are four of them in sequence per step, and the command processor's kernel boundary (which also re-reads 64 - 128 KB of
// W_hh per workgroup from L2 / Infinity Cache) costs less than that.  What would be left to try: the step number carried IN the
// h words (no flags: two round trips per step).

// ---- the whole sequence of a layer in ONE launch (round 6) ---------------------------------------------------------------------
// The per-step kernels above re-read their slice of W_hh (64 / 128 KB per workgroup) from L2 / Infinity Cache at every step and
// pay a launch per step.  Here a workgroup keeps its slice IN REGISTERS for all T steps and the workgroups of a direction meet
// at a barrier in global memory after every step (a counter per direction).  The h rows are the only data that crosses
// workgroups: they are written and read with AGENT-SCOPE accesses (cache-policy bit sc1: past the XCD's L2, which is not
// coherent with the other seven), so the barrier needs no cache write-back / invalidate -- with __threadfence() on both sides
// (buffer_wbl2 + buffer_inv by every wave) a step took 95 us instead of 6.  Launched cooperatively: all workgroups are resident.
// Same arithmetic, same order of operations as the step kernels -> bit-identical outputs (tests/test_gpu_parity.py).
static constexpr int LSTM_SC1 = 16;                       // cache policy of the raw buffer accesses: bit 4 = sc1 (gfx940+)
__device__ __forceinline__ f32x4 h_load4(__amdgpu_buffer_rsrc_t rs, size_t float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_index * 4), 0, LSTM_SC1));
}
__device__ __forceinline__ float h_load1(__amdgpu_buffer_rsrc_t rs, size_t float_index) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(float_index * 4), 0, LSTM_SC1));
}
__device__ __forceinline__ void h_store1(__amdgpu_buffer_rsrc_t rs, size_t float_index, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)(float_index * 4), 0, LSTM_SC1);
}
// The barrier: workgroup i of a direction publishes the step it has finished in flags[i] (one agent-scope store), wave 0 of
// every workgroup reads all the direction's flags (64 per load instruction) until none is behind.  No atomics: 128 - 256
// atomic adds on ONE address are serialised where they execute (10 us per step with a counter, measured).
__device__ __forceinline__ void lstm_dir_barrier(unsigned* flags, int nblk, unsigned epoch) {
    __builtin_amdgcn_s_waitcnt(0);                       // this wave's h stores have been acknowledged
    __syncthreads();
    if (threadIdx.x < 64) {
        const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(flags, 0, (unsigned)nblk * 4u, 0x00020000);
        if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(epoch, frs, (int)blockIdx.x * 4, 0, LSTM_SC1);
        for (;;) {
            bool ok = true;
            for (int i = threadIdx.x; i < nblk; i += 64)
                ok = ok && __builtin_amdgcn_raw_buffer_load_b32(frs, i * 4, 0, LSTM_SC1) >= epoch;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int H>
__global__ __launch_bounds__(256) void lstm_seq_kernel(const float* __restrict__ gx, const float* __restrict__ whh, float* hbuf,
                                                       float* __restrict__ c, float* __restrict__ out,
                                                       const int* __restrict__ lens, unsigned* bar, int B, int T, int ndir) {
    constexpr int PL = H / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int j = blockIdx.x * 4 + wave;
    const size_t hsz = (size_t)ndir * B * H;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hbuf, 0, (unsigned)(2 * hsz * sizeof(float)), 0x00020000);
    float w[4][PL];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wr = whh + ((size_t)dir * 4 * H + (size_t)g * H + j) * H + lane * PL;
#pragma unroll
        for (int k = 0; k < PL; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wr + k);
            w[g][k] = v[0]; w[g][k + 1] = v[1]; w[g][k + 2] = v[2]; w[g][k + 3] = v[3];
        }
    }
    for (int step = 0; step < T; ++step) {
        const size_t h_prev = (size_t)(step & 1) * hsz, h_next = (size_t)((step + 1) & 1) * hsz;      // float offsets in hbuf
        const int t = dir ? T - 1 - step : step;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int nb = min(64, B - b0);
            float mine[4] = {0.f, 0.f, 0.f, 0.f};
            for (int bb = 0; bb < nb; ++bb) {
                const size_t hp = h_prev + ((size_t)dir * B + b0 + bb) * H + lane * PL;
                float hv[PL];
#pragma unroll
                for (int k = 0; k < PL; k += 4) {
                    const f32x4 v = h_load4(hrs, hp + k);
                    hv[k] = v[0]; hv[k + 1] = v[1]; hv[k + 2] = v[2]; hv[k + 3] = v[3];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < PL; ++k) a = fmaf(w[g][k], hv[k], a);
                    a = wave_sum_dpp(a);
                    if (lane == bb) mine[g] = a;
                }
            }
            if (lane < nb) {
                const int b = b0 + lane;
                const size_t sidx = ((size_t)dir * B + b) * H + j;
                const bool active = !lens || t < lens[b];
                const float* gr = gx + ((size_t)b * T + t) * (ndir * 4 * H) + (size_t)dir * 4 * H + j;
                const float gi = gr[0] + mine[0], gf = gr[H] + mine[1], gg = gr[2 * H] + mine[2], go = gr[3 * H] + mine[3];
                const float c_old = c[sidx];
                const float c_new = sigm(gf) * c_old + sigm(gi) * tanhf(gg);
                const float h_new = sigm(go) * tanhf(c_new);
                float* o = out + ((size_t)b * T + t) * (ndir * H) + dir * H + j;
                if (active) {
                    c[sidx] = c_new;
                    h_store1(hrs, h_next + sidx, h_new);
                    *o = h_new;
                } else {
                    h_store1(hrs, h_next + sidx, h_load1(hrs, h_prev + sidx));
                    *o = 0.f;
                }
            }
        }
        if (step + 1 < T) lstm_dir_barrier(bar + (size_t)dir * gridDim.x, (int)gridDim.x, (unsigned)(step + 1));
    }
}

template <int H, int BT>
__global__ __launch_bounds__(512) void lstm_seq_mfma_kernel(const float* __restrict__ gx, const float* __restrict__ whh, float* hbuf,
                                                            float* __restrict__ c, float* __restrict__ out,
                                                            const int* __restrict__ lens, unsigned* bar, int B, int T, int ndir) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    __shared__ float red[8][BT * 2 * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int j0 = blockIdx.x * 8;
    const int col = lane & 15, q = lane >> 4;
    const int k0 = wave * (H / 8);
    const size_t hsz = (size_t)ndir * B * H;
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hbuf, 0, (unsigned)(2 * hsz * sizeof(float)), 0x00020000);
    constexpr int NKB = H / 8 / 16;
    // this wave's K slice of the workgroup's 32 gate rows: 64 registers, loaded once
    f32x4 wv[NKB][2];
    const float* wbase = whh + ((size_t)dir * 4 * H + (size_t)(col >> 3) * H + j0 + (col & 7)) * H + k0 + 4 * q;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wv[kb][p] = *reinterpret_cast<const f32x4*>(wbase + (size_t)(2 * p) * H * H + kb * 16);
    for (int step = 0; step < T; ++step) {
        const size_t h_prev = (size_t)(step & 1) * hsz, h_next = (size_t)((step + 1) & 1) * hsz;      // float offsets in hbuf
        const int t = dir ? T - 1 - step : step;
        f32x4v acc[BT][2];
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[bt][p] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const size_t hbase = h_prev + (size_t)dir * B * H + k0 + 4 * q;
        f32x4 hv[NKB][BT];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                const int b = min(bt * 16 + col, B - 1);
                hv[kb][bt] = h_load4(hrs, hbase + (size_t)b * H + kb * 16);
            }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        acc[bt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[kb][bt][i], wv[kb][p][i], acc[bt][p], 0, 0, 0);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][(bt * 2 + p) * 4 + r][lane] = acc[bt][p][r];
        __syncthreads();
        for (int e = threadIdx.x; e < BT * 16 * 8; e += 512) {
            const int b = e >> 3, u = e & 7;
            if (b >= B) continue;
            const int bt = b >> 4, r = b & 3, lq = (b >> 2) & 3;
            float dot[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int l = lq * 16 + (g & 1) * 8 + u;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) a += red[w][(bt * 2 + (g >> 1)) * 4 + r][l];
                dot[g] = a;
            }
            const int j = j0 + u;
            const size_t sidx = ((size_t)dir * B + b) * H + j;
            const bool active = !lens || t < lens[b];
            const float* gr = gx + ((size_t)b * T + t) * (ndir * 4 * H) + (size_t)dir * 4 * H + j;
            const float gi = gr[0] + dot[0], gf = gr[H] + dot[1], gg = gr[2 * H] + dot[2], go = gr[3 * H] + dot[3];
            const float c_old = c[sidx];
            const float c_new = sigm(gf) * c_old + sigm(gi) * tanhf(gg);
            const float h_new = sigm(go) * tanhf(c_new);
            float* o = out + ((size_t)b * T + t) * (ndir * H) + dir * H + j;
            if (active) {
                c[sidx] = c_new;
                h_store1(hrs, h_next + sidx, h_new);
                *o = h_new;
            } else {
                h_store1(hrs, h_next + sidx, h_load1(hrs, h_prev + sidx));
                *o = 0.f;
            }
        }
        // (the barrier's own __syncthreads also keeps `red` of this step apart from the next step's partial tiles)
        if (step + 1 < T) lstm_dir_barrier(bar + (size_t)dir * gridDim.x, (int)gridDim.x, (unsigned)(step + 1));
    }
}

// all T steps of one layer in one cooperative launch; `bar`: LSTM_BAR_WORDS zeroed words of this launch's own (a flag per workgroup).  Returns 0 when launched,
// 1 when this configuration stays on the per-step launches (sizes, no cooperative launch on the device)
int launch_lstm_seq(const float* gx, const float* whh, float* hbuf, float* c, float* out, const int* lens, unsigned* bar, int B,
                    int T, int H, int ndir, hipStream_t s) {
    if (H != 1024 || T < 2) return 1;
    static int coop = -1;
    if (coop < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess) v = 0;
        coop = v;
    }
    if (!coop) return 1;
    void* args[] = {(void*)&gx, (void*)&whh, (void*)&hbuf, (void*)&c, (void*)&out, (void*)&lens, (void*)&bar, (void*)&B, (void*)&T, (void*)&ndir};
    hipError_t rc;
    if (B > 4 && B <= 32) {
        const dim3 grid(H / 8, ndir), blk(512);
        rc = B <= 16 ? hipLaunchCooperativeKernel((const void*)lstm_seq_mfma_kernel<1024, 1>, grid, blk, args, 0, s)
                     : hipLaunchCooperativeKernel((const void*)lstm_seq_mfma_kernel<1024, 2>, grid, blk, args, 0, s);
    } else {
        rc = hipLaunchCooperativeKernel((const void*)lstm_seq_kernel<1024>, dim3(H / 4, ndir), dim3(256), args, 0, s);
    }
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    return 0;
}

