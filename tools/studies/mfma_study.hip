// Diagnostic micro-study (not part of the library): what bounds a v_mfma_f32_32x32x2_f32 stream on gfx950 in the shape the fused
// FFN kernels run it -- 8 waves per workgroup (2 per SIMD), one workgroup per CU, ~100 us per launch.
// Variants: NACC accumulator chains per wave x side work per four MFMAs (none | one ds_read_b128 | + one coalesced 16-byte global
// load from an L2-resident 4 MB stream, as the packed FFN weights).  Prints ns per MFMA per SIMD and the shader clock the kernel
// saw (s_memtime ticks per s_memrealtime tick x 100 MHz), so that "cycles at the nominal 2.4 GHz" and real cycles can be told apart.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_study.hip -o /tmp/mfma_study && /tmp/mfma_study
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSR, int GLD, int WAVES, int ROT = 0, int DELAY = 0>
__global__ __launch_bounds__(64 * WAVES) void study(float* out, const float* __restrict__ wts, long long* clk, int iters, unsigned mask) {
    __shared__ __align__(16) float tile[32 * 260];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 260; i += 64 * WAVES) tile[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float* xa = tile + (lane & 31) * 260 + 4 * (lane >> 5);
    // this wave's weight stream: 16 bytes per lane per load, consecutive 1 KB blocks, wrapping inside 4 MB (shared by all workgroups)
    // ROT = 0: every workgroup walks the SAME addresses at the same time (as the FFN workgroups do with their weights);
    // ROT = 1: every workgroup starts at its own position of the 4 MB stream (32 KB apart)
    const size_t start = ROT > 2 ? (size_t)blockIdx.x * ROT : ROT ? (size_t)blockIdx.x * 8192 : (size_t)(blockIdx.x & 7) * 16384;   // ROT > 2: start stride in floats
    const float* wp = wts + lane * 4;
    const int adv = ROT == 2 ? 0 : 256;       // ROT = 2: every load of a wave re-reads the same 1 KB (L1 hits): issue cost without the memory system
    const size_t wofs = start + (size_t)wave * 4096;
    f32x4 ring[NACC * 8];                  // chain n: ring[8 n + g], refilled 8 groups (= 32 MFMAs of that chain) ahead
#pragma unroll
    for (int k = 0; k < NACC * 8; ++k) ring[k] = GLD ? *reinterpret_cast<const f32x4*>(wp + ((wofs + k * 256) & mask)) : f32x4{1.f, 2.f, 3.f, 4.f};
    f32x4 a = {1.f, 1.f, 1.f, 1.f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {          // 8 groups of 4 MFMAs per chain
            if (LDSR) a = *reinterpret_cast<const f32x4*>(xa + ((it * 8 + g) & 31) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) {
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], ring[8 * n + g][q], acc[n], 0, 0, 0);
                    // DELAY = 0: the refill of a fragment is issued right behind the last MFMA that reads it (its destination registers
                    // are a source operand of the MFMA still in flight); DELAY = 1: two MFMAs of the next group later
                    if (GLD && DELAY && q == 1)
                        ring[8 * n + ((g + 7) & 7)] = *reinterpret_cast<const f32x4*>(wp + ((wofs + (size_t)(((it * 8 + g + 7) * NACC + n) * adv)) & mask));
                    if (GLD && !DELAY && q == 3)
                        ring[8 * n + g] = *reinterpret_cast<const f32x4*>(wp + ((wofs + (size_t)(((it * 8 + g + 8) * NACC + n) * adv)) & mask));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 17) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// 1 chain, 2 waves / SIMD, the weight stream through raw buffer loads with cache policy AUX (bit 0 sc0, bit 1 nt, bit 4 sc1) and a
// ring of RING fragments (RING x 4 MFMAs ahead)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int AUX, int RING, int GLOBAL = 0, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES) void study_buf(float* out, const float* __restrict__ wts, int iters, unsigned mask) {
    __shared__ __align__(16) float tile[32 * 260];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32 * 260; i += 64 * WAVES) tile[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xa = tile + (lane & 31) * 260 + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 8 << 20, 0x00020000);
    const unsigned wofs = (unsigned)(blockIdx.x & 7) * 16384 + (unsigned)wave * 4096 + lane * 4;
    f32x4 ring[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k)
        ring[k] = GLOBAL ? *reinterpret_cast<const f32x4*>(wts + ((wofs + k * 256) & mask))
                         : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((wofs + k * 256) & mask) * 4, 0, AUX));
    f32x4 a = *reinterpret_cast<const f32x4*>(xa);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < RING; ++g) {
            const f32x4 an = *reinterpret_cast<const f32x4*>(xa + ((it * RING + g + 1) & 31) * 8);      // next group's A fragment
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], ring[g][q], acc, 0, 0, 0);
                if (q == 3)
                    ring[g] = GLOBAL ? *reinterpret_cast<const f32x4*>(wts + ((wofs + (unsigned)((it * RING + g + RING) * 256)) & mask))
                                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((wofs + (unsigned)((it * RING + g + RING) * 256)) & mask) * 4, 0, AUX));
                __builtin_amdgcn_sched_barrier(0);
            }
            a = an;
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}
template <int AUX, int RING, int GLOBAL = 0, int WAVES = 8>
void run_buf(const char* name, float* out, float* wts) {
    const int iters = 4096 / (4 * RING);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) study_buf<AUX, RING, GLOBAL, WAVES><<<256, 64 * WAVES>>>(out, wts, iters, 256 * 1024 - 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 40; ++k) study_buf<AUX, RING, GLOBAL, WAVES><<<256, 64 * WAVES>>>(out, wts, iters, 256 * 1024 - 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 40;
    printf("%-52s %6.1f us/launch  (%.1f TF)\n", name, ms * 1e3, 4096.0 * WAVES * 256 * 4096.0 / (ms * 1e-3) / 1e12);
}

// round 4: what 64-ROW tiles would buy the FFN main loop -- one weight fragment (1 KB per wave) feeds EIGHT MFMAs (two row tiles, two
// independent accumulator chains, two A fragments from LDS per group) instead of four
template <int RING, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES) void study_buf_pair(float* out, const float* __restrict__ wts, int iters, unsigned mask) {
    __shared__ __align__(16) float tile[64 * 260];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 260; i += 64 * WAVES) tile[i] = (float)(i & 7) * 0.01f;
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float* xa = tile + (lane & 31) * 260 + 4 * (lane >> 5);
    const float* xb = xa + 32 * 260;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 8 << 20, 0x00020000);
    const unsigned wofs = (unsigned)(blockIdx.x & 7) * 16384 + (unsigned)wave * 4096 + lane * 4;
    f32x4 ring[RING];
#pragma unroll
    for (int k = 0; k < RING; ++k)
        ring[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((wofs + k * 256) & mask) * 4, 0, 0));
    f32x4 a = *reinterpret_cast<const f32x4*>(xa), b = *reinterpret_cast<const f32x4*>(xb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < RING; ++g) {
            const f32x4 an = *reinterpret_cast<const f32x4*>(xa + ((it * RING + g + 1) & 31) * 8);
            const f32x4 bn = *reinterpret_cast<const f32x4*>(xb + ((it * RING + g + 1) & 31) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], ring[g][q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q], ring[g][q], acc1, 0, 0, 0);
                if (q == 3)
                    ring[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((wofs + (unsigned)((it * RING + g + RING) * 256)) & mask) * 4, 0, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
            a = an;
            b = bn;
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}
template <int RING, int WAVES = 8>
void run_buf_pair(const char* name, float* out, float* wts, int grid = 256) {
    const int iters = 4096 / (8 * RING);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) study_buf_pair<RING, WAVES><<<grid, 64 * WAVES>>>(out, wts, iters, 256 * 1024 - 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 40; ++k) study_buf_pair<RING, WAVES><<<grid, 64 * WAVES>>>(out, wts, iters, 256 * 1024 - 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 40;
    printf("%-60s %6.1f us/launch  (%.1f TF)\n", name, ms * 1e3, 4096.0 * WAVES * grid * 4096.0 / (ms * 1e-3) / 1e12);
}

// the same load stream without any MFMA: what a CU can pull through its L1 when all CUs do (8 loads of 1 KB in flight per wave)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stream_only(float* out, const float* __restrict__ wts, int nload, unsigned mask) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* wp = wts + lane * 4;
    const size_t wofs = (size_t)(blockIdx.x & 7) * 16384 + (size_t)wave * 4096;
    f32x4 ring[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ring[k] = *reinterpret_cast<const f32x4*>(wp + ((wofs + k * 256) & mask));
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nload / 8; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            s += ring[g];
            ring[g] = *reinterpret_cast<const f32x4*>(wp + ((wofs + (size_t)((it * 8 + g + 8) * 256)) & mask));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = s[0] + s[1] + s[2] + s[3];
}
template <int WAVES>
void run_stream(const char* name, float* out, float* wts, unsigned mask) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int nload = 4096;
    for (int k = 0; k < 3; ++k) stream_only<WAVES><<<256, 64 * WAVES>>>(out, wts, nload, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) stream_only<WAVES><<<256, 64 * WAVES>>>(out, wts, nload, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    const double bytes_cu = (double)nload * 1024.0 * WAVES;
    printf("%-52s %6.1f us/launch  %.1f GB/s per CU = %.1f B/clk @2.4 GHz, %.2f TB/s chip\n", name, ms * 1e3, bytes_cu / (ms * 1e-3) / 1e9,
           bytes_cu / (ms * 1e-3) / 2.4e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}

template <int NACC, int LDSR, int GLD, int WAVES, int ROT = 0, int DELAY = 0>
void run(const char* name, int blocks, float* out, float* wts, long long* clk, unsigned mask = 1024 * 1024 - 1) {
    const int iters = 4096 / (32 * NACC);          // 4096 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) study<NACC, LDSR, GLD, WAVES, ROT, DELAY><<<blocks, 64 * WAVES>>>(out, wts, clk, iters, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 40;
    for (int k = 0; k < reps; ++k) study<NACC, LDSR, GLD, WAVES, ROT, DELAY><<<blocks, 64 * WAVES>>>(out, wts, clk, iters, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double per_simd = 4096.0 * WAVES / 4.0;
    const double in_ns = (double)h[1] * 10.0;       // 100 MHz realtime ticks
    printf("%-52s %6.1f us/launch  in-kernel %6.1f us  %5.2f ns/MFMA/SIMD  shader clock %.3f GHz -> %.1f real cycles/MFMA  (%.1f TF)\n",
           name, ms * 1e3, in_ns * 1e-3, in_ns / per_simd, (double)h[0] / in_ns, (double)h[0] / per_simd,
           4096.0 * WAVES * blocks * 4096.0 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    float *out, *wts;
    long long* clk;
    hipMalloc(&out, sizeof(float) * 512 * 256);
    hipMalloc(&wts, 8 << 20);
    hipMemset(wts, 0, 8 << 20);
    hipMalloc(&clk, 16);
    if (argc > 1 && !strcmp(argv[1], "pair")) {      // round 4: 32-row against 64-row tiles (weights per MFMA halved)
        for (int rep = 0; rep < 2; ++rep) {
            run_buf<0, 8>("32-row tile: 1 KB of weights / 4 MFMAs, ring 8", out, wts);
            run_buf_pair<8>("64-row tile: 1 KB of weights / 8 MFMAs (2 chains), ring 8", out, wts);
            run_buf_pair<4>("64-row tile: 1 KB of weights / 8 MFMAs (2 chains), ring 4", out, wts);
            run_buf_pair<8>("64-row tile, 248 workgroups", out, wts, 248);
            run<2, 1, 0, 8>("2 chains, ds_read_b128 / 8 MFMAs, no weight stream", 256, out, wts, clk);
        }
        return 0;
    }
    run<1, 0, 0, 8>("1 chain, no side work, 2 waves/SIMD", 256, out, wts, clk);
    run<2, 0, 0, 8>("2 chains, no side work, 2 waves/SIMD", 256, out, wts, clk);
    run<1, 0, 0, 4>("1 chain, no side work, 1 wave/SIMD", 256, out, wts, clk);
    run<2, 0, 0, 4>("2 chains, no side work, 1 wave/SIMD", 256, out, wts, clk);
    run<1, 1, 0, 8>("1 chain, ds_read_b128 / 4 MFMAs, 2 waves/SIMD", 256, out, wts, clk);
    run<2, 1, 0, 8>("2 chains, ds_read_b128 / 8 MFMAs, 2 waves/SIMD", 256, out, wts, clk);
    run<1, 1, 1, 8>("1 chain, ds_read + global 16 B / 4 MFMAs, 2 w/SIMD", 256, out, wts, clk);
    run<2, 1, 1, 8>("2 chains, ds_read + global 16 B / 4 MFMAs, 2 w/SIMD", 256, out, wts, clk);
    run<1, 1, 1, 8>("same, 1 chain, 248 workgroups", 248, out, wts, clk);
    run<1, 1, 1, 8, 1>("1 chain, ds_read + global, 2 w/SIMD, own start per WG", 256, out, wts, clk);
    run<2, 1, 1, 8, 1>("2 chains, ds_read + global, 2 w/SIMD, own start per WG", 256, out, wts, clk);
    run<1, 1, 1, 8, 1>("1 chain, same, 248 workgroups", 248, out, wts, clk);
    run<1, 1, 1, 8, 0, 1>("1 chain, ds_read + global, 2 w/SIMD, refill 2 MFMAs later", 256, out, wts, clk);
    run<2, 1, 1, 8, 0, 1>("2 chains, ds_read + global, 2 w/SIMD, refill 2 MFMAs later", 256, out, wts, clk);
    run<1, 1, 1, 4, 0, 1>("1 chain, ds_read + global, 1 w/SIMD, refill 2 MFMAs later", 256, out, wts, clk);
    run<1, 1, 1, 8, 8448, 0>("1 chain, ds_read + global, 2 w/SIMD, WG starts 33 KB apart", 256, out, wts, clk);
    run<1, 1, 1, 8, 320, 0>("1 chain, ds_read + global, 2 w/SIMD, WG starts 1.25 KB apart", 256, out, wts, clk);
    run<1, 1, 1, 8, 4160, 0>("1 chain, ds_read + global, 2 w/SIMD, WG starts 16.25 KB apart", 256, out, wts, clk);
    run<1, 1, 1, 8, 33024, 0>("1 chain, ds_read + global, 2 w/SIMD, WG starts 129 KB apart", 256, out, wts, clk);
    run<1, 1, 1, 8, 0, 0>("1 chain, ds_read + global, 2 w/SIMD, 2 MB stream", 256, out, wts, clk, 512 * 1024 - 1);
    run<1, 1, 1, 8, 0, 0>("1 chain, ds_read + global, 2 w/SIMD, 1 MB stream", 256, out, wts, clk, 256 * 1024 - 1);
    run<1, 1, 1, 8, 0, 0>("1 chain, ds_read + global, 2 w/SIMD, 256 KB stream", 256, out, wts, clk, 64 * 1024 - 1);
    run<2, 1, 1, 8, 0, 0>("2 chains, ds_read + global, 2 w/SIMD, 1 MB stream", 256, out, wts, clk, 256 * 1024 - 1);
    run<1, 1, 1, 4, 0, 0>("1 chain, ds_read + global, 1 w/SIMD, 1 MB stream", 256, out, wts, clk, 256 * 1024 - 1);
    run_buf<0, 8>("buffer loads, plain, ring 8 (A fragment one group ahead)", out, wts);
    run_buf<0, 16>("buffer loads, plain, ring 16", out, wts);
    run_buf<0, 4>("buffer loads, plain, ring 4", out, wts);
    run_buf<1, 8>("buffer loads, sc0, ring 8", out, wts);
    run_buf<2, 8>("buffer loads, nt, ring 8", out, wts);
    run_buf<3, 8>("buffer loads, sc0 nt, ring 8", out, wts);
    run_buf<16, 8>("buffer loads, sc1, ring 8", out, wts);
    run_buf<17, 8>("buffer loads, sc0 sc1, ring 8", out, wts);
    run_buf<2, 16>("buffer loads, nt, ring 16", out, wts);
    run_buf<0, 8, 1>("GLOBAL loads (64-bit VGPR address), A one group ahead, ring 8", out, wts);
    run_buf<0, 8, 0, 4>("buffer loads, ring 8, 1 wave/SIMD", out, wts);
    run_buf<0, 8, 1, 4>("global loads, ring 8, 1 wave/SIMD", out, wts);
    run_stream<8>("loads only, 8 waves, 1 MB stream", out, wts, 256 * 1024 - 1);
    run_stream<8>("loads only, 8 waves, 4 MB stream", out, wts, 1024 * 1024 - 1);
    run_stream<4>("loads only, 4 waves, 1 MB stream", out, wts, 256 * 1024 - 1);
    run_stream<16>("loads only, 16 waves, 1 MB stream", out, wts, 256 * 1024 - 1);
    run<1, 1, 1, 8, 2, 0>("1 chain, ds_read + global (same 1 KB = L1 hits), 2 w/SIMD", 256, out, wts, clk);
    run<1, 1, 1, 4, 2, 0>("1 chain, ds_read + global (same 1 KB = L1 hits), 1 w/SIMD", 256, out, wts, clk);
    run<1, 1, 1, 4>("1 chain, ds_read + global / 4 MFMAs, 1 wave/SIMD", 256, out, wts, clk);
    run<2, 1, 1, 4>("2 chains, ds_read / 8 + global / 4 MFMAs, 1 wave/SIMD", 256, out, wts, clk);
    return 0;
}
