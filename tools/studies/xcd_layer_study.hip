// Diagnostic micro-study (not part of the library; VERDICT r4 item 3, with its kill criterion): what ONE streaming chunk-step layer
// would cost as an XCD-RESIDENT PERSISTENT kernel -- the 16-stream step's 256 rows as eight row blocks of 32, one per XCD, the 32
// workgroups of an XCD splitting every stage's COLUMNS (hidden units / output channels / key ranges) and meeting at XCD-local
// arrival counters instead of kernel boundaries.  This is the SKELETON of that kernel: every phase does the real data movement
// (gather the 32 rows another CU of the XCD just wrote, stream this workgroup's weight slice, issue its share of MFMAs, publish its
// output slice, arrive at the XCD's counter), with the light same-XCD protocol of tools/pair_exchange.hip MODE 2 (plain stores, L1-
// bypassing loads, relaxed agent-scope counter: valid ONLY while the 32 workgroups share an L2 -- XCC_ID is checked) and, for
// comparison, with the placement-independent agent-scope release / acquire around every arrival.  Payloads are tagged and checked.
// No arithmetic of the model is reproduced: the number that comes out is a LOWER bound for the real kernel, to be held against the
// nine launches of today's layer (~95 us at 16 streams, profiles/r04_stream16_kernel_stats.txt) and the verdict's kill line
// (>= 25 % faster, i.e. <= 71 us).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_layer_study.hip -o tools/_xcd_layer.bin && tools/_xcd_layer.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Phase {
    int in_bytes;     // activation bytes this workgroup gathers from what the XCD's workgroups published in the previous phase
    int w_bytes;      // weight (or K/V cache) bytes this workgroup streams
    int mfma;         // v_mfma_f32_32x32x2_f32 per wave
    int out_bytes;    // bytes this workgroup publishes for the next phase
};
// one Conformer chunk-step layer for 32 rows (2 streams x 16 frames) on 32 CUs; K / V cache of ~500 keys per stream
__constant__ Phase kPhases[9] = {
    {32768, 131072, 64, 32768},   // LN + FFN1: 64 hidden units per CU (W1 slice 64 KB, W2 slice 64 KB), partial [32, 256] out
    {32768, 0, 0, 1024},          // reduce the 32 partials of my 8 columns + residual
    {32768, 24576, 16, 3072},     // LN + fused QKV: 24 columns per CU
    {4096, 65536, 8, 4352},       // attention: (stream, head) x a quarter of the keys per CU (64 KB of cache), partial O + (m, l)
    {34816, 8192, 4, 1024},       // merge the partials, out-projection (8 columns per CU) + residual
    {32768, 16384, 8, 1024},      // LN + pointwise_conv1 + GLU: 16 of 512 columns per CU
    {32768, 8192, 4, 1024},       // depthwise conv + LN + SiLU + pointwise_conv2 + residual
    {32768, 131072, 64, 32768},   // LN + FFN2
    {32768, 0, 0, 1024},          // reduce + residual + final LayerNorm
};

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }     // HW_REG_XCC_ID, 4 bits

// PROTO 0: same-XCD light protocol; PROTO 1: agent-scope release before / acquire after every arrival (valid for any placement)
template <int PROTO>
__global__ __launch_bounds__(512) void layer_skeleton(float* act /*[8][2][32 slots][8192 floats]*/, const float* weights, size_t w_per_layer,
                                                      unsigned* counters /*[8][32]: arrival counter of the XCD + rank dispenser*/,
                                                      int layers, int* bad, int* mixed, float* sink) {
    __shared__ int s_rank, s_xcc;
    __shared__ float tile[8192 + 1024];
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_xcc = xcc_id();
        s_rank = (int)atomicAdd(&counters[s_xcc * 32 + 1], 1u);
        if (s_xcc != (int)(blockIdx.x & 7)) atomicAdd(mixed, 1);          // the dispatcher did not deal round-robin: the light protocol would be invalid
    }
    __syncthreads();
    const int xcc = s_xcc, rank = s_rank & 31;
    unsigned* bar = counters + xcc * 32;
    float* mine = act + (size_t)xcc * 2 * 32 * 8192;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float keep = 0.f;
    int wrong = 0;
    unsigned target = 0;
    for (int l = 0; l < layers; ++l) {
        const float* wl = weights + (size_t)l * w_per_layer;
        size_t woff = (size_t)rank * 131072 / 4;          // every workgroup streams its own slice; all XCDs read the same bytes
#pragma unroll 1
        for (int p = 0; p < 9; ++p) {
            const Phase ph = kPhases[p];
            const int step = l * 9 + p;
            // 1. gather: in_bytes of the previous phase's slots (slot s was written by rank s), spread over the 32 producers
            if (step > 0) {
                const float* prev = mine + (size_t)((step - 1) & 1) * 32 * 8192;
                const int n16 = ph.in_bytes / 16;                       // 16-byte pieces
                const int per_slot = max(n16 / 32, 1);
                for (int i = tid; i < n16; i += 512) {
                    const int slot = (i / per_slot) & 31, k = i % per_slot;
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(prev + (size_t)slot * 8192 + k * 4));
                    if (__float_as_int(v[0]) != step - 1 || __float_as_int(v[3]) != slot) ++wrong;
                    tile[(i * 4) & 8191] = v[1];
                }
            }
            // 2. weights of the phase: coalesced 16-byte loads, consumed by the MFMAs below
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < ph.w_bytes / 16; i += 512) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(wl + woff + (size_t)i * 4);
                w += v;
            }
            woff += 32 * 131072 / 4;
            __syncthreads();
            // 3. this wave's share of the matrix work (operands from LDS + the weight registers)
            const float a = tile[tid & 1023];
            for (int m = 0; m < ph.mfma; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[m & 3], acc, 0, 0, 0);
            keep += acc[0];
            // 4. publish my slice: tagged {step, *, *, rank}
            float* out = mine + (size_t)(step & 1) * 32 * 8192 + (size_t)rank * 8192;
            for (int i = tid; i < ph.out_bytes / 16; i += 512)
                *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{__int_as_float(step), keep, w[1], __int_as_float(rank)};
            // 5. arrive at the XCD's counter and wait for the other 31 workgroups
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // my wave's stores have left the CU (L1 is write-through)
            __syncthreads();
            target += 32;
            if (tid == 0) {
                if (PROTO == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 20)) { atomicAdd(bad, 1 << 20); break; }
                }
                if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    sink[blockIdx.x * 512 + tid] = keep + acc[3];
}

// the same phases as nine separate launches per layer over all 256 workgroups (no in-kernel barrier): the launch-per-stage form
__global__ __launch_bounds__(512) void phase_launch(float* act, const float* weights, size_t woff0, int p, int step, float* sink) {
    __shared__ float tile[8192 + 1024];
    const int tid = threadIdx.x, xcc = blockIdx.x & 7, rank = blockIdx.x >> 3;
    float* mine = act + (size_t)xcc * 2 * 32 * 8192;
    const Phase ph = kPhases[p];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (step > 0) {
        const float* prev = mine + (size_t)((step - 1) & 1) * 32 * 8192;
        const int n16 = ph.in_bytes / 16, per_slot = max(n16 / 32, 1);
        for (int i = tid; i < n16; i += 512) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(prev + (size_t)((i / per_slot) & 31) * 8192 + (i % per_slot) * 4);
            tile[(i * 4) & 8191] = v[1];
        }
    }
    f32x4 w = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < ph.w_bytes / 16; i += 512) w += *reinterpret_cast<const f32x4*>(weights + woff0 + (size_t)rank * 131072 / 4 + (size_t)i * 4);
    __syncthreads();
    const float a = tile[tid & 1023];
    for (int m = 0; m < ph.mfma; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[m & 3], acc, 0, 0, 0);
    float* out = mine + (size_t)(step & 1) * 32 * 8192 + (size_t)rank * 8192;
    for (int i = tid; i < ph.out_bytes / 16; i += 512) *reinterpret_cast<f32x4*>(out + i * 4) = f32x4{__int_as_float(step), acc[0], w[1], __int_as_float(rank)};
    if (tid == 0) sink[blockIdx.x] = acc[3];
}

int main() {
    const int layers = 12;
    const size_t w_per_layer = (size_t)9 * 32 * 131072 / 4;          // floats: every phase has a 4 MB window, of which ph.w_bytes x 32 are read
    float *act, *weights, *sink;
    unsigned* counters;
    int *bad, *mixed;
    (void)hipMalloc(&act, (size_t)8 * 2 * 32 * 8192 * 4);
    (void)hipMalloc(&weights, w_per_layer * layers * 4);
    (void)hipMalloc(&sink, 256 * 512 * 4);
    (void)hipMalloc(&counters, 8 * 32 * 4);
    (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&mixed, 4);
    (void)hipMemset(weights, 0, w_per_layer * layers * 4);
    (void)hipMemset(act, 0, (size_t)8 * 2 * 32 * 8192 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto run = [&](const char* name, int proto) {
        float best = 1e9f;
        int nb = 0, nm = 0;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipMemsetAsync(counters, 0, 8 * 32 * 4, 0);
            (void)hipMemsetAsync(bad, 0, 4, 0);
            (void)hipMemsetAsync(mixed, 0, 4, 0);
            (void)hipEventRecord(e0);
            if (proto == 0) hipLaunchKernelGGL(layer_skeleton<0>, dim3(256), dim3(512), 0, 0, act, weights, w_per_layer, counters, layers, bad, mixed, sink);
            else hipLaunchKernelGGL(layer_skeleton<1>, dim3(256), dim3(512), 0, 0, act, weights, w_per_layer, counters, layers, bad, mixed, sink);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
            (void)hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&nm, mixed, 4, hipMemcpyDeviceToHost);
        }
        printf("%-78s %7.1f us per layer (%d layers x 9 phases in one launch: %.1f us; %.2f us per phase)  stale / timed-out words: %d, workgroups off their XCD: %d\n",
               name, best * 1e3 / layers, layers, best * 1e3, best * 1e3 / layers / 9, nb, nm);
    };
    run("XCD-resident persistent layer, same-XCD light protocol (no cache maintenance)", 0);
    run("XCD-resident persistent layer, agent-scope release / acquire at every arrival", 1);
    {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipEventRecord(e0);
            for (int l = 0; l < layers; ++l)
                for (int p = 0; p < 9; ++p)
                    hipLaunchKernelGGL(phase_launch, dim3(256), dim3(512), 0, 0, act, weights, (size_t)l * w_per_layer + (size_t)p * 32 * 131072 / 4, p,
                                       l * 9 + p, sink);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-78s %7.1f us per layer (%d launches: %.1f us; %.2f us per launch)\n", "the same nine phases as nine launches per layer (kernel boundaries)",
               best * 1e3 / layers, layers * 9, best * 1e3, best * 1e3 / layers / 9);
    }
    return 0;
}
