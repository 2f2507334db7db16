// Diagnostic micro-study (not part of the library): what an in-kernel PAIR exchange costs on gfx950 -- the hand-off a "64-row FFN
// tile x d_ff halves on a CU pair" kernel (round-3 verdict, item 6) needs once (FFN + QKV tail) or twice (conv head + FFN) per launch:
// each workgroup of a pair publishes 32 rows x 256 floats (32 KB) of partial sums, raises a flag, waits for its partner's flag and
// reads the partner's 32 KB.  256 workgroups of 512 threads, one per CU, `rounds` exchanges back to back; partner = blockIdx ^ 1
// (neighbouring XCD: workgroups are dealt round-robin over the 8 XCDs) or blockIdx ^ 8 (same XCD, same L2).  Bounded spin: a
// workgroup that does not see its partner after 2^18 polls gives up and the run reports it.
//   hipcc --offload-arch=gfx950 -O3 tools/pair_exchange.hip -o tools/_pair_exchange.bin && tools/_pair_exchange.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: every thread fences at agent scope around the flag (the textbook form: each wave writes the L2 back and invalidates)
// MODE 1: the workgroup's stores are drained per wave (workgroup-scope release = s_waitcnt), ONE thread does the agent-scope release /
//         acquire on the flag, the data comes back through loads that bypass L1 (valid across XCDs)
// MODE 2: no cache maintenance at all -- valid only for a pair that shares an L2 (same XCD): L1 is write-through, atomics execute in
//         L2, and the partner's rows are read with L1-bypassing loads
// The payload is checked: word k of round r from workgroup w is r * 65536 + w * 8 + (k & 7); mismatches are counted.
template <int XOR, int MODE>
__global__ __launch_bounds__(512) void exchange(float* slots, int* flags, int rounds, float* out, int* gave_up, int* bad) {
    const int tid = threadIdx.x, me = blockIdx.x, other = me ^ XOR;
    float* mine = slots + (size_t)me * 2 * 8192;              // two buffers of 32 KB per workgroup (round parity)
    const float* theirs = slots + (size_t)other * 2 * 8192;
    bool lost = false;
    int wrong = 0;
    float sum = 0.f;
    for (int r = 1; r <= rounds; ++r) {
        float* dst = mine + (r & 1) * 8192;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = __int_as_float(r * 65536 + me * 8 + k);
            *reinterpret_cast<f32x4*>(dst + (k * 512 + tid) * 4) = f32x4{v, v, v, v};
        }
        if (MODE == 0) __threadfence();
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // my wave's stores have reached L2
        __syncthreads();
        if (tid == 0) {
            if (MODE == 2) __hip_atomic_store(flags + me, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(flags + me, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (!lost && (MODE == 2 ? __hip_atomic_load(flags + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : __hip_atomic_load(flags + other, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < r)
                if (++spins > (1 << 18)) { atomicAdd(gave_up, 1); lost = true; }      // (sticky: no further waiting in this launch)
        }
        __syncthreads();
        if (MODE == 0) __threadfence();
        const float* src = theirs + (r & 1) * 8192;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (k * 512 + tid) * 4));
            if (__float_as_int(v[0]) != r * 65536 + other * 8 + k || __float_as_int(v[3]) != r * 65536 + other * 8 + k) ++wrong;
            sum += v[1];
        }
    }
    if (wrong) atomicAdd(bad, wrong);
    out[(size_t)me * 512 + tid] = sum;
}

// the same stores and loads without the hand-off (each workgroup reads its own slot): what the data movement alone costs
__global__ __launch_bounds__(512) void no_exchange(float* slots, int rounds, float* out) {
    const int tid = threadIdx.x, me = blockIdx.x;
    float* mine = slots + (size_t)me * 2 * 8192;
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{(float)tid, 1.f, 2.f, 3.f};
    for (int r = 1; r <= rounds; ++r) {
        float* dst = mine + (r & 1) * 8192;
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dst + (k * 512 + tid) * 4) = acc[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(dst + (k * 512 + ((tid + 64) & 511)) * 4);
            acc[k] = acc[k] * 0.5f + v * 0.5f;
        }
        __syncthreads();
    }
    out[(size_t)me * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <typename F>
static float timed(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const int grid = 256, rounds = 200;
    float *slots, *out;
    int *flags, *gave_up;
    (void)hipMalloc(&slots, (size_t)256 * 2 * 8192 * 4);
    (void)hipMalloc(&out, (size_t)256 * 512 * 4);
    (void)hipMalloc(&flags, 256 * 4);
    (void)hipMalloc(&gave_up, 4);
    (void)hipMemset(gave_up, 0, 4);
    auto report = [&](const char* name, float ms, float base_ms) {
        int g = 0;
        (void)hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost);
        printf("%-64s %7.2f us per exchange  (launch of %d rounds %.1f us; hand-off alone %+.2f us)%s\n", name, ms * 1e3 / rounds, rounds,
               ms * 1e3, (ms - base_ms) * 1e3 / rounds, g ? "  [some workgroup gave up waiting]" : "");
    };
    const float base = timed([&] { hipLaunchKernelGGL(no_exchange, dim3(grid), dim3(512), 0, 0, slots, rounds, out); });
    report("no hand-off: 32 KB stored and read back per round", base, base);
    int* bad;
    (void)hipMalloc(&bad, 4);
    auto run = [&](const char* name, auto kern) {
        (void)hipMemset(gave_up, 0, 4);
        (void)hipMemset(bad, 0, 4);
        const float t = timed([&] {
            (void)hipMemsetAsync(flags, 0, 256 * 4, 0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, slots, flags, rounds, out, gave_up, bad);
        });
        int nb = 0;
        (void)hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
        report(name, t, base);
        printf("%-64s payload words that were not the partner's current round: %d\n", "", nb);
    };
    run("pair ^ 1 (other XCD), every thread fences at agent scope", exchange<1, 0>);
    run("pair ^ 8 (same XCD),  every thread fences at agent scope", exchange<8, 0>);
    run("pair ^ 1 (other XCD), one agent-scope release / acquire", exchange<1, 1>);
    run("pair ^ 8 (same XCD),  one agent-scope release / acquire", exchange<8, 1>);
    run("pair ^ 8 (same XCD),  no cache maintenance (shared L2)", exchange<8, 2>);
    run("pair ^ 1 (other XCD), no cache maintenance [expected to FAIL]", exchange<1, 2>);
    return 0;
}
