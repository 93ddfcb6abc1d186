// Latency micro-benchmarks that ground the design of the fused kernel (MI355X, gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// dependent pointer chase: one lane, N hops
__global__ void chase(const unsigned* next, int hops, unsigned* out, long long* ticks) {
    unsigned p = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < hops; ++i) p = next[p];
    long long t1 = wall_clock64();
    *out = p; *ticks = t1 - t0;
}
// same with agent-scope relaxed atomic loads (what the look-back polls with)
__global__ void chase_atomic(const unsigned* next, int hops, unsigned* out, long long* ticks) {
    unsigned p = 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < hops; ++i) p = __hip_atomic_load(next + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t1 = wall_clock64();
    *out = p; *ticks = t1 - t0;
}
// returning atomic add on one address, one lane per workgroup, many workgroups: ticket rate
__global__ void tickets(unsigned long long* ctr, unsigned* sink) {
    if (threadIdx.x == 0) { unsigned long long v = atomicAdd(ctr, 1ull); if (v == 0xffffffffffffull) *sink = 1; }
}
// serial returning atomics from one lane
__global__ void atomic_chain(unsigned long long* ctr, int n, long long* ticks) {
    long long t0 = wall_clock64();
    unsigned long long v = 0;
    for (int i = 0; i < n; ++i) v += atomicAdd(ctr, 1ull + (v & 1));
    long long t1 = wall_clock64();
    *ticks = t1 - t0; if (v == 12345) ctr[1] = v;
}
// ping-pong between two workgroups through agent-scope flags (store -> visible -> store back)
__global__ void pingpong(unsigned long long* flags, int n, long long* ticks) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (me == 0) {
            __hip_atomic_store(flags, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(flags + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
        } else {
            while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
            __hip_atomic_store(flags + 16, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    long long t1 = wall_clock64();
    if (me == 0) *ticks = t1 - t0;
}
// streaming read bandwidth with W waves per CU and B bytes in flight per lane (float4 x U)
template <int U>
__global__ void stream_read(const float4* __restrict__ src, size_t n4, float* sink) {
    float acc = 0.f;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void empty_kernel() {}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, wall clock %d kHz, core clock %d kHz\n", prop.name, prop.multiProcessorCount, wc_khz, prop.clockRate);
    const double ns_per_tick = 1e6 / (double)wc_khz;
    unsigned* d_out; long long* d_ticks; CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_ticks, 64));
    long long ticks;
    for (size_t bytes : {size_t(64) << 10, size_t(2) << 20, size_t(64) << 20, size_t(1) << 30}) {
        size_t n = bytes / 4;
        // random cycle with stride >= 64 elements (one hop = one new cache line)
        size_t lines = n / 64;
        std::vector<unsigned> order(lines); std::iota(order.begin(), order.end(), 0u);
        std::mt19937 g(1); std::shuffle(order.begin() + 1, order.end(), g);
        std::vector<unsigned> next(n, 0);
        for (size_t i = 0; i < lines; ++i) next[(size_t)order[i] * 64] = order[(i + 1) % lines] * 64;
        unsigned* d_next; CHECK(hipMalloc(&d_next, bytes)); CHECK(hipMemcpy(d_next, next.data(), bytes, hipMemcpyHostToDevice));
        int hops = 20000;
        hipEvent_t c0, c1; hipEventCreate(&c0); hipEventCreate(&c1); float cms = 0;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(c0); hipLaunchKernelGGL(chase, 1, 1, 0, 0, d_next, hops, d_out, d_ticks); hipEventRecord(c1); CHECK(hipDeviceSynchronize()); }
        hipEventElapsedTime(&cms, c0, c1);
        CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
        double a = ticks * ns_per_tick / hops;
        printf("  (hip events: %.1f ns/hop) ", cms * 1e6 / hops);
        hipLaunchKernelGGL(chase_atomic, 1, 1, 0, 0, d_next, hops, d_out, d_ticks); CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
        printf("pointer chase over %8.2f MiB: plain load %7.1f ns/hop, agent-scope atomic load %7.1f ns/hop\n", bytes / 1048576.0, a, ticks * ns_per_tick / hops);
        CHECK(hipFree(d_next));
    }
    unsigned long long* d_ctr; CHECK(hipMalloc(&d_ctr, 4096)); CHECK(hipMemset(d_ctr, 0, 4096));
    hipLaunchKernelGGL(atomic_chain, 1, 1, 0, 0, d_ctr, 5000, d_ticks); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
    printf("serial returning atomicAdd (one lane): %.1f ns each\n", ticks * ns_per_tick / 5000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int nb : {1024, 16384, 262144}) {
        hipLaunchKernelGGL(tickets, nb, 256, 0, 0, d_ctr, d_out);
        hipEventRecord(e0); hipLaunchKernelGGL(tickets, nb, 256, 0, 0, d_ctr, d_out); hipEventRecord(e1); CHECK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1);
        float ms2; hipEventRecord(e0); hipLaunchKernelGGL(empty_kernel, nb, 256, 0, 0); hipEventRecord(e1); CHECK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms2, e0, e1);
        printf("%7d workgroups x 1 ticket atomic: %.1f us (%.1f ns/wg); empty kernel same grid: %.1f us (%.1f ns/wg)\n", nb, ms * 1e3, ms * 1e6 / nb, ms2 * 1e3, ms2 * 1e6 / nb);
    }
    CHECK(hipMemset(d_ctr, 0, 4096));
    hipLaunchKernelGGL(pingpong, 2, 64, 0, 0, d_ctr, 2000, d_ticks); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
    printf("flag ping-pong between two workgroups: %.1f ns per round trip (2 store->load hops)\n", ticks * ns_per_tick / 2000);
    // bandwidth vs occupancy
    size_t bytes = size_t(2) << 30; float4* d_src; float* d_sink; CHECK(hipMalloc(&d_src, bytes)); CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemset(d_src, 1, bytes));
    for (int wg_per_cu : {1, 2, 4, 8}) {
        auto run = [&](auto kern, int U) {
            int grid = prop.multiProcessorCount * wg_per_cu;
            hipLaunchKernelGGL(kern, grid, 256, 0, 0, d_src, bytes / 16, d_sink);
            hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, 256, 0, 0, d_src, bytes / 16, d_sink); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
            printf("stream read, %d wg/CU (%2d waves/CU), %d x 16 B in flight per lane: %7.1f GB/s\n", wg_per_cu, wg_per_cu * 4, U, bytes / (ms * 1e-3) / 1e9);
        };
        run(stream_read<1>, 1); run(stream_read<4>, 4); run(stream_read<8>, 8);
    }
    return 0;
}
