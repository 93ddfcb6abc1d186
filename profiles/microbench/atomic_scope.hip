// Round-trip latency of a returning atomic add by memory scope (MI355X, gfx950): the dense kernel's ticket is an agent-scope atomic - what would a
// scope that stays inside the XCD's L2 cost?   hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int kScope>
__global__ void chain(unsigned long long* ctr, int n, long long* ticks) {
    if (threadIdx.x != 0) return;          // ONE lane: 64 lanes on one address would be 64 serialised atomics per instruction
    long long t0 = wall_clock64();
    unsigned long long v = 0;
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(ctr, 1ull + (v & 1ull), __ATOMIC_RELAXED, kScope);
    long long t1 = wall_clock64();
    *ticks = t1 - t0; if (v == 12345) ctr[1] = v;
}
// one lane of many workgroups, each ONE returning atomic on one of `lanes` counters 128 B apart: the ticket pattern (throughput under contention)
template <int kScope>
__global__ void tickets(unsigned long long* ctr, int lanes, unsigned* sink) {
    if (threadIdx.x == 0) {
        unsigned long long v = __hip_atomic_fetch_add(ctr + (size_t)(blockIdx.x % lanes) * 16, 1ull, __ATOMIC_RELAXED, kScope);
        if (v == 0xffffffffffffull) *sink = 1;
    }
}
// the same, the round trip timed per workgroup (mean over the grid)
template <int kScope>
__global__ void ticket_latency(unsigned long long* ctr, int lanes, unsigned long long* sum_ticks) {
    if (threadIdx.x == 0) {
        long long t0 = wall_clock64();
        unsigned long long v = __hip_atomic_fetch_add(ctr + (size_t)(blockIdx.x % lanes) * 16, 1ull, __ATOMIC_RELAXED, kScope);
        long long t1 = wall_clock64() + (long long)(v & 0);
        atomicAdd(sum_ticks, (unsigned long long)(t1 - t0));
    }
}

int main() {
    int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    const double ns = 1e6 / (double)wc_khz;
    unsigned long long* ctr; long long* ticks; unsigned* sink; unsigned long long* sum;
    CHECK(hipMalloc(&ctr, 4096)); CHECK(hipMalloc(&ticks, 64)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&sum, 64));
    CHECK(hipMemset(ctr, 0, 4096));
    const int n = 2000;
    long long t;
#define RUN_CHAIN(scope, name) do { chain<scope><<<1, 64>>>(ctr, n, ticks); chain<scope><<<1, 64>>>(ctr, n, ticks); CHECK(hipDeviceSynchronize()); \
        CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); printf("serial returning atomic add, %-10s scope: %7.1f ns per round trip\n", name, t * ns / n); } while (0)
    RUN_CHAIN(__HIP_MEMORY_SCOPE_SYSTEM, "system");
    RUN_CHAIN(__HIP_MEMORY_SCOPE_AGENT, "agent");
    RUN_CHAIN(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup");
    RUN_CHAIN(__HIP_MEMORY_SCOPE_WAVEFRONT, "wavefront");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lanes : {1, 8, 32}) {
#define RUN_T(scope, name) do { tickets<scope><<<16384, 256>>>(ctr, lanes, sink); hipEventRecord(e0); for (int r = 0; r < 10; ++r) tickets<scope><<<16384, 256>>>(ctr, lanes, sink); hipEventRecord(e1); \
        CHECK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1); \
        CHECK(hipMemset(sum, 0, 8)); ticket_latency<scope><<<16384, 256>>>(ctr, lanes, sum); CHECK(hipDeviceSynchronize()); unsigned long long s; CHECK(hipMemcpy(&s, sum, 8, hipMemcpyDeviceToHost)); \
        printf("16384 workgroups, one ticket each, %2d counters, %-9s scope: %7.1f us per launch, mean round trip %7.1f ns\n", lanes, name, ms * 100.0, (double)s * ns / 16384.0); } while (0)
        RUN_T(__HIP_MEMORY_SCOPE_AGENT, "agent");
        RUN_T(__HIP_MEMORY_SCOPE_WORKGROUP, "workgroup");
    }
    return 0;
}
