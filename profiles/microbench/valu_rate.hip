// Issue rate of the vector-ALU instructions the dense kernel is made of (gfx950): cycles per wave64 instruction on one
// SIMD, measured with 8 independent dependency chains per lane and 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/microbench/valu_rate.hip -o profiles/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIters = 2048;
constexpr int kChains = 8;

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(double* out, double seed, int iters) {
    double a[kChains];
    float f[kChains];
    unsigned u[kChains];
    for (int i = 0; i < kChains; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; u[i] = (unsigned)a[i]; }
    const double m = seed * 1.0000001, c = seed * 0.5;
    const float fm = (float)m, fc = (float)c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) a[i] = fma(a[i], m, c);                                   // v_fma_f64
            if (OP == 1) a[i] = a[i] * m;                                          // v_mul_f64
            if (OP == 2) a[i] = a[i] + c;                                          // v_add_f64
            if (OP == 3) f[i] = fmaf(f[i], fm, fc);                                // v_fma_f32
            if (OP == 4) { a[i] = (double)f[i]; f[i] = (float)a[i] + fc; }         // v_cvt_f64_f32 + v_cvt_f32_f64 + v_add_f32
            if (OP == 5) a[i] = __builtin_amdgcn_rcp(a[i]);                        // v_rcp_f64
            if (OP == 6) u[i] = u[i] * 12345u + 7u;                                // v_mul_lo_u32 (+add)
            if (OP == 7) u[i] = __umul24(u[i], 12345u) + 7u;                       // v_mad_u32_u24
            if (OP == 8) f[i] = __builtin_amdgcn_rcpf(f[i]);                       // v_rcp_f32
            if (OP == 9) f[i] = __builtin_amdgcn_sqrtf(f[i]);                      // v_sqrt_f32
            if (OP == 10) a[i] = (double)u[i] + a[i];                              // v_cvt_f64_u32 + v_add_f64
        }
    }
    double s = 0.0;
    for (int i = 0; i < kChains; ++i) s += a[i] + (double)f[i] + (double)u[i];
    if (s == 1.2345e-300) out[0] = s;
}

template <int OP>
double run(const char* name, int insts_per_op, int waves_per_simd) {
    int dev = 0; hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;     // 256 threads = 4 waves = one per SIMD
    double* d; CHK(hipMalloc(&d, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(d, 1.000001, 16);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(d, 1.000001, kIters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts_per_simd = (double)kIters * kChains * insts_per_op * waves_per_simd;
    const double clk = (double)p.clockRate * 1e3;   // Hz
    const double cyc = ms * 1e-3 * clk / wave_insts_per_simd;
    printf("%-34s waves/SIMD %d  %.3f ms  %.2f cycles per wave-instruction (at %.0f MHz)\n", name, waves_per_simd, ms, cyc, clk / 1e6);
    CHK(hipFree(d));
    return cyc;
}

// accuracy of v_rcp_f64 and of its Newton refinements: max relative error over n random arguments
__global__ void rcp_accuracy_kernel(const double* x, int n, double* err3) {
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double d = x[i];
        const double exact = 1.0 / d;
        double r = __builtin_amdgcn_rcp(d);
        e0 = fmax(e0, fabs(r - exact) / fabs(exact));
        r = fma(fma(-d, r, 1.0), r, r);
        e1 = fmax(e1, fabs(r - exact) / fabs(exact));
        r = fma(fma(-d, r, 1.0), r, r);
        e2 = fmax(e2, fabs(r - exact) / fabs(exact));
    }
    // one value per thread is enough for a max: atomics on the bit patterns (non-negative doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long*>(err3 + 0), (unsigned long long)__double_as_longlong(e0));
    atomicMax(reinterpret_cast<unsigned long long*>(err3 + 1), (unsigned long long)__double_as_longlong(e1));
    atomicMax(reinterpret_cast<unsigned long long*>(err3 + 2), (unsigned long long)__double_as_longlong(e2));
}

void rcp_accuracy() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);      // [1, 2)
        const int e = (int)((s >> 3) % 80) - 40;
        h[i] = ldexp(m, e) * ((s & 1) ? 1.0 : -1.0);
    }
    double *d, *derr; CHK(hipMalloc(&d, n * 8)); CHK(hipMalloc(&derr, 24)); CHK(hipMemset(derr, 0, 24));
    CHK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
    rcp_accuracy_kernel<<<256, 256>>>(d, n, derr);
    double r[3]; CHK(hipMemcpy(r, derr, 24, hipMemcpyDeviceToHost));
    printf("v_rcp_f64 max relative error over %d arguments: raw %.3e (2^%.1f), one Newton step %.3e (2^%.1f), two steps %.3e\n", n,
           r[0], log2(r[0]), r[1], log2(r[1] > 0 ? r[1] : 1e-300), r[2]);
    CHK(hipFree(d)); CHK(hipFree(derr));
}

int main() {
    rcp_accuracy();
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f64", 1, w);
        run<1>("v_mul_f64", 1, w);
        run<2>("v_add_f64", 1, w);
        run<3>("v_fma_f32", 1, w);
        run<4>("cvt f64<-f32, f32<-f64, add_f32", 3, w);
        run<5>("v_rcp_f64", 1, w);
        run<6>("v_mul_lo_u32 + v_add", 2, w);
        run<7>("v_mad_u32_u24", 1, w);
        run<8>("v_rcp_f32", 1, w);
        run<9>("v_sqrt_f32", 1, w);
        run<10>("v_cvt_f64_u32 + v_add_f64", 2, w);
    }
    return 0;
}
