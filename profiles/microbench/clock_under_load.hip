// Sustained shader clock of the whole chip under the vector-ALU loads the dense kernel is made of (gfx950): every wave reads the
// shader-clock counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around a long run of arithmetic; the ratio of the
// two differences is the clock the wave actually ran at.  The per-instruction issue cost follows in REAL cycles (valu_rate.hip prices
// at the nominal 2400 MHz).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/microbench/clock_under_load.hip -o profiles/microbench/clock_under_load
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kChains = 8;

// MIX 0: v_fma_f64 only; 1: v_fma_f32 only; 2: one f64 fma to two f32 fma (about the dense kernel's 166 : 350 split); 3: idle (sleep)
template <int MIX>
__global__ void __launch_bounds__(256, 8) load_kernel(unsigned long long* stamps, double* out, double seed, int iters) {
    double a[kChains];
    float f[kChains], g[kChains];
    for (int i = 0; i < kChains; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; g[i] = f[i] * 0.5f; }
    const double m = seed * 1.0000001, c = seed * 0.5;
    const float fm = (float)m, fc = (float)c;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (MIX == 0 || MIX == 2) a[i] = fma(a[i], m, c);
            if (MIX == 1 || MIX == 2) { f[i] = fmaf(f[i], fm, fc); g[i] = fmaf(g[i], fm, fc); }
            if (MIX == 3) __builtin_amdgcn_s_sleep(8);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    double s = 0.0;
    for (int i = 0; i < kChains; ++i) s += a[i] + (double)f[i] + (double)g[i];
    if (s == 1.2345e-300) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * w] = t1 - t0;
        stamps[4 * w + 1] = r1 - r0;
        stamps[4 * w + 2] = r0;
        stamps[4 * w + 3] = r1;
    }
}

template <int MIX>
void run(const char* name, int vec_insts_per_step, int iters) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 8;           // eight waves per SIMD on every CU
    unsigned long long* d_st; double* d_out;
    CHK(hipMalloc(&d_st, (size_t)blocks * 4 * 4 * sizeof(unsigned long long))); CHK(hipMalloc(&d_out, 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int warm = 0; warm < 3; ++warm) load_kernel<MIX><<<blocks, 256>>>(d_st, d_out, 1.000001, iters);      // the clock settles under the load
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    load_kernel<MIX><<<blocks, 256>>>(d_st, d_out, 1.000001, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 4 * 4);
    CHK(hipMemcpy(h.data(), d_st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    double cyc_sum = 0.0;
    const size_t n_waves = h.size() / 4;
    unsigned long long first = ~0ull, last_start = 0, last_end = 0;
    double busy_us = 0.0;
    for (size_t w = 0; w < n_waves; ++w) {
        mhz.push_back((double)h[4 * w] / (double)h[4 * w + 1] * 100.0); cyc_sum += (double)h[4 * w];
        first = std::min(first, h[4 * w + 2]); last_start = std::max(last_start, h[4 * w + 2]); last_end = std::max(last_end, h[4 * w + 3]);
        busy_us += (double)h[4 * w + 1] / 100.0;
    }
    std::sort(mhz.begin(), mhz.end());
    const double wave_insts = (double)iters * kChains * vec_insts_per_step;
    // eight waves share a SIMD: a wave's cycles / (its instructions x 8) = cycles per wave-instruction of the SIMD
    const double cyc_per_inst = vec_insts_per_step ? cyc_sum / (double)n_waves / (wave_insts * 8.0) : 0.0;
    printf("%-44s %.3f ms  shader clock min %.0f  p50 %.0f  max %.0f MHz (nominal %.0f)  %.2f real cycles per wave-instruction\n", name, ms, mhz.front(),
           mhz[mhz.size() / 2], mhz.back(), p.clockRate / 1e3, cyc_per_inst);
    printf("    waves %zu: a wave's loop lasts %.1f us on average; first start -> last start %.1f us, first start -> last end %.1f us (100 MHz counter)\n", n_waves,
           busy_us / (double)n_waves, (double)(last_start - first) / 100.0, (double)(last_end - first) / 100.0);
    CHK(hipFree(d_st)); CHK(hipFree(d_out));
}

int main() {
    run<3>("idle waves (s_sleep)", 0, 4096);
    run<0>("v_fma_f64, 8 waves per SIMD, all CUs", 1, 16384);
    run<1>("v_fma_f32 x 2", 2, 16384);
    run<2>("1 x v_fma_f64 + 2 x v_fma_f32 (dense kernel's mix)", 3, 16384);
    run<0>("v_fma_f64 again (long: 4 x)", 1, 65536);
    return 0;
}
