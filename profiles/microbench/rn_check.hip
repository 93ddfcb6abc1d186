// How often do the one-correction-step square root / reciprocal / quotient of csrc/lfd_geometry.hpp (lfd_sqrt_rn_f32, lfd_rcp_rn_f32 + Markstein)
// differ from the compiler's IEEE operations?  2^32 inputs each: ALL positive normal f32 bit patterns in the range of squared ray lengths and norms
// of a scene ([2^-20, 2^40)) for the root and the reciprocal, 2^32 pseudo-random (a, b) pairs for the quotient.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math profiles/microbench/rn_check.hip -I lichtfeld-densification-plugin_amd/csrc -o rn_check && ./rn_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "lfd_geometry.hpp"

__global__ void check(unsigned long long* bad, unsigned lo_bits, unsigned long long n_patterns) {
    unsigned long long b_sqrt = 0, b_rcp = 0, b_div = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_patterns; i += stride) {
        const float x = __uint_as_float(lo_bits + (unsigned)i);
        if (lfd_sqrt_rn_f32(x) != sqrtf(x)) ++b_sqrt;
        const float r = lfd_rcp_rn_f32(x);
        if (r != 1.0f / x) ++b_rcp;
        // a numerator from a hash of the index, |a| <= x as for the components of a ray over its norm
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float a = x * ((float)(h >> 8) * (1.0f / 16777216.0f)) * ((h & 1u) ? -1.0f : 1.0f);
        if (lfd_div_by_recip_f32(a, x, r) != a / x) ++b_div;
    }
    atomicAdd(bad + 0, b_sqrt); atomicAdd(bad + 1, b_rcp); atomicAdd(bad + 2, b_div);
}

int main() {
    unsigned long long* d; unsigned long long h[3] = {0, 0, 0};
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    const unsigned lo = 0x35800000u;                       // 2^-20
    const unsigned hi = 0x53800000u;                       // 2^40
    const unsigned long long n = (unsigned long long)(hi - lo);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, lo, n);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%llu f32 values in [2^-20, 2^40): sqrt differs from IEEE on %llu (%.2e), reciprocal on %llu (%.2e), Markstein quotient on %llu (%.2e)\n", n, h[0],
           (double)h[0] / n, h[1], (double)h[1] / n, h[2], (double)h[2] / n);
    return 0;
}
