// N1: point-cloud record packing on the device (upstream core/writers.py:15-46 + core/image_utils.py:24-26).
//
// Upstream quantises colours with np.clip(np.round(rgb*255), 0, 255).astype(uint8) (f32 multiply,
// round-half-to-even) and then packs every point with struct.pack in a Python loop.  Here one kernel
// produces the final file payload - 15-byte PLY vertex records (3 x f32 LE + 3 x u8) or upstream's
// 43-byte points3D.bin records (u64 id, 3 x f64, 3 x u8, f64 error) - so only the packed bytes cross
// PCIe.  Records are assembled in LDS and leave with coalesced dword stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfd_device.hpp"

namespace {

__device__ __forceinline__ unsigned char quantise_u8(float c) {
    const float v = rintf(c * 255.0f);            // np.round: half to even, in f32
    if (!(v == v)) return 0;                      // NaN -> 0 (x86 float->int conversion upstream runs on)
    return (unsigned char)fminf(fmaxf(v, 0.0f), 255.0f);
}

template <int REC>
__device__ __forceinline__ void flush_records(const unsigned char* lds, unsigned char* out, long long first_point,
                                              int n_points, int tid, int nthreads) {
    // [first_point*REC, +n_points*REC) bytes; the block start is 4-byte aligned because the block
    // size is a multiple of 4 points
    const long long byte0 = first_point * REC;
    const int nbytes = n_points * REC;
    const int nwords = nbytes >> 2;
    unsigned* out32 = reinterpret_cast<unsigned*>(out + byte0);
    const unsigned* lds32 = reinterpret_cast<const unsigned*>(lds);
    for (int i = tid; i < nwords; i += nthreads) out32[i] = lds32[i];
    for (int i = (nwords << 2) + tid; i < nbytes; i += nthreads) out[byte0 + i] = lds[i];
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256) lfd_pack_ply_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                                      long long n, unsigned char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[256 * 15];
    const int tid = (int)threadIdx.x;
    for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
        const long long i = base + tid;
        if (i < n) {
            unsigned char* r = rec + tid * 15;
            const float v[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned u = __float_as_uint(v[c]);
                r[4 * c + 0] = (unsigned char)(u); r[4 * c + 1] = (unsigned char)(u >> 8);
                r[4 * c + 2] = (unsigned char)(u >> 16); r[4 * c + 3] = (unsigned char)(u >> 24);
            }
            r[12] = quantise_u8(rgb[3 * i]); r[13] = quantise_u8(rgb[3 * i + 1]); r[14] = quantise_u8(rgb[3 * i + 2]);
        }
        __syncthreads();
        const long long left = n - base;
        flush_records<15>(rec, out, base, left < 256 ? (int)left : 256, tid, 256);
        __syncthreads();
    }
}

extern "C" __global__ void __launch_bounds__(256) lfd_pack_points3d_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                                           const float* __restrict__ err, long long n,
                                                                           unsigned long long id_base, unsigned char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[256 * 43];
    const int tid = (int)threadIdx.x;
    auto put64 = [](unsigned char* p, unsigned long long u) {
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = (unsigned char)(u >> (8 * b));
    };
    for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
        const long long i = base + tid;
        if (i < n) {
            unsigned char* r = rec + tid * 43;
            put64(r, id_base + (unsigned long long)i + 1ull);
#pragma unroll
            for (int c = 0; c < 3; ++c) put64(r + 8 + 8 * c, (unsigned long long)__double_as_longlong((double)xyz[3 * i + c]));
            r[32] = quantise_u8(rgb[3 * i]); r[33] = quantise_u8(rgb[3 * i + 1]); r[34] = quantise_u8(rgb[3 * i + 2]);
            put64(r + 35, (unsigned long long)__double_as_longlong(err ? (double)err[i] : 0.0));
        }
        __syncthreads();
        const long long left = n - base;
        flush_records<43>(rec, out, base, left < 256 ? (int)left : 256, tid, 256);
        __syncthreads();
    }
}

extern "C" __global__ void lfd_quantise_rgb_kernel(const float* __restrict__ rgb, long long n3, unsigned char* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (long long)gridDim.x * blockDim.x)
        out[i] = quantise_u8(rgb[i]);
}
