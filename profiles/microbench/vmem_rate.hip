// Per-CU throughput of the vector-memory access patterns the dense kernel is made of (gfx950): bytes per cycle per CU and
// ns per wave-instruction, all CUs busy (7 x 256-thread workgroups per CU like the kernel), streaming over buffers far
// larger than the caches unless the pattern says "L2" (a 1.5 MB image re-read by everybody, like the colour taps).
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/vmem_rate.hip -o profiles/microbench/vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };   // 12-byte record, dwordx3 store (generic pointer, like the kernel)
#define GAS __attribute__((address_space(1)))

enum { LD_X4, LD_X2, LD_X1, LD_X2_STRIDE32, LD_X2_UNALIGNED_L2, LD_X4_PAIR32, ST_X4, ST_X3, ST_X2, ST_X1, ST_REC28, ST_X4_UNALIGNED12 };

// every workgroup streams `per_wg` bytes starting at its own offset; iteration i of a thread touches chunk i of the workgroup
template <int P>
__global__ void __launch_bounds__(256) pattern(char* buf, size_t per_wg, int iters, float* sink) {
    const int tid = threadIdx.x;
    char* base = buf + (size_t)blockIdx.x * per_wg;
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if (P == LD_X4) { const f4 v = *(const f4 GAS*)(base + ((size_t)it * 256 + tid) * 16); acc += v.x + v.w; }
        if (P == LD_X2) { const f2 v = *(const f2 GAS*)(base + ((size_t)it * 256 + tid) * 8); acc += v.x + v.y; }
        if (P == LD_X1) { acc += *(const float GAS*)(base + ((size_t)it * 256 + tid) * 4); }
        if (P == LD_X2_STRIDE32) {     // the dense kernel's winner-warp loads: 8 B at a 32-B stride, four instructions cover the span
            const size_t chunk = (size_t)(it >> 2) * 256 * 32;
            const f2 v = *(const f2 GAS*)(base + chunk + (size_t)tid * 32 + (it & 3) * 8);
            acc += v.x + v.y;
        }
        if (P == LD_X4_PAIR32) {       // the same bytes as two 16-B loads per thread (32 B contiguous per lane)
            const size_t chunk = (size_t)(it >> 1) * 256 * 32;
            const f4 v = *(const f4 GAS*)(base + chunk + (size_t)tid * 32 + (it & 1) * 16);
            acc += v.x + v.w;
        }
        if (P == LD_X2_UNALIGNED_L2) { // colour taps: 8 B at a byte offset 3*x of a small image (L2 / L1 resident), consecutive lanes ~3.4 B apart
            typedef unsigned long long __attribute__((aligned(1), may_alias)) u64u;
            const size_t off = ((size_t)(blockIdx.x % 64) * 24576 + (size_t)(it % 8) * 3072 + (size_t)tid * 3 + (tid >> 3)) % (1536 * 1024 - 16);
            const unsigned long long v = *(const u64u GAS*)(buf + off);
            acc += (float)(unsigned)(v >> 40);
        }
        if (P == ST_X4) { f4 v = {acc, 1.0f, 2.0f, 3.0f}; *(f4 GAS*)(base + ((size_t)it * 256 + tid) * 16) = v; }
        if (P == ST_X3) { F3 v = {acc, 1.0f, 2.0f}; *(F3*)(base + ((size_t)it * 256 + tid) * 12) = v; }
        if (P == ST_X2) { f2 v = {acc, 1.0f}; *(f2 GAS*)(base + ((size_t)it * 256 + tid) * 8) = v; }
        if (P == ST_X1) { *(float GAS*)(base + ((size_t)it * 256 + tid) * 4) = acc; }
        if (P == ST_REC28) {           // the kernel's copy-out: per record 12 B + 12 B + 4 B into three arrays
            F3 v = {acc, 1.0f, 2.0f};
            *(F3*)(base + ((size_t)it * 256 + tid) * 12) = v;
            *(F3*)(base + per_wg / 2 + ((size_t)it * 256 + tid) * 12) = v;
            *(float GAS*)(base + per_wg / 2 + per_wg / 4 + ((size_t)it * 256 + tid) * 4) = acc;
        }
        if (P == ST_X4_UNALIGNED12) {  // 16-B stores whose base is only 4-byte aligned (a float stream that starts at 12 * prefix)
            f4 v = {acc, 1.0f, 2.0f, 3.0f};
            *(f4 GAS*)(base + 12 + ((size_t)it * 256 + tid) * 16) = v;
        }
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

template <int P>
void run(const char* name, char* buf, size_t buf_bytes, int bytes_per_thread_iter, float* sink) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 7;
    const size_t per_wg = (buf_bytes / blocks) & ~size_t(255);
    int iters = (int)(per_wg / ((size_t)256 * 32));          // stay inside the workgroup's slice for every pattern (<= 32 B per thread-iteration)
    if (P == ST_REC28) iters = (int)(per_wg / 2 / (256 * 12));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    pattern<P><<<blocks, 256>>>(buf, per_wg, iters, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    pattern<P><<<blocks, 256>>>(buf, per_wg, iters, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * 256 * iters * bytes_per_thread_iter;
    const double insts_per_cu = (double)7 * 4 * iters * (P == ST_REC28 ? 3 : 1);
    printf("%-58s %8.1f GB/s  %6.2f B/cycle/CU (at 2.1 GHz)  %7.1f ns per wave-instruction per CU\n", name, bytes / (ms * 1e-3) / 1e9,
           bytes / (ms * 1e-3) / p.multiProcessorCount / 2.1e9, ms * 1e6 / insts_per_cu);
}

int main() {
    const size_t bytes = (size_t)3 << 30;
    char* buf; float* sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 64)); CHK(hipMemset(buf, 0, bytes));
    run<LD_X4>("load  dwordx4 dense (1 KB / wave-instruction)", buf, bytes, 16, sink);
    run<LD_X2>("load  dwordx2 dense", buf, bytes, 8, sink);
    run<LD_X1>("load  dword   dense", buf, bytes, 4, sink);
    run<LD_X2_STRIDE32>("load  dwordx2 at 32-B stride, 4 instructions per span", buf, bytes, 8, sink);
    run<LD_X4_PAIR32>("load  dwordx4 x2 per thread (32 B contiguous per lane)", buf, bytes, 16, sink);
    run<LD_X2_UNALIGNED_L2>("load  8 B unaligned, ~3.4 B apart, cache resident (taps)", buf, bytes, 8, sink);
    run<ST_X4>("store dwordx4 dense", buf, bytes, 16, sink);
    run<ST_X4_UNALIGNED12>("store dwordx4 dense, base 4-byte aligned only", buf, bytes, 16, sink);
    run<ST_X3>("store dwordx3 dense (12-B records)", buf, bytes, 12, sink);
    run<ST_X2>("store dwordx2 dense", buf, bytes, 8, sink);
    run<ST_X1>("store dword   dense", buf, bytes, 4, sink);
    run<ST_REC28>("store 12 B + 12 B + 4 B per record (the kernel's copy-out)", buf, bytes, 28, sink);
    return 0;
}
