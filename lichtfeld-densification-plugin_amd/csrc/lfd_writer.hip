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

__device__ __forceinline__ unsigned char quantise_u8(float c) { return lfd_quantise_u8(c); }

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

// ---- tile segments (lfd_triangulate_dense_segments) ------------------------------------------------------------------------------------
// The unordered dense kernel leaves reference r's survivors in [r*HW, r*HW + count_r) of the output buffers, tile after tile in the order the
// tiles retired; the tile table says where each tile went.  Raster order = tile order, so the ordered position of tile t's first record is
// the exclusive prefix of the table's counts (lfd_segment_scan_kernel), and record d of the ordered sequence is found by looking d up in
// that prefix: once per 256-record block over the whole table, then per record over the block's few tiles.
struct SegMap {
    const long long* tile_dst;      // [n_tiles + 1] exclusive prefix of the counts, in tile (= raster) order
    const LfdTileSeg* table;
    long long hw;
    int tiles_per_ref, n_tiles;
    // [n_tiles + 2] behind tile_dst: the tile that holds ordered record 1024 c (a tile has at most 1024 survivors, so there are at most
    // n_tiles such records; entries past the total hold the last tile) - a record's tile is then a search over the two or three tiles
    // between two consecutive entries instead of over the whole prefix
    __device__ const int* chunk_tile() const { return reinterpret_cast<const int*>(tile_dst + n_tiles + 1); }
};

// last tile t in [lo, hi] with tile_dst[t] <= d  (tile_dst is non-decreasing; tile_dst[lo] <= d is given)
__device__ __forceinline__ int seg_find(const long long* tile_dst, int lo, int hi, long long d) {
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_dst[mid] <= d) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The window of the block of ordered records [base, base + span): the tiles it touches (threads 0 and 1 search the whole prefix), their
// first ordered record and first SOURCE record staged in LDS (at most kSegWin tiles: a tile holds up to 1024 survivors, so a 1024-record
// block touches two or three unless many tiles are empty; a wider window is searched in memory instead).
constexpr int kSegWin = 64;
struct SegWindow {
    int t_lo, t_hi;
    long long dst[kSegWin + 1];
    long long src[kSegWin];
};

__device__ __forceinline__ void seg_window(const SegMap& m, long long base, long long span, long long n, int tid, int nthreads, SegWindow& w) {
    if (tid < 2) {
        long long d = tid == 0 ? base : base + span - 1;
        if (d > n - 1) d = n - 1;
        const int* ct = m.chunk_tile();
        const long long c = d >> 10;
        const int t = seg_find(m.tile_dst, ct[c], ct[c + 1], d);
        if (tid == 0) w.t_lo = t; else w.t_hi = t;
    }
    __syncthreads();
    const int cnt = w.t_hi - w.t_lo + 1;
    if (cnt <= kSegWin) {
        for (int i = tid; i < cnt; i += nthreads) {
            const int t = w.t_lo + i;
            w.dst[i] = m.tile_dst[t];
            w.src[i] = (long long)(t / m.tiles_per_ref) * m.hw + (long long)m.table[t].offset;
        }
        if (tid == 0) w.dst[cnt] = m.tile_dst[w.t_hi + 1];
    }
    __syncthreads();
}

// source index of ordered record d of the block whose window is w
__device__ __forceinline__ long long seg_source(const SegMap& m, const SegWindow& w, long long d) {
    const int cnt = w.t_hi - w.t_lo + 1;
    if (cnt <= kSegWin) {
        int i = 0;
        while (i + 1 < cnt && w.dst[i + 1] <= d) ++i;       // (two or three steps)
        return w.src[i] + (d - w.dst[i]);
    }
    const int t = seg_find(m.tile_dst, w.t_lo, w.t_hi, d);
    return (long long)(t / m.tiles_per_ref) * m.hw + (long long)m.table[t].offset + (d - m.tile_dst[t]);
}

}  // namespace

// exclusive prefix of the tile table's counts in tile order (one workgroup; a launch has 10^4 .. 10^5 tiles), the ordered ref_offsets
// (device i64 [n_refs + 1], may be null) and the total at tile_dst[n_tiles]
extern "C" __global__ void __launch_bounds__(1024) lfd_segment_scan_kernel(const LfdTileSeg* __restrict__ table, int n_tiles, int tiles_per_ref,
                                                                          int n_refs, long long* __restrict__ tile_dst,
                                                                          long long* __restrict__ ref_offsets) {
    __shared__ long long wave_sum[16];
    __shared__ long long carry;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* chunk_tile = reinterpret_cast<int*>(tile_dst + n_tiles + 1);
    for (int i = tid; i < n_tiles + 2; i += 1024) chunk_tile[i] = n_tiles - 1;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int t = base + tid;
        const long long c = t < n_tiles ? (long long)table[t].count : 0;
        long long incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned lo = __shfl_up((unsigned)incl, off, 64), hi = __shfl_up((unsigned)((unsigned long long)incl >> 32), off, 64);
            const long long up = (long long)(((unsigned long long)hi << 32) | lo);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        const long long excl = before + incl - c;
        if (t < n_tiles) {
            tile_dst[t] = excl;
            if (ref_offsets && t % tiles_per_ref == 0) ref_offsets[t / tiles_per_ref] = excl;
            for (long long ch = (excl + 1023) >> 10; (ch << 10) < excl + c; ++ch) chunk_tile[ch] = t;      // (at most two: c <= 1024)
        }
        __syncthreads();
        if (tid == 1023) carry = excl + c;
        __syncthreads();
    }
    if (tid == 0) {
        tile_dst[n_tiles] = carry;
        if (ref_offsets) ref_offsets[n_refs] = carry;
    }
}

// the ordered structure-of-arrays copy of a segmented result (what the ordered kernel would have written): record d <- its source
extern "C" __global__ void __launch_bounds__(256) lfd_order_segments_kernel(const long long* __restrict__ tile_dst, const LfdTileSeg* __restrict__ table,
                                                                            long long hw, int tiles_per_ref, int n_tiles,
                                                                            const float* __restrict__ sxyz, const float* __restrict__ srgb,
                                                                            const float* __restrict__ serr, const int* __restrict__ scell,
                                                                            const unsigned char* __restrict__ sslot, float* __restrict__ dxyz,
                                                                            float* __restrict__ drgb, float* __restrict__ derr, int* __restrict__ dcell,
                                                                            unsigned char* __restrict__ dslot, long long capacity) {
    __shared__ SegWindow win;
    const SegMap m{tile_dst, table, hw, tiles_per_ref, n_tiles};
    const int tid = (int)threadIdx.x;
    long long n = tile_dst[n_tiles];
    if (n > capacity) n = capacity;
    for (long long base = (long long)blockIdx.x * 1024; base < n; base += (long long)gridDim.x * 1024) {
        seg_window(m, base, 1024, n, tid, 256, win);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long d = base + tid + 256 * u;
            if (d < n) {
                const long long s = seg_source(m, win, d);
#pragma unroll
                for (int c = 0; c < 3; ++c) { dxyz[3 * d + c] = sxyz[3 * s + c]; drgb[3 * d + c] = srgb[3 * s + c]; }
                derr[d] = serr[s];
                if (dcell && scell) dcell[d] = scell[s];
                if (dslot && sslot) dslot[d] = sslot[s];
            }
        }
        __syncthreads();
    }
}

// lfd_pack_ply_kernel / lfd_pack_points3d_kernel reading through the tile table: the file payload in raster order straight from the
// unordered buffers (no ordered copy of the 28-byte records in between)
extern "C" __global__ void __launch_bounds__(256) lfd_pack_ply_segments_kernel(const long long* __restrict__ tile_dst, const LfdTileSeg* __restrict__ table,
                                                                               long long hw, int tiles_per_ref, int n_tiles,
                                                                               const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                                               long long capacity, unsigned char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[256 * 15];
    __shared__ SegWindow win;
    const SegMap m{tile_dst, table, hw, tiles_per_ref, n_tiles};
    const int tid = (int)threadIdx.x;
    long long n = tile_dst[n_tiles];
    if (n > capacity) n = capacity;
    for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
        seg_window(m, base, 256, n, tid, 256, win);
        const long long d = base + tid;
        if (d < n) {
            const long long i = seg_source(m, win, d);
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

extern "C" __global__ void __launch_bounds__(256) lfd_pack_points3d_segments_kernel(const long long* __restrict__ tile_dst, const LfdTileSeg* __restrict__ table,
                                                                                    long long hw, int tiles_per_ref, int n_tiles,
                                                                                    const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                                                    const float* __restrict__ err, long long capacity,
                                                                                    unsigned long long id_base, unsigned char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char rec[256 * 43];
    __shared__ SegWindow win;
    const SegMap m{tile_dst, table, hw, tiles_per_ref, n_tiles};
    const int tid = (int)threadIdx.x;
    auto put64 = [](unsigned char* p, unsigned long long u) {
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = (unsigned char)(u >> (8 * b));
    };
    long long n = tile_dst[n_tiles];
    if (n > capacity) n = capacity;
    for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
        seg_window(m, base, 256, n, tid, 256, win);
        const long long d = base + tid;
        if (d < n) {
            const long long i = seg_source(m, win, d);
            unsigned char* r = rec + tid * 43;
            put64(r, id_base + (unsigned long long)d + 1ull);
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


// ---- lfd_copy_segments: many (source offset, destination offset, bytes) copies in ONE launch -------------------------------------------------------
// The overlapped exchange of a sharded run (core/distributed.py) receives every rank's records of a round as one padded block per rank and owes
// each reference's records their place in the ordered cloud: world x references-per-round copies per round, a few MB each.  As separate copies they
// cost more host time than the kernel that produced the records; here they are one launch.  Offsets and lengths are in BYTES and need no alignment
// (15-byte PLY records): the destination is written in aligned 16-byte units, the source read with whatever alignment it has.
struct __attribute__((packed, aligned(1))) LfdU4u { unsigned x, y, z, w; };

extern "C" __global__ void __launch_bounds__(256) lfd_copy_segments_kernel(LfdCopyArgs A, const unsigned char* __restrict__ src, unsigned char* __restrict__ dst) {
    const int tid = (int)threadIdx.x;
    const int b = (int)blockIdx.x;
    int lo = 0, hi = A.n_segs;                     // chunk0[lo] <= b < chunk0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.chunk0[mid] <= b) lo = mid; else hi = mid;
    }
    const long long off = (long long)(b - A.chunk0[lo]) * LFD_COPY_CHUNK;
    long long len = A.n[lo] - off;
    if (len > LFD_COPY_CHUNK) len = LFD_COPY_CHUNK;
    if (len <= 0) return;
    const unsigned char* s = src + A.src[lo] + off;
    unsigned char* d = dst + A.dst[lo] + off;
    int head = (int)((16u - (unsigned)(reinterpret_cast<unsigned long long>(d) & 15u)) & 15u);
    if (head > len) head = (int)len;
    if (tid < head) d[tid] = s[tid];
    const long long body = (len - head) >> 4;
    const LfdU4u* sv = reinterpret_cast<const LfdU4u*>(s + head);
    uint4* dv = reinterpret_cast<uint4*>(d + head);
    for (long long i = tid; i < body; i += 256) {
        const LfdU4u v = sv[i];
        dv[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
    const long long done = head + (body << 4);
    if (tid < len - done) d[done + tid] = s[done + tid];
}
