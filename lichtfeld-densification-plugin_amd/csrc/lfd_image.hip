// N3: image preparation on the device (upstream core/image_utils.py:40-91, core/pipeline.py:163-171).
//
// Upstream prepares every image on the host with Pillow: Image.resize(size, BILINEAR) for the RGB images, convert("L") +
// Image.resize(size, NEAREST) + "> threshold" for the masks, and masked pixels are blacked out before matching.  Decoding
// stays on the host (PIL); what follows runs here, bit for bit like Pillow 12 (libImaging/Resample.c, Geometry.c):
//
//   BILINEAR, 8 bits per channel: separable convolution, horizontal pass first, then vertical; per output pixel a window
//   [min, min+count) of the input and coefficients computed in f64 (triangle filter of support max(scale, 1), normalised to
//   sum 1) and rounded to 22-bit fixed point; a pass accumulates 2^21 + sum(pixel * k) in int32, shifts right by 22 and clamps
//   to [0, 255]; the value between the passes is 8-bit.  The kernel fuses the passes: one thread per output pixel recomputes
//   the horizontally resampled value of each row of its vertical window (a few dozen integer multiply-adds; the image is read
//   through the caches), which is exactly the two-pass result because the intermediate rounding is reproduced.
//   NEAREST: source index (int)offset with the offset ACCUMULATED in f64 from scale/2 (Pillow's affine scale path).
//
// The coefficient / index tables are built on the host in f64 exactly as Pillow builds them and cached per size pair.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "lfd_context.hpp"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the triangle filter over the whole input
int resize_tables(int in_size, int out_size, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;       // box coordinates are floats in Pillow
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) k[(size_t)x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            const double w = v < 1.0 ? 1.0 - v : 0.0;
            k[(size_t)x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[(size_t)x] /= ww;
        for (int x = 0; x < ksize; ++x) {
            const double p = k[(size_t)x] * (double)(1 << kPrecisionBits);
            kk[(size_t)xx * ksize + x] = k[(size_t)x] < 0 ? (int)(-0.5 + p) : (int)(0.5 + p);
        }
        bounds[(size_t)xx * 2 + 0] = xmin;
        bounds[(size_t)xx * 2 + 1] = xmax;
    }
    return ksize;
}

void nearest_indices(int in_size, int out_size, std::vector<int32_t>& idx) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    double xo = 0.0 + scale * 0.5;
    idx.assign((size_t)out_size, 0);
    for (int x = 0; x < out_size; ++x) {
        int xin = xo < 0.0 ? -1 : (int)xo;
        if (xin < 0) xin = 0;
        if (xin > in_size - 1) xin = in_size - 1;
        idx[(size_t)x] = xin;
        xo += scale;
    }
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

}  // namespace

// The same resize with the passes in the other order - vertical first, the vertically resampled pixel of every source column rounded to 8 bits,
// then horizontal: what Pillow's Image.resize does for an image more than 100 times taller than wide that shrinks vertically (PIL/Image.py:
// two resize calls).  tab as below.
extern "C" __global__ void __launch_bounds__(256) lfd_resize_bilinear_vfirst_kernel(const uint8_t* __restrict__ src, int w_in, int h_in,
                                                                                    uint8_t* __restrict__ dst, int w_out, int h_out,
                                                                                    const int32_t* __restrict__ tab, int ks_x, int ks_y,
                                                                                    const uint8_t* __restrict__ mask01) {
    const int ox = (int)(blockIdx.x * 64 + (threadIdx.x & 63));
    const int oy = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (ox >= w_out || oy >= h_out) return;
    const int32_t* bx = tab;
    const int32_t* kx = bx + 2 * w_out;
    const int32_t* by = kx + (size_t)w_out * ks_x;
    const int32_t* ky = by + 2 * h_out;
    const bool horiz = w_out != w_in;                 // (vertical: always - the caller's rule)
    const int xmin = horiz ? bx[2 * ox] : ox, xcnt = horiz ? bx[2 * ox + 1] : 1;
    const int ymin = by[2 * oy], ycnt = by[2 * oy + 1];
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    int o0 = 0, o1 = 0, o2 = 0;
    for (int tx = 0; tx < xcnt; ++tx) {
        const uint8_t* col = src + ((size_t)ymin * w_in + (xmin + tx)) * 3;
        int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
        for (int ty = 0; ty < ycnt; ++ty) {
            const int k = ky[(size_t)oy * ks_y + ty];
            const uint8_t* px = col + (size_t)ty * w_in * 3;
            s0 += (int)px[0] * k; s1 += (int)px[1] * k; s2 += (int)px[2] * k;
        }
        const int v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2);
        if (horiz) {
            const int k = kx[(size_t)ox * ks_x + tx];
            a0 += v0 * k; a1 += v1 * k; a2 += v2 * k;
        } else {
            o0 = v0; o1 = v1; o2 = v2;
        }
    }
    if (horiz) { o0 = clip8(a0); o1 = clip8(a1); o2 = clip8(a2); }
    if (mask01 && mask01[(size_t)oy * w_out + ox] == 0) o0 = o1 = o2 = 0;
    uint8_t* d = dst + ((size_t)oy * w_out + ox) * 3;
    d[0] = (uint8_t)o0; d[1] = (uint8_t)o1; d[2] = (uint8_t)o2;
}

// tab: [bounds_x (2*w_out) | kk_x (w_out*ks_x) | bounds_y (2*h_out) | kk_y (h_out*ks_y)]
extern "C" __global__ void __launch_bounds__(256) lfd_resize_bilinear_kernel(const uint8_t* __restrict__ src, int w_in, int h_in,
                                                                             uint8_t* __restrict__ dst, int w_out, int h_out,
                                                                             const int32_t* __restrict__ tab, int ks_x, int ks_y,
                                                                             const uint8_t* __restrict__ mask01) {
    const int ox = (int)(blockIdx.x * 64 + (threadIdx.x & 63));
    const int oy = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (ox >= w_out || oy >= h_out) return;
    const int32_t* bx = tab;
    const int32_t* kx = bx + 2 * w_out;
    const int32_t* by = kx + (size_t)w_out * ks_x;
    const int32_t* ky = by + 2 * h_out;
    const bool horiz = w_out != w_in, vert = h_out != h_in;
    const int xmin = horiz ? bx[2 * ox] : ox, xcnt = horiz ? bx[2 * ox + 1] : 1;
    const int ymin = vert ? by[2 * oy] : oy, ycnt = vert ? by[2 * oy + 1] : 1;
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    int o0 = 0, o1 = 0, o2 = 0;
    for (int ty = 0; ty < ycnt; ++ty) {
        const uint8_t* row = src + ((size_t)(ymin + ty) * w_in + xmin) * 3;
        int h0, h1, h2;
        if (horiz) {                         // the horizontally resampled pixel of this row, rounded to 8 bits like Pillow's intermediate image
            int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
            for (int tx = 0; tx < xcnt; ++tx) {
                const int k = kx[(size_t)ox * ks_x + tx];
                s0 += (int)row[3 * tx + 0] * k; s1 += (int)row[3 * tx + 1] * k; s2 += (int)row[3 * tx + 2] * k;
            }
            h0 = clip8(s0); h1 = clip8(s1); h2 = clip8(s2);
        } else {
            h0 = row[0]; h1 = row[1]; h2 = row[2];
        }
        if (vert) {
            const int k = ky[(size_t)oy * ks_y + ty];
            a0 += h0 * k; a1 += h1 * k; a2 += h2 * k;
        } else {
            o0 = h0; o1 = h1; o2 = h2;
        }
    }
    if (vert) { o0 = clip8(a0); o1 = clip8(a1); o2 = clip8(a2); }
    if (mask01 && mask01[(size_t)oy * w_out + ox] == 0) o0 = o1 = o2 = 0;      // apply_mask_to_rgb: masked pixels become black
    uint8_t* d = dst + ((size_t)oy * w_out + ox) * 3;
    d[0] = (uint8_t)o0; d[1] = (uint8_t)o1; d[2] = (uint8_t)o2;
}

// tab: [idx_x (w_out) | idx_y (h_out) | lut (256)]
extern "C" __global__ void __launch_bounds__(256) lfd_mask_nearest_kernel(const uint8_t* __restrict__ src, int w_in, uint8_t* __restrict__ dst,
                                                                          int w_out, int h_out, const int32_t* __restrict__ tab) {
    const int ox = (int)(blockIdx.x * 64 + (threadIdx.x & 63));
    const int oy = (int)(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (ox >= w_out || oy >= h_out) return;
    const int32_t* ix = tab;
    const int32_t* iy = tab + w_out;
    const int32_t* lut = iy + h_out;
    dst[(size_t)oy * w_out + ox] = (uint8_t)lut[src[(size_t)iy[oy] * w_in + ix[ox]]];
}

namespace {

#define LFD_IMG_HIP(ctx, expr)                                                                         \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return lfd_fail((ctx), LFD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int upload_table(lfd_context* ctx, DeviceBuffer& buf, const std::vector<int32_t>& host) {
    const size_t bytes = host.size() * sizeof(int32_t);
    if (buf.bytes < bytes || !buf.ptr) {
        if (buf.ptr) { LFD_IMG_HIP(ctx, hipStreamSynchronize(ctx->stream)); LFD_IMG_HIP(ctx, hipFree(buf.ptr)); buf.ptr = nullptr; buf.bytes = 0; }
        LFD_IMG_HIP(ctx, hipMalloc(&buf.ptr, (bytes + 255) & ~size_t(255)));
        buf.bytes = (bytes + 255) & ~size_t(255);
    }
    LFD_IMG_HIP(ctx, hipMemcpyAsync(buf.ptr, host.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    LFD_IMG_HIP(ctx, hipStreamSynchronize(ctx->stream));     // the host vector goes out of scope
    return LFD_OK;
}

}  // namespace

extern "C" {

int lfd_prepare_image(lfd_context* ctx, const uint8_t* src_rgb, int32_t w_in, int32_t h_in, int32_t w_out, int32_t h_out,
                      const uint8_t* mask01, uint8_t* dst_rgb) {
    if (!ctx) return lfd_fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return lfd_fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (!src_rgb || !dst_rgb || w_in <= 0 || h_in <= 0 || w_out <= 0 || h_out <= 0) return lfd_fail(ctx, LFD_ERR_INVALID, "bad arguments");
    LFD_IMG_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->img_key[0] != w_in || ctx->img_key[1] != h_in || ctx->img_key[2] != w_out || ctx->img_key[3] != h_out || !ctx->img_tab.ptr) {
        std::vector<int32_t> bx, kx, by, ky, all;
        ctx->img_ks[0] = resize_tables(w_in, w_out, bx, kx);
        ctx->img_ks[1] = resize_tables(h_in, h_out, by, ky);
        all.insert(all.end(), bx.begin(), bx.end()); all.insert(all.end(), kx.begin(), kx.end());
        all.insert(all.end(), by.begin(), by.end()); all.insert(all.end(), ky.begin(), ky.end());
        int rc = upload_table(ctx, ctx->img_tab, all);
        if (rc != LFD_OK) return rc;
        ctx->img_key[0] = w_in; ctx->img_key[1] = h_in; ctx->img_key[2] = w_out; ctx->img_key[3] = h_out;
    }
    const dim3 grid((unsigned)((w_out + 63) / 64), (unsigned)((h_out + 3) / 4));
    // Pillow's Image.resize: "if self.size[1] > self.size[0] * 100 and size[1] < self.size[1]" - vertical pass first
    const bool vfirst = (long long)h_in > (long long)w_in * 100 && h_out < h_in;
    hipLaunchKernelGGL(vfirst ? lfd_resize_bilinear_vfirst_kernel : lfd_resize_bilinear_kernel, grid, dim3(256), 0, ctx->stream, src_rgb, w_in, h_in,
                       dst_rgb, w_out, h_out, static_cast<const int32_t*>(ctx->img_tab.ptr), ctx->img_ks[0], ctx->img_ks[1], mask01);
    LFD_IMG_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_prepare_mask(lfd_context* ctx, const uint8_t* src_l, int32_t w_in, int32_t h_in, int32_t w_out, int32_t h_out,
                     float threshold, int32_t invert, uint8_t* dst01) {
    if (!ctx) return lfd_fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return lfd_fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (!src_l || !dst01 || w_in <= 0 || h_in <= 0 || w_out <= 0 || h_out <= 0) return lfd_fail(ctx, LFD_ERR_INVALID, "bad arguments");
    LFD_IMG_HIP(ctx, hipSetDevice(ctx->device));
    const int inv = invert ? 1 : 0;
    if (ctx->msk_key[0] != w_in || ctx->msk_key[1] != h_in || ctx->msk_key[2] != w_out || ctx->msk_key[3] != h_out ||
        ctx->msk_thr != threshold || ctx->msk_inv != inv || !ctx->msk_tab.ptr) {
        std::vector<int32_t> ix, iy, all;
        if (w_out == w_in) { ix.resize((size_t)w_out); for (int i = 0; i < w_out; ++i) ix[(size_t)i] = i; } else nearest_indices(w_in, w_out, ix);
        if (h_out == h_in) { iy.resize((size_t)h_out); for (int i = 0; i < h_out; ++i) iy[(size_t)i] = i; } else nearest_indices(h_in, h_out, iy);
        // upstream resizes only when the size differs, and then in both directions at once: Pillow's affine path handles an axis of
        // unchanged length with scale 1 and offset 0.5, i.e. the identity, so the per-axis tables above are the same thing
        all.insert(all.end(), ix.begin(), ix.end()); all.insert(all.end(), iy.begin(), iy.end());
        for (int v = 0; v < 256; ++v) {          // (arr.astype(float32) / 255.0) > threshold, compared in f32 (core/image_utils.py:61-63)
            const bool keep = ((float)v / 255.0f) > threshold;
            all.push_back((keep != (inv != 0)) ? 1 : 0);
        }
        int rc = upload_table(ctx, ctx->msk_tab, all);
        if (rc != LFD_OK) return rc;
        ctx->msk_key[0] = w_in; ctx->msk_key[1] = h_in; ctx->msk_key[2] = w_out; ctx->msk_key[3] = h_out;
        ctx->msk_thr = threshold; ctx->msk_inv = inv;
    }
    const dim3 grid((unsigned)((w_out + 63) / 64), (unsigned)((h_out + 3) / 4));
    hipLaunchKernelGGL(lfd_mask_nearest_kernel, grid, dim3(256), 0, ctx->stream, src_l, w_in, dst01, w_out, h_out,
                       static_cast<const int32_t*>(ctx->msk_tab.ptr));
    LFD_IMG_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_host_resize_tables(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk, int32_t kk_capacity, int32_t* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !kk || !ksize_out) return LFD_ERR_INVALID;
    std::vector<int32_t> b, k;
    const int ks = resize_tables(in_size, out_size, b, k);
    *ksize_out = ks;
    if ((long long)kk_capacity < (long long)out_size * ks) return LFD_ERR_CAPACITY;
    std::memcpy(bounds, b.data(), b.size() * sizeof(int32_t));
    std::memcpy(kk, k.data(), k.size() * sizeof(int32_t));
    return LFD_OK;
}

int lfd_host_nearest_indices(int32_t in_size, int32_t out_size, int32_t* idx) {
    if (in_size <= 0 || out_size <= 0 || !idx) return LFD_ERR_INVALID;
    std::vector<int32_t> v;
    nearest_indices(in_size, out_size, v);
    std::memcpy(idx, v.data(), v.size() * sizeof(int32_t));
    return LFD_OK;
}

}  // extern "C"
