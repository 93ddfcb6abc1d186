// Host side of the C-ABI declared in include/lfd_densify.h: context, tables, launches.
// No compute happens on the host here except the tiny helpers the header lists as host-side.
#include <cstddef>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lfd_context.hpp"

extern "C" __global__ void lfd_aggregate_kernel(LfdLaunch L, float* best_cert, uint8_t* best_slot);
extern "C" __global__ void lfd_dense_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_exact_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_ply_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_ply_exact_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_segments_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_segments_exact_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_ply_segments_kernel(LfdLaunch L);
extern "C" __global__ void lfd_dense_ply_segments_exact_kernel(LfdLaunch L);
extern "C" __global__ void lfd_segment_scan_kernel(const LfdTileSeg* table, int n_tiles, int tiles_per_ref, int n_refs, long long* tile_dst, long long* ref_offsets);
extern "C" __global__ void lfd_order_segments_kernel(const long long* tile_dst, const LfdTileSeg* table, long long hw, int tiles_per_ref, int n_tiles,
                                                     const float* sxyz, const float* srgb, const float* serr, const int* scell, const unsigned char* sslot,
                                                     float* dxyz, float* drgb, float* derr, int* dcell, unsigned char* dslot, long long capacity);
extern "C" __global__ void lfd_pack_ply_segments_kernel(const long long* tile_dst, const LfdTileSeg* table, long long hw, int tiles_per_ref, int n_tiles,
                                                        const float* xyz, const float* rgb, long long capacity, unsigned char* out);
extern "C" __global__ void lfd_pack_points3d_segments_kernel(const long long* tile_dst, const LfdTileSeg* table, long long hw, int tiles_per_ref, int n_tiles,
                                                             const float* xyz, const float* rgb, const float* err, long long capacity,
                                                             unsigned long long id_base, unsigned char* out);
extern "C" __global__ void lfd_pair_setup_kernel(LfdLaunch L, LfdRefConst* ref_out, LfdPairConst* pair_out);
extern "C" __global__ void lfd_select_filter_kernel(LfdSelectArgs A);
extern "C" __global__ void lfd_pack_ply_kernel(const float* xyz, const float* rgb, long long n, unsigned char* out);
extern "C" __global__ void lfd_pack_points3d_kernel(const float* xyz, const float* rgb, const float* err, long long n,
                                                    unsigned long long id_base, unsigned char* out);
extern "C" __global__ void lfd_copy_segments_kernel(LfdCopyArgs A, const unsigned char* src, unsigned char* dst);
extern "C" __global__ void lfd_quantise_rgb_kernel(const float* rgb, long long n3, unsigned char* out);
extern "C" __global__ void lfd_select_topm_kernel(LfdSelectArgs A);
extern "C" __global__ void lfd_select_filter_mw_kernel(LfdSelectArgs A, LfdSelectNorms norms);
extern "C" __global__ void lfd_select_begins_kernel(long long* pairs, long long stride, int n);
extern "C" __global__ void lfd_mt_seed_kernel(unsigned* mt, unsigned seed);
extern "C" __global__ void lfd_mt_seed_batch_kernel(unsigned* mt_base, LfdSeedBatch seeds);
extern "C" __global__ void lfd_indexed_eval_kernel(LfdLaunch L, const long long* sel_idx, const long long* sel_offsets, float* scratch,
                                                   uint8_t* codes, unsigned* tab, int off_pairs);
extern "C" __global__ void lfd_indexed_kernel(LfdLaunch L, const long long* sel_idx, const long long* sel_offsets,
                                              float* scratch, uint8_t* codes, int32_t* seg_order, const unsigned* tab, int off_pairs);

namespace {

// the one piece of process-wide state: the message of the last lfd_create() that failed (there is no context to hold it).
// Written under a mutex; lfd_last_error(NULL) hands out a per-thread copy, so concurrent creators cannot tear it.
std::mutex g_create_mutex;
std::string g_create_error;
thread_local std::string t_create_error_copy;

}  // namespace

namespace {

int fail(lfd_context* ctx, int code, const std::string& msg) { return lfd_fail(ctx, code, msg); }

#define LFD_HIP(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((ctx), LFD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

int ensure(lfd_context* ctx, DeviceBuffer& b, size_t bytes, bool zero = false) {
    if (b.bytes >= bytes && b.ptr) return LFD_OK;
    if (b.ptr) {
        LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->prep_stream) LFD_HIP(ctx, hipStreamSynchronize(ctx->prep_stream));
        LFD_HIP(ctx, hipFree(b.ptr));
        b.ptr = nullptr; b.bytes = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    want = (want + 255) & ~size_t(255);
    LFD_HIP(ctx, hipMalloc(&b.ptr, want));
    b.bytes = want;
    if (zero) LFD_HIP(ctx, hipMemsetAsync(b.ptr, 0, want, ctx->stream));
    return LFD_OK;
}

int validate_batch(lfd_context* ctx, const lfd_batch* b, const lfd_params* p) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "this context was made by lfd_create_host: it serves the *_host entry points only");
    if (!b || !p) return fail(ctx, LFD_ERR_INVALID, "null batch/params");
    if (ctx->n_cams <= 0) return fail(ctx, LFD_ERR_STATE, "lfd_upload_cameras must be called first");
    if (b->n_refs <= 0) return fail(ctx, LFD_ERR_INVALID, "n_refs must be > 0");
    if (b->k <= 0 || b->k > LFD_MAX_SLOTS) return fail(ctx, LFD_ERR_INVALID, "k must be in [1, LFD_MAX_SLOTS]");
    if (b->H <= 0 || b->W <= 0 || b->w_match <= 1 || b->h_match <= 1) return fail(ctx, LFD_ERR_INVALID, "bad grid / match size");
    // byte offsets into the match-size image are 32-bit, its rows and columns 24-bit multiplicands (lfd_bilinear_fetch)
    if (b->w_match >= (1 << 24) || b->h_match >= (1 << 24) || (long long)b->w_match * b->h_match * 3 > 0x7fffffffLL)
        return fail(ctx, LFD_ERR_INVALID, "match-size image too large");
    if ((long long)b->H * b->W > 0x7fffffffLL) return fail(ctx, LFD_ERR_INVALID, "grid too large");
    if (b->warp_channels != 2 && b->warp_channels != 4) return fail(ctx, LFD_ERR_INVALID, "warp_channels must be 2 or 4");
    if (!b->ref_cam || !b->n_slots || !b->nbr_cam || !b->cert || !b->warp || !b->image)
        return fail(ctx, LFD_ERR_INVALID, "null table in batch");
    if ((b->axis_x == nullptr) != (b->axis_y == nullptr)) return fail(ctx, LFD_ERR_INVALID, "axis_x and axis_y must both be given or both be null");
    for (int r = 0; r < b->n_refs; ++r) {
        if (b->ref_cam[r] < 0 || b->ref_cam[r] >= ctx->n_cams) return fail(ctx, LFD_ERR_INVALID, "ref_cam out of range");
        if (b->n_slots[r] < 1 || b->n_slots[r] > b->k) return fail(ctx, LFD_ERR_INVALID, "n_slots must be in [1, k]");
        if (!b->image[r]) return fail(ctx, LFD_ERR_INVALID, "null image pointer");
        for (int j = 0; j < b->n_slots[r]; ++j) {
            const size_t s = (size_t)r * b->k + j;
            if (b->nbr_cam[s] < 0 || b->nbr_cam[s] >= ctx->n_cams) return fail(ctx, LFD_ERR_INVALID, "nbr_cam out of range");
            if (!b->cert[s] || !b->warp[s]) return fail(ctx, LFD_ERR_INVALID, "null cert/warp pointer in a valid slot");
            if ((reinterpret_cast<uintptr_t>(b->cert[s]) & 15u) || (reinterpret_cast<uintptr_t>(b->warp[s]) & 15u))
                return fail(ctx, LFD_ERR_INVALID, "cert/warp planes must be 16-byte aligned");
        }
    }
    return LFD_OK;
}

int slot_events(lfd_context* ctx, LfdBatchSlot& sl) {
    if (!sl.pinned_free) LFD_HIP(ctx, hipEventCreateWithFlags(&sl.pinned_free, hipEventDisableTiming));
    if (!sl.ready) LFD_HIP(ctx, hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    if (!sl.idle) LFD_HIP(ctx, hipEventCreateWithFlags(&sl.idle, hipEventDisableTiming));
    return LFD_OK;
}

// Build refs|slots|fundamental matrices into one blob and find it a slot: a slot that already holds these tables is reused as it is;
// otherwise the blob is uploaded - by a launch into the slot of the last launch, on the launch stream (stream order protects the kernels
// still reading it); by lfd_prepare_batch (`ahead`) into the OTHER slot, on the preparation stream, behind the event that marks the end
// of that slot's last user - so that it runs beside the kernels of the batch before it.
// `extra` (the selection offsets of lfd_triangulate_indexed) is NOT part of what is cached and compared: it sits behind the tables and
// is uploaded on its own, on the launch stream, by the launch that brings it - so a batch staged by lfd_prepare_batch serves the indexed
// entry point as well (round 3 kept the offsets inside the blob: a staged batch then never matched).
int upload_tables(lfd_context* ctx, const lfd_batch* b, const long long* extra, size_t n_extra, bool ahead, int* slot_out, hipStream_t* stream_out,
                  const LfdRefDesc** d_refs, const LfdSlotDesc** d_slots, const long long** d_extra, const float** d_fund) {
    const size_t nr = (size_t)b->n_refs, ns = nr * (size_t)b->k;
    const size_t off_slots = (nr * sizeof(LfdRefDesc) + 15) & ~size_t(15);
    const size_t off_fund = (off_slots + ns * sizeof(LfdSlotDesc) + 15) & ~size_t(15);
    const size_t total = (off_fund + (b->fundamental ? ns * 9 * sizeof(float) : 0) + 15) & ~size_t(15);       // what is cached
    const size_t off_extra = total;
    const size_t extra_room = ((nr + 1) * sizeof(long long) + 15) & ~size_t(15);                              // always reserved: [n_refs + 1] offsets
    std::vector<unsigned char> blob(total, 0);
    LfdRefDesc* refs = reinterpret_cast<LfdRefDesc*>(blob.data());
    LfdSlotDesc* slots = reinterpret_cast<LfdSlotDesc*>(blob.data() + off_slots);
    for (size_t r = 0; r < nr; ++r) {
        refs[r].image = b->image[r];
        refs[r].mask_a = b->mask_a ? b->mask_a[r] : nullptr;
        refs[r].cam = b->ref_cam[r];
        refs[r].n_slots = b->n_slots[r];
        refs[r].any_mask = refs[r].mask_a ? 1 : 0;
        refs[r].pad = 0;
        for (int j = 0; j < b->k; ++j) {
            LfdSlotDesc& s = slots[r * b->k + j];
            const bool valid = j < b->n_slots[r];
            s.cert = valid ? b->cert[r * b->k + j] : nullptr;
            s.warp = valid ? b->warp[r * b->k + j] : nullptr;
            s.mask_b = (valid && b->mask_b) ? b->mask_b[r * b->k + j] : nullptr;
            s.cam = valid ? b->nbr_cam[r * b->k + j] : 0;
            s.pad = 0;
            if (s.mask_b) refs[r].any_mask = 1;
        }
    }
    if (b->fundamental) std::memcpy(blob.data() + off_fund, b->fundamental, ns * 9 * sizeof(float));
    if (n_extra * sizeof(long long) > extra_room) return fail(ctx, LFD_ERR_INVALID, "internal: selection offsets beyond the reserved room");

    int si = -1;
    for (int c = 0; c < 2; ++c) {
        const int i = c == 0 ? ctx->cur : 1 - ctx->cur;
        if (ctx->slot[i].desc.ptr && ctx->slot[i].cache == blob) { si = i; break; }
    }
    hipStream_t st = ctx->stream;
    if (si >= 0) {                                   // the tables are (or are about to be) in place
        LfdBatchSlot& sl = ctx->slot[si];
        if (!ahead && sl.ready_pending) {            // staged ahead on the preparation stream: the launch stream waits for that, once
#if !defined(LFD_EXPERIMENT_NO_READY_WAIT)           // (measurement only, UNSAFE: what the wait packet costs a step - profiles/r4/ab_ready_wait.txt)
            LFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.ready, 0));
#endif
            sl.ready_pending = false;
        }
        if (ahead && sl.ready_pending) st = ctx->prep_stream;      // (a second lfd_prepare_batch of the same batch: stay behind the first)
    } else {
        if (ahead) {
            if (!ctx->prep_stream) LFD_HIP(ctx, hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking));
            si = ctx->last_used < 0 ? ctx->cur : 1 - ctx->cur;
            st = ctx->prep_stream;
        } else {
            si = ctx->cur;
        }
        LfdBatchSlot& sl = ctx->slot[si];
        int rc = slot_events(ctx, sl);
        if (rc != LFD_OK) return rc;
        if (ahead) {
            // the kernels of this slot's last user must be through (the event sits where the launch after it began), and so must
            // whatever a launch issued for the slot in the launch stream since (si == cur only before the first launch)
            if (sl.used) LFD_HIP(ctx, hipStreamWaitEvent(st, sl.idle_attached ? sl.idle_ext : sl.idle, 0));
        } else if (sl.ready_pending) {               // an unused staging of another batch is still on its way into this slot
            LFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.ready, 0));
            sl.ready_pending = false;
        }
        rc = ensure(ctx, sl.desc, total + extra_room);
        if (rc != LFD_OK) return rc;
        if (sl.pinned_bytes < total + extra_room) {
            if (sl.pinned_in_flight) { LFD_HIP(ctx, hipEventSynchronize(sl.pinned_free)); sl.pinned_in_flight = false; }
            if (sl.pinned) LFD_HIP(ctx, hipHostFree(sl.pinned));
            sl.pinned = nullptr;
            sl.pinned_bytes = std::max<size_t>((total + extra_room) * 2, 1 << 16);
            LFD_HIP(ctx, hipHostMalloc(&sl.pinned, sl.pinned_bytes, hipHostMallocDefault));
        }
        if (sl.pinned_in_flight) { LFD_HIP(ctx, hipEventSynchronize(sl.pinned_free)); sl.pinned_in_flight = false; }
        std::memcpy(sl.pinned, blob.data(), total);
        LFD_HIP(ctx, hipMemcpyAsync(sl.desc.ptr, sl.pinned, total, hipMemcpyHostToDevice, st));
        LFD_HIP(ctx, hipEventRecord(sl.pinned_free, st));
        sl.pinned_in_flight = true;
        sl.cache.swap(blob);
        sl.consts_valid = false;
    }
    LfdBatchSlot& sl = ctx->slot[si];
    if (n_extra) {
        // this launch's selection offsets, behind the tables: a few dozen bytes, on the launch stream (the launch that reads them follows
        // in the same stream; an earlier launch that read the previous offsets precedes the copy in it)
        if (sl.desc.bytes < total + extra_room) return fail(ctx, LFD_ERR_STATE, "internal: batch slot without room for the selection offsets");
        if (sl.x_in_flight) { LFD_HIP(ctx, hipEventSynchronize(sl.x_free)); sl.x_in_flight = false; }      // (the previous indexed launch into this slot: long through)
        if (sl.pinned_x_bytes < extra_room) {
            if (sl.pinned_x) LFD_HIP(ctx, hipHostFree(sl.pinned_x));
            sl.pinned_x = nullptr;
            sl.pinned_x_bytes = std::max<size_t>(extra_room * 2, 4096);
            LFD_HIP(ctx, hipHostMalloc(&sl.pinned_x, sl.pinned_x_bytes, hipHostMallocDefault));
        }
        if (!sl.x_free) LFD_HIP(ctx, hipEventCreateWithFlags(&sl.x_free, hipEventDisableTiming));
        std::memcpy(sl.pinned_x, extra, n_extra * sizeof(long long));
        LFD_HIP(ctx, hipMemcpyAsync(static_cast<unsigned char*>(sl.desc.ptr) + off_extra, sl.pinned_x, n_extra * sizeof(long long), hipMemcpyHostToDevice,
                                    ctx->stream));
        LFD_HIP(ctx, hipEventRecord(sl.x_free, ctx->stream));
        sl.x_in_flight = true;
    }
    *slot_out = si;
    *stream_out = st;
    *d_refs = reinterpret_cast<const LfdRefDesc*>(sl.desc.ptr);
    *d_slots = reinterpret_cast<const LfdSlotDesc*>(static_cast<unsigned char*>(sl.desc.ptr) + off_slots);
    if (d_extra) *d_extra = reinterpret_cast<const long long*>(static_cast<unsigned char*>(sl.desc.ptr) + off_extra);
    *d_fund = b->fundamental ? reinterpret_cast<const float*>(static_cast<unsigned char*>(sl.desc.ptr) + off_fund) : nullptr;
    return LFD_OK;
}

int default_axes(lfd_context* ctx, int W, int H, const float** ax, const float** ay) {
    if (ctx->axes_w != W || ctx->axes_h != H || !ctx->axes.ptr) {
        int rc = ensure(ctx, ctx->axes, sizeof(float) * (size_t)(W + H));
        if (rc != LFD_OK) return rc;
        std::vector<float> host((size_t)W + H);
        lfd_identity_axis(W, host.data());
        lfd_identity_axis(H, host.data() + W);
        LFD_HIP(ctx, hipMemcpyAsync(ctx->axes.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->axes_w = W; ctx->axes_h = H;
    }
    *ax = static_cast<const float*>(ctx->axes.ptr);
    *ay = *ax + W;
    return LFD_OK;
}

// dense mode's colour tables for the analytic A-grid (lfd_geometry.hpp): built on the host with the per-cell arithmetic itself,
// once per (grid, match size)
int colour_tables(lfd_context* ctx, int W, int H, int wm, int hm, const LfdColourCol** cols, const LfdColourRow** rows) {
    const int key[4] = {W, H, wm, hm};
    if (!ctx->colour_tab.ptr || std::memcmp(key, ctx->colour_key, sizeof(key)) != 0) {
        static_assert(sizeof(LfdColourCol) == 16 && sizeof(LfdColourRow) == 16, "colour table entries are 16 bytes");
        int rc = ensure(ctx, ctx->colour_tab, sizeof(LfdColourCol) * (size_t)W + sizeof(LfdColourRow) * (size_t)H);
        if (rc != LFD_OK) return rc;
        std::vector<unsigned char> host(sizeof(LfdColourCol) * (size_t)W + sizeof(LfdColourRow) * (size_t)H);
        LfdColourCol* hc = reinterpret_cast<LfdColourCol*>(host.data());
        LfdColourRow* hr = reinterpret_cast<LfdColourRow*>(host.data() + sizeof(LfdColourCol) * (size_t)W);
        const LfdAxis ax = lfd_make_axis(W), ay = lfd_make_axis(H);
        for (int x = 0; x < W; ++x) hc[x] = lfd_colour_col(lfd_match_px(lfd_axis_value(ax, x), (float)(wm - 1)), wm);
        for (int y = 0; y < H; ++y) hr[y] = lfd_colour_row(lfd_match_px(lfd_axis_value(ay, y), (float)(hm - 1)), wm, hm);
        LFD_HIP(ctx, hipMemcpyAsync(ctx->colour_tab.ptr, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
        LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(ctx->colour_key, key, sizeof(key));
    }
    *cols = static_cast<const LfdColourCol*>(ctx->colour_tab.ptr);
    *rows = reinterpret_cast<const LfdColourRow*>(static_cast<const unsigned char*>(ctx->colour_tab.ptr) + sizeof(LfdColourCol) * (size_t)W);
    return LFD_OK;
}

}  // namespace

void lfd_fill_kernel_params(const lfd_batch* b, const lfd_params* p, LfdKernelParams& kp) {
    kp.sampson_thresh = p->sampson_thresh;
    kp.certainty_thresh = p->certainty_thresh;
    kp.reproj_thresh = p->reproj_thresh;
    kp.no_filter = p->no_filter ? 1 : 0;
    (void)LFD_FLAG_EXACT_COLOUR;     // consumed by lfd_triangulate_dense (kernel choice), not by the per-cell routine
    kp.use_sampson = (!p->no_filter && p->sampson_thresh > 0.0) ? 1 : 0;
    kp.use_parallax = (!p->no_filter && p->min_parallax_deg > 0.0f) ? 1 : 0;
    kp.dot_thresh = kp.use_parallax ? lfd_parallax_dot_threshold(p->min_parallax_deg) : 2.0f;
    kp.wm1 = (float)(b->w_match - 1);
    kp.hm1 = (float)(b->h_match - 1);
}

namespace {

int prepare_launch(lfd_context* ctx, const lfd_batch* b, const lfd_params* p, const long long* extra, size_t n_extra,
                   LfdLaunch& L, const long long** d_extra, bool ahead = false) {
    int rc = validate_batch(ctx, b, p);
    if (rc != LFD_OK) return rc;
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    std::memset(&L, 0, sizeof(L));
    if (!ahead && ctx->last_used >= 0) {
        // the kernels of the previous launch end here in the launch stream: from this point on its slot may be refilled ahead
        LfdBatchSlot& prev = ctx->slot[ctx->last_used];
        rc = slot_events(ctx, prev);
        if (rc != LFD_OK) return rc;
        if (!prev.idle_attached) LFD_HIP(ctx, hipEventRecord(prev.idle, ctx->stream));     // (a dense launch carried its own marker)
        prev.used = true;
    }
    int si = 0;
    hipStream_t st = ctx->stream;
    rc = upload_tables(ctx, b, extra, n_extra, ahead, &si, &st, &L.refs, &L.slots, d_extra, &L.fund_override);
    if (rc != LFD_OK) return rc;
    LfdBatchSlot& sl = ctx->slot[si];
    L.cams = static_cast<const LfdCam*>(ctx->cams.ptr);
    if (b->axis_x) { L.axis_x = b->axis_x; L.axis_y = b->axis_y; }
    else {
        rc = default_axes(ctx, b->W, b->H, &L.axis_x, &L.axis_y);
        if (rc != LFD_OK) return rc;
        L.axis_identity = 1;
    }
    L.ax = lfd_make_axis(b->W);
    L.ay = lfd_make_axis(b->H);
    L.n_refs = b->n_refs; L.k = b->k; L.H = b->H; L.W = b->W;
    L.w_match = b->w_match; L.h_match = b->h_match; L.warp_channels = b->warp_channels;
    const long long HW = (long long)b->H * b->W;
    const int tile = LFD_DENSE_BLOCK * LFD_DENSE_CPT;
    L.tiles_per_ref = (int)((HW + tile - 1) / tile);
    L.mask_sx = (float)b->w_match / (float)b->W;
    L.mask_sy = (float)b->h_match / (float)b->H;
    L.inv_w = 1.0f / (float)b->W;
    L.w_log2 = -1;
    if ((b->W & (b->W - 1)) == 0) { int l = 0; while ((1 << l) < b->W) ++l; L.w_log2 = l; }
    lfd_fill_kernel_params(b, p, L.kp);
    // per-pair constants: (re)derive when the batch tables, the cameras or the match size changed
    const size_t ref_bytes = ((size_t)b->n_refs * sizeof(LfdRefConst) + 15) & ~size_t(15);
    const size_t need = ref_bytes + (size_t)b->n_refs * b->k * sizeof(LfdPairConst);
    if (sl.consts.bytes < need) sl.consts_valid = false;
    rc = ensure(ctx, sl.consts, need);
    if (rc != LFD_OK) return rc;
    LfdRefConst* d_rc = static_cast<LfdRefConst*>(sl.consts.ptr);
    LfdPairConst* d_pc = reinterpret_cast<LfdPairConst*>(static_cast<unsigned char*>(sl.consts.ptr) + ref_bytes);
    L.ref_const = d_rc;
    L.pair_const = d_pc;
    if (!sl.consts_valid || sl.wm != b->w_match || sl.hm != b->h_match) {
        const int n = b->n_refs * b->k + b->n_refs;
        hipLaunchKernelGGL(lfd_pair_setup_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, L, d_rc, d_pc);
        LFD_HIP(ctx, hipGetLastError());
        sl.consts_valid = true;
        sl.refs = b->n_refs; sl.k = b->k;
        sl.wm = b->w_match;
        sl.hm = b->h_match;
        if (st != ctx->stream) {                 // issued ahead: the launch that uses the slot waits for this
            LFD_HIP(ctx, hipEventRecord(sl.ready, st));
            sl.ready_pending = true;
        }
    }
    if (!ahead) {
        ctx->cur = si; ctx->last_used = si;
        sl.idle_attached = false;          // this launch becomes the slot's last user: its end is marked by the launch itself or by the next one
    }
    return LFD_OK;
}

// ticket counter + tile states.  Tickets are never reset: each launch is given the value the
// counter holds when it starts; tile-state words carry a launch epoch, so stale words of earlier
// launches read as "empty" and no per-launch memset is needed.
// workspace: [0] u64 ticket counter, [8] u32 launch status, [16..] tile states
// workspace: [0] u64 ticket counter (indexed kernel), [8] u32 launch status, [16] u32 seg_ready, [128 + 128*s] u64 ticket counter of
// sequence s (dense kernel), [kWsHeader..] tile states
constexpr size_t kWsHeader = 128 + 128 * LFD_TICKET_LANES;

int prepare_lookback(lfd_context* ctx, size_t n_tiles, size_t n_tickets, bool lanes, LfdLaunch& L) {
    const size_t need = kWsHeader + n_tiles * sizeof(unsigned long long);
    const bool fresh = ctx->ws.bytes < need || !ctx->ws.ptr;
    int rc = ensure(ctx, ctx->ws, need, true);
    if (rc != LFD_OK) return rc;
    if (fresh) {
        ctx->tickets_issued = 0; ctx->epoch = 0;
        for (int s = 0; s < LFD_TICKET_LANES; ++s) ctx->lane_issued[s] = 0;
    }
    ctx->epoch = (ctx->epoch + 1) & LFD_EPOCH_MASK;
    if (ctx->epoch == 0) {   // epoch wrapped: clear stale words once
        LFD_HIP(ctx, hipMemsetAsync(static_cast<unsigned char*>(ctx->ws.ptr) + kWsHeader, 0, ctx->ws.bytes - kWsHeader, ctx->stream));
        LFD_HIP(ctx, hipMemsetAsync(static_cast<unsigned char*>(ctx->ws.ptr) + 16, 0, 4, ctx->stream));   // seg_ready
        ctx->epoch = 1;
    }
    unsigned char* base = static_cast<unsigned char*>(ctx->ws.ptr);
    L.ticket = reinterpret_cast<unsigned long long*>(base);
    L.ticket_lanes = reinterpret_cast<unsigned long long*>(base + 128);
    L.tile_state = reinterpret_cast<unsigned long long*>(base + kWsHeader);
    L.status = reinterpret_cast<unsigned int*>(base + 8);
    L.seg_ready = reinterpret_cast<unsigned int*>(base + 16);
    L.ticket_base = ctx->tickets_issued;
    L.epoch = ctx->epoch;
    if (lanes) {             // n_tickets workgroups, workgroup b draws from sequence b % LANES
        for (int s = 0; s < LFD_TICKET_LANES; ++s) {
            L.ticket_base_lane[s] = ctx->lane_issued[s];
            if ((size_t)s < n_tickets) ctx->lane_issued[s] += (n_tickets - (size_t)s + LFD_TICKET_LANES - 1) / LFD_TICKET_LANES;
        }
    } else {
        ctx->tickets_issued += n_tickets;
    }
    return LFD_OK;
}

void read_env_switches(lfd_context* ctx) {
    ctx->env = LfdEnvSwitches();
    if (const char* e = std::getenv("LFD_DENSE_EXTRA_LDS")) ctx->env.dense_extra_lds = (size_t)std::atol(e);
    if (const char* e = std::getenv("LFD_DENSE_TIMING")) ctx->env.dense_timing_path = e;
    if (const char* e = std::getenv("LFD_INDEXED_SPLIT")) ctx->env.indexed_split = std::atoi(e) != 0;
    if (const char* e = std::getenv("LFD_SELECT_TIMING")) ctx->env.select_timing = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("LFD_SELECT_WORKGROUPS")) ctx->env.select_workgroups = std::atoi(e);
}

int check_points(lfd_context* ctx, const lfd_points* out, const long long* ref_offsets) {
    if (!out || !out->xyz || !out->rgb || !out->err) return fail(ctx, LFD_ERR_INVALID, "null output buffers");
    if (out->capacity < 0) return fail(ctx, LFD_ERR_INVALID, "negative capacity");
    if (!ref_offsets) return fail(ctx, LFD_ERR_INVALID, "ref_offsets is required");
    return LFD_OK;
}

}  // namespace

int lfd_fail(lfd_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> lock(g_create_mutex); g_create_error = msg; }
    return code;
}

// =================================================================================================
extern "C" {

int lfd_abi_version(void) { return LFD_ABI_VERSION; }

int lfd_struct_layout(int32_t* out, int32_t capacity) {
#define LFD_OFF(T, f) (int32_t)offsetof(T, f), (int32_t)sizeof(((T*)nullptr)->f)
    const int32_t table[] = {
        (int32_t)sizeof(lfd_params), 7, LFD_OFF(lfd_params, sampson_thresh), LFD_OFF(lfd_params, certainty_thresh), LFD_OFF(lfd_params, sample_cap),
        LFD_OFF(lfd_params, reproj_thresh), LFD_OFF(lfd_params, min_parallax_deg), LFD_OFF(lfd_params, no_filter), LFD_OFF(lfd_params, flags),
        (int32_t)sizeof(lfd_batch), 19, LFD_OFF(lfd_batch, n_refs), LFD_OFF(lfd_batch, k), LFD_OFF(lfd_batch, H), LFD_OFF(lfd_batch, W),
        LFD_OFF(lfd_batch, w_match), LFD_OFF(lfd_batch, h_match), LFD_OFF(lfd_batch, warp_channels), LFD_OFF(lfd_batch, reserved),
        LFD_OFF(lfd_batch, ref_cam), LFD_OFF(lfd_batch, n_slots), LFD_OFF(lfd_batch, nbr_cam), LFD_OFF(lfd_batch, cert), LFD_OFF(lfd_batch, warp),
        LFD_OFF(lfd_batch, image), LFD_OFF(lfd_batch, mask_a), LFD_OFF(lfd_batch, mask_b), LFD_OFF(lfd_batch, axis_x), LFD_OFF(lfd_batch, axis_y),
        LFD_OFF(lfd_batch, fundamental),
        (int32_t)sizeof(lfd_points), 6, LFD_OFF(lfd_points, xyz), LFD_OFF(lfd_points, rgb), LFD_OFF(lfd_points, err), LFD_OFF(lfd_points, cell),
        LFD_OFF(lfd_points, slot), LFD_OFF(lfd_points, capacity),
        (int32_t)sizeof(lfd_tile_segment), 2, LFD_OFF(lfd_tile_segment, offset), LFD_OFF(lfd_tile_segment, count),
        (int32_t)sizeof(lfd_copy_segment), 3, LFD_OFF(lfd_copy_segment, src_offset), LFD_OFF(lfd_copy_segment, dst_offset), LFD_OFF(lfd_copy_segment, nbytes),
    };
#undef LFD_OFF
    const int32_t n = (int32_t)(sizeof(table) / sizeof(table[0]));
    for (int32_t i = 0; out != nullptr && i < n && i < capacity; ++i) out[i] = table[i];
    return n;
}

const char* lfd_struct_fields(void) {
    return "lfd_params:sampson_thresh,certainty_thresh,sample_cap,reproj_thresh,min_parallax_deg,no_filter,flags;"
           "lfd_batch:n_refs,k,H,W,w_match,h_match,warp_channels,reserved,ref_cam,n_slots,nbr_cam,cert,warp,image,mask_a,mask_b,axis_x,axis_y,fundamental;"
           "lfd_points:xyz,rgb,err,cell,slot,capacity;"
           "lfd_tile_segment:offset,count;"
           "lfd_copy_segment:src_offset,dst_offset,nbytes";
}

const char* lfd_last_error(const lfd_context* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lock(g_create_mutex);
    t_create_error_copy = g_create_error;
    return t_create_error_copy.c_str();
}

int lfd_create(int device_index, void* hip_stream, lfd_context** out) {
    if (!out) return fail(nullptr, LFD_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, LFD_ERR_HIP, std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device_index < 0 || device_index >= n) return fail(nullptr, LFD_ERR_INVALID, "device index out of range");
    lfd_context* ctx = new lfd_context();
    ctx->device = device_index;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    read_env_switches(ctx);      // profiling / A-B switches: read ONCE here, never on a launch path
    e = hipSetDevice(device_index);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, device_index);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned_words), 16 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        std::string m = std::string("context init: ") + hipGetErrorString(e);
        delete ctx;
        return fail(nullptr, LFD_ERR_HIP, m);
    }
    *out = ctx;
    return LFD_OK;
}

void lfd_destroy(lfd_context* ctx) {
    if (!ctx) return;
    if (ctx->is_host) { lfd_host_pool_destroy(ctx->host_pool); delete ctx; return; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->prep_stream) { (void)hipStreamSynchronize(ctx->prep_stream); (void)hipStreamDestroy(ctx->prep_stream); }
    for (LfdBatchSlot& sl : ctx->slot) {
        if (sl.desc.ptr) (void)hipFree(sl.desc.ptr);
        if (sl.consts.ptr) (void)hipFree(sl.consts.ptr);
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        if (sl.pinned_x) (void)hipHostFree(sl.pinned_x);
        for (hipEvent_t ev : {sl.pinned_free, sl.ready, sl.idle, sl.x_free}) if (ev) (void)hipEventDestroy(ev);
    }
    for (hipEvent_t ev : ctx->kt_start) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->kt_stop) (void)hipEventDestroy(ev);
    for (DeviceBuffer* b : {&ctx->cams, &ctx->ws, &ctx->axes, &ctx->scratch, &ctx->codes, &ctx->idx_tab, &ctx->agg, &ctx->sel_buf, &ctx->colour_tab, &ctx->mt, &ctx->mt_ckpt, &ctx->mt_batch, &ctx->sel_scratch, &ctx->sel_chain, &ctx->stamps, &ctx->img_tab, &ctx->msk_tab, &ctx->seg_scan})
        if (b->ptr) (void)hipFree(b->ptr);
    if (ctx->pinned_words) (void)hipHostFree(ctx->pinned_words);
    delete ctx;
}

int lfd_kernel_timing(lfd_context* ctx, int32_t n_launches) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (n_launches < 0 || n_launches > (1 << 20)) return fail(ctx, LFD_ERR_INVALID, "lfd_kernel_timing: n_launches must be in [0, 2^20]");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));           // no launch is still using the events that go
    if (ctx->prep_stream) LFD_HIP(ctx, hipStreamSynchronize(ctx->prep_stream));
    for (LfdBatchSlot& sl : ctx->slot) { sl.idle_attached = false; sl.idle_ext = nullptr; sl.used = false; }      // (nor is a slot waiting on one)
    while (ctx->kt_start.size() > (size_t)n_launches) {
        (void)hipEventDestroy(ctx->kt_start.back()); (void)hipEventDestroy(ctx->kt_stop.back());
        ctx->kt_start.pop_back(); ctx->kt_stop.pop_back();
    }
    while (ctx->kt_start.size() < (size_t)n_launches) {
        hipEvent_t a = nullptr, b = nullptr;
        LFD_HIP(ctx, hipEventCreate(&a));
        hipError_t e = hipEventCreate(&b);
        if (e != hipSuccess) { (void)hipEventDestroy(a); return fail(ctx, LFD_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
        ctx->kt_start.push_back(a); ctx->kt_stop.push_back(b);
    }
    ctx->kt_used = 0;
    return LFD_OK;
}

int lfd_kernel_timing_read(lfd_context* ctx, float* ms, int32_t capacity, int32_t* n_out) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (!n_out || capacity < 0 || (capacity > 0 && !ms)) return fail(ctx, LFD_ERR_INVALID, "lfd_kernel_timing_read: bad arguments");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = std::min(ctx->kt_used, (size_t)capacity);
    if (ctx->kt_used) LFD_HIP(ctx, hipEventSynchronize(ctx->kt_stop[ctx->kt_used - 1]));
    for (size_t i = 0; i < n; ++i) LFD_HIP(ctx, hipEventElapsedTime(ms + i, ctx->kt_start[i], ctx->kt_stop[i]));
    *n_out = (int32_t)n;
    ctx->kt_used = 0;
    return LFD_OK;
}

int lfd_reload_env(lfd_context* ctx) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    read_env_switches(ctx);
    return LFD_OK;
}

int lfd_set_stream(lfd_context* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->prep_stream) LFD_HIP(ctx, hipStreamSynchronize(ctx->prep_stream));
    for (LfdBatchSlot& sl : ctx->slot) { sl.ready_pending = false; sl.used = false; sl.idle_attached = false; }      // everything issued so far has completed
    ctx->last_used = -1;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return LFD_OK;
}

int lfd_launch_status(lfd_context* ctx, int32_t* status_out) {
    if (!ctx || !status_out) return fail(ctx, LFD_ERR_INVALID, "null argument");
    *status_out = 0;
    if (ctx->is_host) return LFD_OK;
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->ws.ptr) { LFD_HIP(ctx, hipStreamSynchronize(ctx->stream)); return LFD_OK; }
    LFD_HIP(ctx, hipMemcpyAsync(ctx->pinned_words, static_cast<unsigned char*>(ctx->ws.ptr) + 8, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned int st = (unsigned int)ctx->pinned_words[0];
    if (st != 0) {
        LFD_HIP(ctx, hipMemset(static_cast<unsigned char*>(ctx->ws.ptr) + 8, 0, sizeof(st)));
        *status_out = (int32_t)st;
        return fail(ctx, LFD_ERR_HIP, "a look-back spin timed out (workgroups of the persistent grid were not co-resident); results of the last launch are invalid");
    }
    return LFD_OK;
}

int lfd_get_pair_fundamental(lfd_context* ctx, int32_t n_pairs, double* F_out_host) {
    if (!ctx || !F_out_host || n_pairs <= 0) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    const LfdBatchSlot& sl = ctx->slot[ctx->cur];        // the batch of the last launch
    if (ctx->last_used < 0 || !sl.consts_valid || !sl.consts.ptr) return fail(ctx, LFD_ERR_STATE, "no batch has been launched on this context yet");
    if (n_pairs != sl.refs * sl.k) return fail(ctx, LFD_ERR_INVALID, "n_pairs must be n_refs * k of the last batch");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ref_bytes = ((size_t)sl.refs * sizeof(LfdRefConst) + 15) & ~size_t(15);
    std::vector<LfdPairConst> host((size_t)n_pairs);
    LFD_HIP(ctx, hipMemcpyAsync(host.data(), static_cast<unsigned char*>(sl.consts.ptr) + ref_bytes, host.size() * sizeof(LfdPairConst),
                                hipMemcpyDeviceToHost, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_pairs; ++i)
        for (int j = 0; j < 9; ++j) F_out_host[(size_t)i * 9 + j] = host[(size_t)i].F[j];
    return LFD_OK;
}

int lfd_upload_cameras(lfd_context* ctx, int32_t n, const float* K, const float* R, const float* t, const float* P,
                       const float* C, const int32_t* wh) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (n <= 0 || !K || !R || !t || !P || !C || !wh) return fail(ctx, LFD_ERR_INVALID, "bad camera arrays");
    std::vector<LfdCam> cams((size_t)n);
    for (int i = 0; i < n; ++i) {
        LfdCam& c = cams[i];
        std::memcpy(c.K, K + (size_t)i * 9, sizeof(c.K));
        std::memcpy(c.R, R + (size_t)i * 9, sizeof(c.R));
        std::memcpy(c.t, t + (size_t)i * 3, sizeof(c.t));
        std::memcpy(c.P, P + (size_t)i * 12, sizeof(c.P));
        std::memcpy(c.C, C + (size_t)i * 3, sizeof(c.C));
        c.w = wh[i * 2 + 0]; c.h = wh[i * 2 + 1];
        c.pad[0] = c.pad[1] = 0;
        if (c.w <= 0 || c.h <= 0) return fail(ctx, LFD_ERR_INVALID, "camera width/height must be positive");
    }
    if (ctx->is_host) { ctx->host_cams.swap(cams); ctx->n_cams = n; return LFD_OK; }
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->cams, cams.size() * sizeof(LfdCam));
    if (rc != LFD_OK) return rc;
    LFD_HIP(ctx, hipMemcpyAsync(ctx->cams.ptr, cams.data(), cams.size() * sizeof(LfdCam), hipMemcpyHostToDevice, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->prep_stream) LFD_HIP(ctx, hipStreamSynchronize(ctx->prep_stream));
    ctx->n_cams = n;
    for (LfdBatchSlot& sl : ctx->slot) sl.consts_valid = false;      // every constant block derives from the camera table
    return LFD_OK;
}

int lfd_prepare_batch(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params) {
    LfdLaunch L;
    return prepare_launch(ctx, batch, params, nullptr, 0, L, nullptr, /*ahead=*/true);
}

int lfd_aggregate(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, float* best_cert, uint8_t* best_slot) {
    LfdLaunch L;
    int rc = prepare_launch(ctx, batch, params, nullptr, 0, L, nullptr);
    if (rc != LFD_OK) return rc;
    if (!best_cert) return fail(ctx, LFD_ERR_INVALID, "best_cert is null");
    const long long HW = (long long)batch->H * batch->W;
    const int per_block = 256 * 4;
    int gx = (int)std::min<long long>((HW + per_block - 1) / per_block, 2048);
    dim3 grid((unsigned)gx, (unsigned)batch->n_refs, 1);
    hipLaunchKernelGGL(lfd_aggregate_kernel, grid, dim3(256), 0, ctx->stream, L, best_cert, best_slot);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

// the dense launch in its two forms: ordered (ref_offsets; ref_counts / table null) and unordered retirement (ref_counts + table)
static int dense_launch(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, const lfd_points* out,
                        int64_t* ref_offsets, int32_t* seg_counts, int64_t* ref_counts, lfd_tile_segment* table, uint8_t* ply = nullptr) {
    const bool unordered = table != nullptr;
    LfdLaunch L;
    int rc = prepare_launch(ctx, batch, params, nullptr, 0, L, nullptr);
    if (rc != LFD_OK) return rc;
    if (ply) {                         // file-payload output: the records go to `ply`; out carries the capacity (and optionally cell / slot)
        if (!out || out->capacity < 0 || !(unordered ? (void*)ref_counts : (void*)ref_offsets))
            return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_dense_ply[_segments]: the capacity and ref_offsets (ordered) / ref_counts (unordered) are required");
    } else {
        rc = check_points(ctx, out, reinterpret_cast<long long*>(unordered ? ref_counts : ref_offsets));
        if (rc != LFD_OK) return rc;
    }
    if (unordered && out->capacity < (int64_t)batch->n_refs * batch->H * batch->W)
        return fail(ctx, LFD_ERR_CAPACITY, "lfd_triangulate_dense_segments: capacity must be n_refs * H * W (every reference owns a region of H * W records)");
    const size_t n_tiles = (size_t)batch->n_refs * (size_t)L.tiles_per_ref;
    if (n_tiles > 0x7fffffffu) return fail(ctx, LFD_ERR_INVALID, "too many tiles in one launch");
    const size_t grid = n_tiles;              // one workgroup per tile, numbered by ticket
    rc = prepare_lookback(ctx, n_tiles, grid, true, L);
    if (rc != LFD_OK) return rc;
    L.xyz = out->xyz; L.rgb = out->rgb; L.err = out->err; L.cell = out->cell; L.slot = out->slot;
    L.capacity = out->capacity;
    L.ref_offsets = reinterpret_cast<long long*>(ref_offsets);
    L.seg_counts = seg_counts;                // zeroed inside the kernel (tile 0), no memset launch
    static_assert(sizeof(lfd_tile_segment) == sizeof(LfdTileSeg), "lfd_tile_segment is the kernels' LfdTileSeg");
    L.ref_cursor = reinterpret_cast<unsigned long long*>(ref_counts);      // (zeroed the same way)
    L.tile_table = reinterpret_cast<LfdTileSeg*>(table);
    L.ply = ply;
    size_t extra_lds = 0;                     // profiling switch: dynamic LDS lowers the number of resident workgroups
    extra_lds = ctx->env.dense_extra_lds;
#if defined(LFD_DENSE_TIMING)            // profiling builds: per-tile phase stamps, dumped to the file named by LFD_DENSE_TIMING
    const char* stamp_path = ctx->env.dense_timing_path.empty() ? nullptr : ctx->env.dense_timing_path.c_str();
    if (stamp_path) {
        rc = ensure(ctx, ctx->stamps, n_tiles * 2 * 12 * sizeof(unsigned long long));
        if (rc != LFD_OK) return rc;
        LFD_HIP(ctx, hipMemsetAsync(ctx->stamps.ptr, 0, n_tiles * 2 * 12 * sizeof(unsigned long long), ctx->stream));
        L.phase_stamps = static_cast<unsigned long long*>(ctx->stamps.ptr);
    }
#endif
#if defined(LFD_FORCE_EXACT_COLOUR)      // profiling builds: what the f64 colour costs on the bench workload
    const bool exact_colour = true;
#else
    const bool exact_colour = (params->flags & LFD_FLAG_EXACT_COLOUR) != 0;
#endif
    if (!exact_colour && L.axis_identity && batch->warp_channels == 2) {      // the f32 blend reads its positions and weights from tables
        rc = colour_tables(ctx, batch->W, batch->H, batch->w_match, batch->h_match, &L.colour_cols, &L.colour_rows);
        if (rc != LFD_OK) return rc;
    }
    // The launch carries its own stop event (hipExtLaunchKernelGGL): it marks the end of this batch slot's last user - what the staging
    // of a later batch into the slot waits for - without a marker packet of its own in the stream; with lfd_kernel_timing switched on
    // the pair of events is the next of its ring and doubles as that marker.
    LfdBatchSlot& used_slot = ctx->slot[ctx->cur];
    rc = slot_events(ctx, used_slot);
    if (rc != LFD_OK) return rc;
    hipEvent_t t_start = nullptr, t_stop = used_slot.idle;
    if (ctx->kt_used < ctx->kt_start.size()) { t_start = ctx->kt_start[ctx->kt_used]; t_stop = ctx->kt_stop[ctx->kt_used]; ++ctx->kt_used; }
    auto kernel = (ply && unordered) ? (exact_colour ? lfd_dense_ply_segments_exact_kernel : lfd_dense_ply_segments_kernel)
                : ply ? (exact_colour ? lfd_dense_ply_exact_kernel : lfd_dense_ply_kernel)
                : unordered ? (exact_colour ? lfd_dense_segments_exact_kernel : lfd_dense_segments_kernel)
                            : (exact_colour ? lfd_dense_exact_kernel : lfd_dense_kernel);
    hipExtLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(LFD_DENSE_BLOCK), (unsigned)extra_lds, ctx->stream, t_start, t_stop, 0, L);
    LFD_HIP(ctx, hipGetLastError());
    used_slot.idle_ext = t_stop;          // (only behind a launch that went out: a later staging into the slot waits for THIS event)
    used_slot.idle_attached = true;
#if defined(LFD_DENSE_TIMING)
    if (stamp_path) {
        std::vector<unsigned long long> host(n_tiles * 2 * 12);
        LFD_HIP(ctx, hipMemcpyAsync(host.data(), ctx->stamps.ptr, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (FILE* f = std::fopen(stamp_path, "wb")) { std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f); std::fclose(f); }
    }
#endif
    return LFD_OK;
}

int lfd_triangulate_dense(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, const lfd_points* out,
                          int64_t* ref_offsets, int32_t* seg_counts) {
    return dense_launch(ctx, batch, params, out, ref_offsets, seg_counts, nullptr, nullptr);
}

int lfd_triangulate_dense_ply(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, uint8_t* records, int64_t capacity,
                              int64_t* ref_offsets, int32_t* seg_counts, int32_t* cell, uint8_t* slot) {
    if (ctx && !records) return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_dense_ply: records is null");
    lfd_points out;
    std::memset(&out, 0, sizeof(out));
    out.capacity = capacity;
    out.cell = cell;
    out.slot = slot;
    return dense_launch(ctx, batch, params, &out, ref_offsets, seg_counts, nullptr, nullptr, records);
}

int lfd_dense_tiles_per_ref(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    const long long tile = LFD_DENSE_BLOCK * LFD_DENSE_CPT;
    return (int)(((long long)H * W + tile - 1) / tile);
}

int lfd_triangulate_dense_segments(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, const lfd_points* out,
                                   int64_t* ref_counts, int32_t* seg_counts, lfd_tile_segment* table) {
    if (ctx && !table) return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_dense_segments: the tile table is required");
    return dense_launch(ctx, batch, params, out, nullptr, seg_counts, ref_counts, table);
}

int lfd_triangulate_dense_ply_segments(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, uint8_t* records, int64_t capacity,
                                       int64_t* ref_counts, int32_t* seg_counts, lfd_tile_segment* table) {
    if (ctx && (!records || !table)) return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_dense_ply_segments: records and the tile table are required");
    lfd_points out;
    std::memset(&out, 0, sizeof(out));
    out.capacity = capacity;
    return dense_launch(ctx, batch, params, &out, nullptr, seg_counts, ref_counts, table, records);
}

// exclusive prefix of the table in tile order into the context's scratch (tile_dst[n_tiles + 1]); the launches that follow read it
static int segment_scan(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, int64_t* ref_offsets,
                        const long long** tile_dst, int* tpr_out, int* n_tiles_out) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (n_refs <= 0 || H <= 0 || W <= 0 || !table) return fail(ctx, LFD_ERR_INVALID, "bad segment arguments");
    const int tpr = lfd_dense_tiles_per_ref(H, W);
    const long long n_tiles = (long long)n_refs * tpr;
    if (n_tiles > 0x7fffffffLL) return fail(ctx, LFD_ERR_INVALID, "too many tiles");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->seg_scan, ((size_t)n_tiles + 1) * sizeof(long long) + ((size_t)n_tiles + 2) * sizeof(int));      // prefix + chunk index
    if (rc != LFD_OK) return rc;
    long long* dst = static_cast<long long*>(ctx->seg_scan.ptr);
    hipLaunchKernelGGL(lfd_segment_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, reinterpret_cast<const LfdTileSeg*>(table), (int)n_tiles, tpr,
                       (int)n_refs, dst, reinterpret_cast<long long*>(ref_offsets));
    LFD_HIP(ctx, hipGetLastError());
    *tile_dst = dst; *tpr_out = tpr; *n_tiles_out = (int)n_tiles;
    return LFD_OK;
}

int lfd_order_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const lfd_points* src,
                       const lfd_points* dst, int64_t* ref_offsets) {
    const long long* tile_dst = nullptr; int tpr = 0, n_tiles = 0;
    int rc = segment_scan(ctx, n_refs, H, W, table, ref_offsets, &tile_dst, &tpr, &n_tiles);
    if (rc != LFD_OK) return rc;
    if (!src || !dst || !src->xyz || !src->rgb || !src->err || !dst->xyz || !dst->rgb || !dst->err) return fail(ctx, LFD_ERR_INVALID, "null point buffers");
    if (dst->capacity <= 0) return LFD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((dst->capacity + 1023) / 1024, 8192);
    hipLaunchKernelGGL(lfd_order_segments_kernel, dim3(grid), dim3(256), 0, ctx->stream, tile_dst, reinterpret_cast<const LfdTileSeg*>(table),
                       (long long)H * W, tpr, n_tiles, src->xyz, src->rgb, src->err, src->cell, src->slot, dst->xyz, dst->rgb, dst->err, dst->cell,
                       dst->slot, (long long)dst->capacity);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_pack_ply_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const float* xyz,
                          const float* rgb, int64_t capacity, uint8_t* out, int64_t* ref_offsets) {
    const long long* tile_dst = nullptr; int tpr = 0, n_tiles = 0;
    int rc = segment_scan(ctx, n_refs, H, W, table, ref_offsets, &tile_dst, &tpr, &n_tiles);
    if (rc != LFD_OK) return rc;
    if (capacity < 0 || (capacity > 0 && (!xyz || !rgb || !out))) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (reinterpret_cast<uintptr_t>(out) & 3u) return fail(ctx, LFD_ERR_INVALID, "out must be 4-byte aligned");
    if (capacity == 0) return LFD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((capacity + 255) / 256, 4096);
    hipLaunchKernelGGL(lfd_pack_ply_segments_kernel, dim3(grid), dim3(256), 0, ctx->stream, tile_dst, reinterpret_cast<const LfdTileSeg*>(table),
                       (long long)H * W, tpr, n_tiles, xyz, rgb, (long long)capacity, out);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_pack_points3d_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const float* xyz,
                               const float* rgb, const float* err, int64_t capacity, uint64_t id_base, uint8_t* out, int64_t* ref_offsets) {
    const long long* tile_dst = nullptr; int tpr = 0, n_tiles = 0;
    int rc = segment_scan(ctx, n_refs, H, W, table, ref_offsets, &tile_dst, &tpr, &n_tiles);
    if (rc != LFD_OK) return rc;
    if (capacity < 0 || (capacity > 0 && (!xyz || !rgb || !err || !out))) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (reinterpret_cast<uintptr_t>(out) & 3u) return fail(ctx, LFD_ERR_INVALID, "out must be 4-byte aligned");
    if (capacity == 0) return LFD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((capacity + 255) / 256, 4096);
    hipLaunchKernelGGL(lfd_pack_points3d_segments_kernel, dim3(grid), dim3(256), 0, ctx->stream, tile_dst, reinterpret_cast<const LfdTileSeg*>(table),
                       (long long)H * W, tpr, n_tiles, xyz, rgb, err, (long long)capacity, (unsigned long long)id_base, out);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_triangulate_indexed(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, const int64_t* sel_idx,
                            const int64_t* sel_offsets, const lfd_points* out, int64_t* ref_offsets, int32_t* seg_counts,
                            int32_t* seg_order) {
    if (ctx && (!sel_idx || !sel_offsets)) return fail(ctx, LFD_ERR_INVALID, "sel_idx / sel_offsets are required");
    if (ctx && batch && batch->n_refs > 0) {
        if (sel_offsets[0] != 0) return fail(ctx, LFD_ERR_INVALID, "sel_offsets[0] must be 0");
        for (int r = 0; r < batch->n_refs; ++r)
            if (sel_offsets[r + 1] < sel_offsets[r]) return fail(ctx, LFD_ERR_INVALID, "sel_offsets must be non-decreasing");
    }
    LfdLaunch L;
    const long long* d_off = nullptr;
    int rc = prepare_launch(ctx, batch, params, reinterpret_cast<const long long*>(sel_offsets),
                            batch ? (size_t)std::max(batch->n_refs, 0) + 1 : 0, L, &d_off);
    if (rc != LFD_OK) return rc;
    rc = check_points(ctx, out, reinterpret_cast<long long*>(ref_offsets));
    if (rc != LFD_OK) return rc;
    const size_t n_sel = (size_t)sel_offsets[batch->n_refs];
    rc = prepare_lookback(ctx, (size_t)batch->n_refs, (size_t)batch->n_refs, false, L);
    if (rc != LFD_OK) return rc;
    rc = ensure(ctx, ctx->scratch, std::max<size_t>(n_sel, 1) * 8 * sizeof(float));
    if (rc != LFD_OK) return rc;
    rc = ensure(ctx, ctx->codes, std::max<size_t>(n_sel, 1));
    if (rc != LFD_OK) return rc;
    L.xyz = out->xyz; L.rgb = out->rgb; L.err = out->err; L.cell = out->cell; L.slot = out->slot;
    L.capacity = out->capacity;
    L.ref_offsets = reinterpret_cast<long long*>(ref_offsets);
    L.seg_counts = seg_counts;
    // pass A (the per-cell arithmetic) runs on the whole chip, one thread per selected cell; the per-reference workgroups
    // of lfd_indexed_kernel then only order and scatter.  LFD_INDEXED_SPLIT=0 keeps everything in the one kernel.
    bool split = true;
    split = ctx->env.indexed_split;
    long long max_sel = 0;
    for (int r = 0; r < batch->n_refs; ++r) max_sel = std::max<long long>(max_sel, sel_offsets[r + 1] - sel_offsets[r]);
    unsigned* tab = nullptr;
    if (split && max_sel > 0) {
        const size_t tab_bytes = (size_t)batch->n_refs * LFD_MAX_SLOTS * 2 * sizeof(unsigned);
        rc = ensure(ctx, ctx->idx_tab, tab_bytes);
        if (rc != LFD_OK) return rc;
        tab = static_cast<unsigned*>(ctx->idx_tab.ptr);
        LFD_HIP(ctx, hipMemsetAsync(tab, 0, tab_bytes, ctx->stream));
        const unsigned chunks = (unsigned)((max_sel + LFD_INDEXED_EVAL_BLOCK - 1) / LFD_INDEXED_EVAL_BLOCK);
        hipLaunchKernelGGL(lfd_indexed_eval_kernel, dim3(chunks, (unsigned)batch->n_refs), dim3(LFD_INDEXED_EVAL_BLOCK), 0, ctx->stream, L,
                           reinterpret_cast<const long long*>(sel_idx), d_off, static_cast<float*>(ctx->scratch.ptr),
                           static_cast<uint8_t*>(ctx->codes.ptr), tab, 0);
        LFD_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(lfd_indexed_kernel, dim3((unsigned)batch->n_refs), dim3(LFD_INDEXED_BLOCK), 0, ctx->stream, L,
                       reinterpret_cast<const long long*>(sel_idx), d_off, static_cast<float*>(ctx->scratch.ptr),
                       static_cast<uint8_t*>(ctx->codes.ptr), seg_order, static_cast<const unsigned*>(tab), 0);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

// ---- S: selection ---------------------------------------------------------------------------------
int lfd_rng_seed(lfd_context* ctx, uint32_t seed) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->mt, 625 * sizeof(unsigned));
    if (rc != LFD_OK) return rc;
    hipLaunchKernelGGL(lfd_mt_seed_kernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<unsigned*>(ctx->mt.ptr), seed);
    LFD_HIP(ctx, hipGetLastError());
    ctx->mt_seeded = true;
    return LFD_OK;
}

int lfd_rng_get_state(lfd_context* ctx, uint32_t* key624, int32_t* pos) {
    if (!ctx || !key624 || !pos) return fail(ctx, LFD_ERR_INVALID, "null argument");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (!ctx->mt_seeded) return fail(ctx, LFD_ERR_STATE, "lfd_rng_seed / lfd_rng_set_state must be called first");
    unsigned host[625];
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    LFD_HIP(ctx, hipMemcpyAsync(host, ctx->mt.ptr, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(key624, host, 624 * sizeof(unsigned));
    *pos = (int32_t)host[624];
    return LFD_OK;
}

int lfd_rng_set_state(lfd_context* ctx, const uint32_t* key624, int32_t pos) {
    if (!ctx || !key624 || pos < 0 || pos > 624) return fail(ctx, LFD_ERR_INVALID, "bad MT19937 state");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->mt, 625 * sizeof(unsigned));
    if (rc != LFD_OK) return rc;
    unsigned host[625];
    std::memcpy(host, key624, 624 * sizeof(unsigned));
    host[624] = (unsigned)pos;
    LFD_HIP(ctx, hipMemcpyAsync(ctx->mt.ptr, host, sizeof(host), hipMemcpyHostToDevice, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->mt_seeded = true;
    return LFD_OK;
}

// The stream put aside / taken back in stream order (device-to-device, 2.5 KB): see the header.
static int rng_checkpoint_copy(lfd_context* ctx, int32_t place, bool take_back) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (place < 0 || place >= LFD_RNG_CHECKPOINTS) return fail(ctx, LFD_ERR_INVALID, "checkpoint place out of range");
    if (!ctx->mt_seeded) return fail(ctx, LFD_ERR_STATE, "lfd_rng_seed / lfd_rng_set_state must be called first");
    if (take_back && !(ctx->mt_ckpt_taken & (1u << place))) return fail(ctx, LFD_ERR_STATE, "no checkpoint was taken at this place");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->mt_ckpt, (size_t)LFD_RNG_CHECKPOINTS * 640 * sizeof(unsigned));
    if (rc != LFD_OK) return rc;
    unsigned* slot = static_cast<unsigned*>(ctx->mt_ckpt.ptr) + (size_t)place * 640;
    if (take_back) LFD_HIP(ctx, hipMemcpyAsync(ctx->mt.ptr, slot, 625 * sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream));
    else LFD_HIP(ctx, hipMemcpyAsync(slot, ctx->mt.ptr, 625 * sizeof(unsigned), hipMemcpyDeviceToDevice, ctx->stream));
    if (!take_back) ctx->mt_ckpt_taken |= 1u << place;
    return LFD_OK;
}

int lfd_rng_checkpoint(lfd_context* ctx, int32_t place) { return rng_checkpoint_copy(ctx, place, false); }
int lfd_rng_rollback(lfd_context* ctx, int32_t place) { return rng_checkpoint_copy(ctx, place, true); }

// whether a filtered selection of an N-cell map runs on the multi-workgroup kernel (the only one that can chain references on one stream)
static bool select_runs_on_several_workgroups(lfd_context* ctx, long long N) {
    int n_wg = LFD_SELECT_DEFAULT_WG;
    if (ctx->env.select_workgroups >= 0) n_wg = ctx->env.select_workgroups;
    n_wg = std::min(std::min(n_wg, (int)LFD_SELECT_MAX_WG), (int)(N / 8192));
    return n_wg >= 2 && ctx->env.select_timing == 0;
}

// launches the selection of one reference; *d_info = device {n_out, status}; nothing is read back here.
// n_batch > 1 (or info_batch set): n_batch references in ONE launch (blockIdx.y = reference) - maps best_cert + r*H*W, cells
// sel_out + r*capacity, {begin, end} pairs sel_offsets_dev + 2r, MT19937 states mt_batch + r*LFD_MT_STATE_STRIDE, results at
// info_batch + 2r; every reference has its own scratch block and barrier words, so they run side by side (a selection
// occupies 17 of the 256 CUs).
static int select_launch(lfd_context* ctx, bool topm, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                         int32_t border, int32_t tiles, float s_override, int64_t* sel_out, int64_t capacity,
                         long long* sel_offsets_dev, int** d_info, unsigned char** d_time, int n_batch = 1,
                         unsigned* mt_batch = nullptr, int* info_batch = nullptr, const float* s_batch = nullptr) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (!best_cert || !sel_out) return fail(ctx, LFD_ERR_INVALID, "null argument");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL || M < 0 || tiles <= 0 || border < 0 || capacity < 0)
        return fail(ctx, LFD_ERR_INVALID, "bad selection arguments");
    if (!topm && !mt_batch && !ctx->mt_seeded) return fail(ctx, LFD_ERR_STATE, "lfd_rng_seed must be called before lfd_select_samples");
    if (n_batch < 1 || n_batch > LFD_SELECT_BATCH_MAX) return fail(ctx, LFD_ERR_INVALID, "bad selection batch");
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->mt.ptr) { int rc0 = ensure(ctx, ctx->mt, 625 * sizeof(unsigned)); if (rc0 != LFD_OK) return rc0; }
    const size_t N = (size_t)H * W, Mz = (size_t)std::max(M, 1);
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t o_w = 0, o_p = o_w + up(N * 4), o_cdf = o_p + up(N * 8), o_first = o_cdf + up(N * 8), o_mark = o_first + up(N * 4),
                 o_draws = o_mark + up(N), o_cand = o_draws + up(Mz * 8), o_found = o_cand + up(Mz * 4), o_out = o_found + up(Mz * 4),
                 o_time = o_out + 256, o_coop = o_time + 256, total = o_coop + up(LFD_SELECT_COOP_BYTES);
    int rc = ensure(ctx, ctx->sel_scratch, total * (size_t)n_batch);
    if (rc != LFD_OK) return rc;
    unsigned char* base = static_cast<unsigned char*>(ctx->sel_scratch.ptr);
    LfdSelectArgs A;
    std::memset(&A, 0, sizeof(A));
    A.best_cert = best_cert;
    A.weights = reinterpret_cast<float*>(base + o_w);
    A.p = reinterpret_cast<double*>(base + o_p);
    A.cdf = reinterpret_cast<double*>(base + o_cdf);
    A.first = reinterpret_cast<int*>(base + o_first);
    A.mark = base + o_mark;
    A.draws = reinterpret_cast<double*>(base + o_draws);
    A.cand = reinterpret_cast<int*>(base + o_cand);
    A.found = reinterpret_cast<int*>(base + o_found);
    A.mt = mt_batch ? mt_batch : static_cast<unsigned*>(ctx->mt.ptr);
    A.batch_scratch_stride = (long long)total;
    A.batch_cert_stride = (long long)N;
    A.batch_out_stride = (long long)capacity;
    A.batch_mt_stride = mt_batch ? LFD_MT_STATE_STRIDE : 0;
    A.batch_info = info_batch;
    A.sel_out = reinterpret_cast<long long*>(sel_out);
    A.n_out = reinterpret_cast<int*>(base + o_out);
    A.status = reinterpret_cast<int*>(base + o_out + 4);
    A.capacity = capacity;
    A.H = H; A.W = W; A.M = M; A.border = border; A.tiles = tiles; A.cap = cap; A.s_override = s_override;
    A.sel_offsets_out = sel_offsets_dev;
    LfdSelectNorms norms;
    std::memset(&norms, 0, sizeof(norms));
    if (s_batch) {
        norms.use = 1;
        for (int i = 0; i < n_batch; ++i) norms.s[i] = s_batch[i];
    }
    *d_info = reinterpret_cast<int*>(base + o_out);
    *d_time = nullptr;
    const bool timing = !topm && n_batch == 1 && ctx->env.select_timing != 0;
    if (timing) {
        A.timing = reinterpret_cast<unsigned long long*>(base + o_time);
        LFD_HIP(ctx, hipMemsetAsync(base + o_time, 0, 256, ctx->stream));
        *d_time = base + o_time;
    }
    if (topm) {
        const size_t lds = (size_t)LFD_SELECT_TOPM_MAX * sizeof(unsigned long long);
        if (!ctx->topm_lds_attr_set) {     // after hipSetDevice(ctx->device) above: the attribute belongs to this context's device
            LFD_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(lfd_select_topm_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->topm_lds_attr_set = true;
        }
        hipLaunchKernelGGL(lfd_select_topm_kernel, dim3(1, (unsigned)n_batch), dim3(LFD_SELECT_BLOCK), lds, ctx->stream, A);
    } else {
        // several workgroups (one per CU) when the map is large enough to share out; LFD_SELECT_WORKGROUPS=0 keeps the
        // single-workgroup kernel (both produce the same selection)
        int n_wg = LFD_SELECT_DEFAULT_WG;
        if (ctx->env.select_workgroups >= 0) n_wg = ctx->env.select_workgroups;
        n_wg = std::min(std::min(n_wg, (int)LFD_SELECT_MAX_WG), (int)(N / 8192));
        {
            const int tile = std::max(1, W / tiles);
            const long long nbins = (long long)((W - 1) / tile + 1) * ((H - 1) / tile + 1);
            if (nbins > LFD_SELECT_MAX_BINS)
                return fail(ctx, LFD_ERR_INVALID, "selection: too many coverage bins: a " + std::to_string(H) + " x " + std::to_string(W) + " grid with " +
                            std::to_string(tiles) + " tiles per side has " + std::to_string(nbins) + " coverage tiles, the device stage holds " +
                            std::to_string((int)LFD_SELECT_MAX_BINS) + " (every square grid fits; RoMa's, 320 ... 1280 cells per side, have 576 ... 625); the host selection "
                            "stage (selection_backend=\"host\" in the Python mirror: upstream's own library calls) has no such limit");
        }
        const bool timing_mw = timing && ctx->env.select_timing == 2;     // 2: stamps of workgroup 0 of the multi-workgroup kernel
        const bool several_wg = n_wg >= 2 && (!timing || timing_mw);
        const bool one_stream = mt_batch == nullptr;              // the references of the launch share the context's stream, in batch order
        if (one_stream && n_batch > 1 && !several_wg)
            return fail(ctx, LFD_ERR_STATE, "selection: several references on one stream need the multi-workgroup kernel");
        if (several_wg) {
            // the stream as an array (lfd_select.hip, mt_stream_producer): a ring of doubles four first rounds long (later rounds are shorter than
            // the first, the producer stays at most one first round ahead) and the key of every twist the producer can be ahead by - one set for
            // the launch when its references share a stream, one per reference when every reference has its own
            const long long first_round = std::max<long long>(std::min<long long>((long long)((double)M * 0.85), (long long)N), 1);
            long long ring_cap = 4096;
            while (ring_cap < 4 * first_round) ring_cap *= 2;
            const int snap_slots = (int)(2 * ring_cap / 624) + 8;
            const size_t o_ring = LFD_CHAIN_BYTES, o_snaps = o_ring + (size_t)ring_cap * 8,
                         chain_total = (o_snaps + (size_t)snap_slots * 624 * 4 + 255) & ~size_t(255);
            const int n_chains = one_stream ? 1 : n_batch;
            rc = ensure(ctx, ctx->sel_chain, chain_total * (size_t)n_chains);
            if (rc != LFD_OK) return rc;
            unsigned char* cb = static_cast<unsigned char*>(ctx->sel_chain.ptr);
            if (n_chains == 1) LFD_HIP(ctx, hipMemsetAsync(cb, 0, LFD_CHAIN_BYTES, ctx->stream));
            else LFD_HIP(ctx, hipMemset2DAsync(cb, chain_total, 0, LFD_CHAIN_BYTES, (size_t)n_chains, ctx->stream));
            A.chain = cb;
            A.ring = reinterpret_cast<double*>(cb + o_ring);
            A.snaps = reinterpret_cast<unsigned*>(cb + o_snaps);
            A.ring_cap = ring_cap;
            A.snap_slots = snap_slots;
            A.chain_refs = one_stream ? n_batch : 1;
            A.batch_chain_stride = one_stream ? 0 : (long long)chain_total;
        }
        if (several_wg) {
            A.coop = base + o_coop;
            A.n_wg = n_wg;
            // barrier words, flags and the coverage bins (everything in front of LFD_COOP_MT)
            if (n_batch == 1) LFD_HIP(ctx, hipMemsetAsync(base + o_coop, 0, LFD_COOP_MT, ctx->stream));
            else LFD_HIP(ctx, hipMemset2DAsync(base + o_coop, total, 0, LFD_COOP_MT, (size_t)n_batch, ctx->stream));
            hipLaunchKernelGGL(lfd_select_filter_mw_kernel, dim3((unsigned)n_wg + 1u, (unsigned)n_batch), dim3(LFD_SELECT_BLOCK), 0, ctx->stream, A, norms);
        } else {
            hipLaunchKernelGGL(lfd_select_filter_kernel, dim3(1, (unsigned)n_batch), dim3(LFD_SELECT_BLOCK), 0, ctx->stream, A);
        }
    }
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

static const char* select_status_name(int st) {
    static const char* names[] = {"ok", "probabilities contain NaN", "probabilities are not non-negative",
                                  "Fewer non-zero entries in p than size", "weight below 2^-29: exact parallel cumsum not guaranteed",
                                  "no progress", "too many coverage bins", "sel_out capacity too small"};
    return names[std::min(std::max(st, 0), 7)];
}

static int select_impl(lfd_context* ctx, bool topm, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                       int32_t border, int32_t tiles, float s_override, int64_t* sel_out, int64_t capacity,
                       int32_t* n_sel_host, int32_t* status_host) {
    if (ctx && (!n_sel_host || !status_host)) return fail(ctx, LFD_ERR_INVALID, "null argument");
    int* d_info = nullptr;
    unsigned char* d_time = nullptr;
    int rc = select_launch(ctx, topm, best_cert, H, W, M, cap, border, tiles, s_override, sel_out, capacity, nullptr, &d_info, &d_time);
    if (rc != LFD_OK) return rc;
    int* host = ctx->pinned_words + 4;
    LFD_HIP(ctx, hipMemcpyAsync(host, d_info, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    LFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_sel_host = host[0];
    *status_host = host[1];
    if (d_time) {   // phase times of the filter kernel (100 MHz wall clock), profiling only
        unsigned long long t[32];
        LFD_HIP(ctx, hipMemcpy(t, d_time, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "[lfd] select phases (us):");
        for (int i = 1; i < 30 && t[i]; ++i) fprintf(stderr, " %.1f", (double)(t[i] - t[i - 1]) * 0.01);
        if (t[30]) fprintf(stderr, " | stream workgroup: start +%.1f, first draws ready +%.1f", (double)((long long)t[30] - (long long)t[0]) * 0.01, (double)((long long)t[31] - (long long)t[0]) * 0.01);
        fprintf(stderr, "\n");
    }
    if (host[1] != LFD_SELECT_OK)
        return fail(ctx, host[1] == LFD_SELECT_CAPACITY ? LFD_ERR_CAPACITY : LFD_ERR_INVALID,
                    std::string("selection: ") + select_status_name(host[1]));
    return LFD_OK;
}

// seeds: one MT19937 stream per reference (lfd_triangulate_sampled_multi).  chain: all references on the CONTEXT's stream, consumed in batch order
// (lfd_triangulate_sampled_chain; s_chain: their normalisers, or null).  Neither: one reference on the context's stream.
static int sampled_impl(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap, int32_t border, int32_t tiles,
                        float s_override, const uint32_t* seeds, const lfd_points* out, int64_t* ref_offsets, int32_t* seg_counts,
                        int32_t* seg_order, int32_t* sel_info, int64_t* sel_cells, bool chain = false, const float* s_chain = nullptr) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (!batch || !params || !sel_info) return fail(ctx, LFD_ERR_INVALID, "null argument");
    const int R = batch->n_refs;
    if (R != 1 && !seeds && !chain) return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_sampled takes one reference view per call (several: lfd_triangulate_sampled_chain on the context's stream, lfd_triangulate_sampled_multi on per-reference streams)");
    if (R < 1) return fail(ctx, LFD_ERR_INVALID, "no reference view");
    if (M < 0 || tiles <= 0 || border < 0) return fail(ctx, LFD_ERR_INVALID, "bad selection arguments");
    const bool topm = params->no_filter != 0;
    if (topm && M > LFD_SELECT_TOPM_MAX) return fail(ctx, LFD_ERR_INVALID, "no_filter selection is limited to 16384 matches per reference");
    LfdLaunch L;
    int rc = prepare_launch(ctx, batch, params, nullptr, 0, L, nullptr);
    if (rc != LFD_OK) return rc;
    rc = check_points(ctx, out, reinterpret_cast<long long*>(ref_offsets));
    if (rc != LFD_OK) return rc;
    const long long HW = (long long)batch->H * batch->W;
    const long long cap_sel = topm ? std::max<long long>(std::min<long long>(M, HW), 1) : (long long)M + (long long)tiles * tiles + 64;
    if (out->capacity < cap_sel * R) return fail(ctx, LFD_ERR_CAPACITY, "output capacity below (M + tiles*tiles + 64) per reference");
    // P1 + F1 for every reference of the batch in one launch
    rc = ensure(ctx, ctx->agg, (size_t)HW * (size_t)R * sizeof(float));
    if (rc != LFD_OK) return rc;
    float* best = static_cast<float*>(ctx->agg.ptr);
    {
        const int per_block = 256 * 4;
        const int gx = (int)std::min<long long>((HW + per_block - 1) / per_block, 2048);
        hipLaunchKernelGGL(lfd_aggregate_kernel, dim3((unsigned)gx, (unsigned)R, 1u), dim3(256), 0, ctx->stream, L, best, static_cast<uint8_t*>(nullptr));
        LFD_HIP(ctx, hipGetLastError());
    }
    // S, reference after reference (the selection kernels share one scratch area): the counts stay on the device as {begin, end}
    // pairs for the kernels below; every reference's cells sit at a fixed stride
    rc = ensure(ctx, ctx->sel_buf, (size_t)R * 16 + (size_t)cap_sel * (size_t)R * sizeof(long long));
    if (rc != LFD_OK) return rc;
    long long* sel_pairs = static_cast<long long*>(ctx->sel_buf.ptr);
    long long* cells = sel_cells ? reinterpret_cast<long long*>(sel_cells) : sel_pairs + 2 * (size_t)R;
    if (R == 1) {
        LFD_HIP(ctx, hipMemsetAsync(sel_pairs, 0, 16, ctx->stream));
    } else {
        // (written on the device: a copy from a host vector had to be waited for - a drain of the stream inside every grouped call, which kept the
        // caller from running ahead of the device)
        hipLaunchKernelGGL(lfd_select_begins_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, ctx->stream, sel_pairs, (long long)cap_sel, R);
        LFD_HIP(ctx, hipGetLastError());
    }
    if (!seeds && (topm || select_runs_on_several_workgroups(ctx, HW)) && R > 1) {
        // the context's stream, R references in ONE launch per LFD_SELECT_BATCH_MAX of them: everything that does not depend on the stream runs
        // side by side, a reference starts drawing where the one before it stopped (no_filter draws nothing: plain side by side)
        if (!topm && !ctx->mt_seeded) return fail(ctx, LFD_ERR_STATE, "lfd_rng_seed must be called before lfd_triangulate_sampled_chain");
        for (int r0 = 0; r0 < R; r0 += LFD_SELECT_BATCH_MAX) {
            const int nb = std::min(R - r0, (int)LFD_SELECT_BATCH_MAX);
            int* d_info = nullptr;
            unsigned char* d_time = nullptr;
            rc = select_launch(ctx, topm, best + (size_t)r0 * HW, batch->H, batch->W, M, cap, topm ? 0 : border, topm ? 1 : tiles, 0.0f,
                               reinterpret_cast<int64_t*>(cells + (size_t)r0 * cap_sel), cap_sel, sel_pairs + 2 * (size_t)r0, &d_info, &d_time,
                               nb, nullptr, sel_info + 2 * (size_t)r0, (s_chain && !topm) ? s_chain + r0 : nullptr);
            if (rc != LFD_OK) return rc;
        }
    } else if (!seeds) {     // the context's MT19937 stream (upstream's single global stream), reference after reference
        for (int r = 0; r < R; ++r) {
            int* d_info = nullptr;
            unsigned char* d_time = nullptr;
            const float s_r = topm ? 0.0f : (s_chain ? s_chain[r] : s_override);
            rc = select_launch(ctx, topm, best + (size_t)r * HW, batch->H, batch->W, M, cap, topm ? 0 : border, topm ? 1 : tiles, s_r,
                               reinterpret_cast<int64_t*>(cells + (size_t)r * cap_sel), cap_sel, sel_pairs + 2 * (size_t)r, &d_info, &d_time);
            if (rc != LFD_OK) return rc;
            LFD_HIP(ctx, hipMemcpyAsync(sel_info + 2 * (size_t)r, d_info, 2 * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        }
    } else {                 // every reference on its own stream: up to LFD_SELECT_BATCH_MAX selections side by side per launch
        rc = ensure(ctx, ctx->mt_batch, (size_t)LFD_SELECT_BATCH_MAX * LFD_MT_STATE_STRIDE * sizeof(unsigned));
        if (rc != LFD_OK) return rc;
        unsigned* mtb = static_cast<unsigned*>(ctx->mt_batch.ptr);
        for (int r0 = 0; r0 < R; r0 += LFD_SELECT_BATCH_MAX) {
            const int nb = std::min(R - r0, (int)LFD_SELECT_BATCH_MAX);
            if (!topm) {
                LfdSeedBatch sb;
                std::memset(&sb, 0, sizeof(sb));
                for (int i = 0; i < nb; ++i) sb.seed[i] = seeds[r0 + i];
                hipLaunchKernelGGL(lfd_mt_seed_batch_kernel, dim3((unsigned)nb), dim3(64), 0, ctx->stream, mtb, sb);
                LFD_HIP(ctx, hipGetLastError());
            }
            int* d_info = nullptr;
            unsigned char* d_time = nullptr;
            rc = select_launch(ctx, topm, best + (size_t)r0 * HW, batch->H, batch->W, M, cap, topm ? 0 : border, topm ? 1 : tiles, 0.0f,
                               reinterpret_cast<int64_t*>(cells + (size_t)r0 * cap_sel), cap_sel, sel_pairs + 2 * (size_t)r0, &d_info, &d_time,
                               nb, mtb, sel_info + 2 * (size_t)r0);
            if (rc != LFD_OK) return rc;
        }
    }
    // F2..F10 on the selected cells of all references
    rc = prepare_lookback(ctx, (size_t)R, (size_t)R, false, L);
    if (rc != LFD_OK) return rc;
    rc = ensure(ctx, ctx->scratch, (size_t)cap_sel * (size_t)R * 8 * sizeof(float));
    if (rc != LFD_OK) return rc;
    rc = ensure(ctx, ctx->codes, (size_t)cap_sel * (size_t)R);
    if (rc != LFD_OK) return rc;
    const size_t tab_bytes = (size_t)R * LFD_MAX_SLOTS * 2 * sizeof(unsigned);
    rc = ensure(ctx, ctx->idx_tab, tab_bytes);
    if (rc != LFD_OK) return rc;
    unsigned* tab = static_cast<unsigned*>(ctx->idx_tab.ptr);
    LFD_HIP(ctx, hipMemsetAsync(tab, 0, tab_bytes, ctx->stream));
    L.xyz = out->xyz; L.rgb = out->rgb; L.err = out->err; L.cell = out->cell; L.slot = out->slot;
    L.capacity = out->capacity;
    L.ref_offsets = reinterpret_cast<long long*>(ref_offsets);
    L.seg_counts = seg_counts;
    const unsigned chunks = (unsigned)((cap_sel + LFD_INDEXED_EVAL_BLOCK - 1) / LFD_INDEXED_EVAL_BLOCK);
    hipLaunchKernelGGL(lfd_indexed_eval_kernel, dim3(chunks, (unsigned)R), dim3(LFD_INDEXED_EVAL_BLOCK), 0, ctx->stream, L,
                       static_cast<const long long*>(cells), static_cast<const long long*>(sel_pairs), static_cast<float*>(ctx->scratch.ptr),
                       static_cast<uint8_t*>(ctx->codes.ptr), tab, 1);
    LFD_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(lfd_indexed_kernel, dim3((unsigned)R), dim3(LFD_INDEXED_BLOCK), 0, ctx->stream, L,
                       static_cast<const long long*>(cells), static_cast<const long long*>(sel_pairs), static_cast<float*>(ctx->scratch.ptr),
                       static_cast<uint8_t*>(ctx->codes.ptr), seg_order, static_cast<const unsigned*>(tab), 1);
    LFD_HIP(ctx, hipGetLastError());
    // the look-back status word of this launch rides along with the counts: one read-back tells the caller everything
    LFD_HIP(ctx, hipMemcpyAsync(sel_info + 2 * (size_t)R, static_cast<unsigned char*>(ctx->ws.ptr) + 8, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    return LFD_OK;
}

int lfd_triangulate_sampled(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                            int32_t border, int32_t tiles, float s_override, const lfd_points* out, int64_t* ref_offsets,
                            int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info, int64_t* sel_cells) {
    if (ctx && batch && batch->n_refs != 1) return fail(ctx, LFD_ERR_INVALID, "lfd_triangulate_sampled takes one reference view per call");
    return sampled_impl(ctx, batch, params, M, cap, border, tiles, s_override, nullptr, out, ref_offsets, seg_counts, seg_order, sel_info, sel_cells);
}

int lfd_triangulate_sampled_chain(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                                  int32_t border, int32_t tiles, const float* s_overrides, const lfd_points* out, int64_t* ref_offsets,
                                  int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info, int64_t* sel_cells) {
    return sampled_impl(ctx, batch, params, M, cap, border, tiles, 0.0f, nullptr, out, ref_offsets, seg_counts, seg_order, sel_info, sel_cells, true, s_overrides);
}

int lfd_triangulate_sampled_multi(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                                  int32_t border, int32_t tiles, const uint32_t* seeds, const lfd_points* out, int64_t* ref_offsets,
                                  int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info, int64_t* sel_cells) {
    if (ctx && !seeds) return fail(ctx, LFD_ERR_INVALID, "seeds is required: every reference draws from its own MT19937 stream");
    return sampled_impl(ctx, batch, params, M, cap, border, tiles, 0.0f, seeds, out, ref_offsets, seg_counts, seg_order, sel_info, sel_cells);
}

int lfd_select_samples(lfd_context* ctx, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                       int32_t border, int32_t tiles, float s_override, int64_t* sel_out, int64_t capacity,
                       int32_t* n_sel_host, int32_t* status_host) {
    return select_impl(ctx, false, best_cert, H, W, M, cap, border, tiles, s_override, sel_out, capacity, n_sel_host, status_host);
}

int lfd_select_top_m(lfd_context* ctx, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                     int64_t* sel_out, int64_t capacity, int32_t* n_sel_host, int32_t* status_host) {
    return select_impl(ctx, true, best_cert, H, W, M, cap, 0, 1, 0.0f, sel_out, capacity, n_sel_host, status_host);
}

// ---- N1: writers ----------------------------------------------------------------------------------
int lfd_pack_ply(lfd_context* ctx, const float* xyz, const float* rgb, int64_t n, uint8_t* out) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (n < 0 || (n > 0 && (!xyz || !rgb || !out))) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (reinterpret_cast<uintptr_t>(out) & 3u) return fail(ctx, LFD_ERR_INVALID, "out must be 4-byte aligned");
    if (n == 0) return LFD_OK;
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(lfd_pack_ply_kernel, dim3(grid), dim3(256), 0, ctx->stream, xyz, rgb, (long long)n, out);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_pack_points3d(lfd_context* ctx, const float* xyz, const float* rgb, const float* err, int64_t n, uint64_t id_base,
                      uint8_t* out) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (n < 0 || (n > 0 && (!xyz || !rgb || !out))) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (reinterpret_cast<uintptr_t>(out) & 3u) return fail(ctx, LFD_ERR_INVALID, "out must be 4-byte aligned");
    if (n == 0) return LFD_OK;
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(lfd_pack_points3d_kernel, dim3(grid), dim3(256), 0, ctx->stream, xyz, rgb, err, (long long)n,
                       (unsigned long long)id_base, out);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

int lfd_copy_segments(void* hip_stream, int32_t device_index, const void* src, void* dst, const lfd_copy_segment* segs, int32_t n) {
    if (n < 0 || (n > 0 && (!src || !dst || !segs))) return LFD_ERR_INVALID;
    if (n == 0) return LFD_OK;
    if (hipSetDevice(device_index) != hipSuccess) return LFD_ERR_HIP;
    for (int32_t first = 0; first < n;) {
        LfdCopyArgs A;
        int m = 0;
        long long chunks = 0;
        A.chunk0[0] = 0;
        for (; first < n && m < LFD_COPY_MAX_SEGS; ++first) {
            const lfd_copy_segment& sg = segs[first];
            if (sg.nbytes < 0 || sg.src_offset < 0 || sg.dst_offset < 0) return LFD_ERR_INVALID;
            if (sg.nbytes == 0) continue;
            const long long c = (sg.nbytes + LFD_COPY_CHUNK - 1) / LFD_COPY_CHUNK;
            if (chunks + c > 0x7fffffffLL) { if (m == 0) return LFD_ERR_INVALID; break; }       // (one launch's grid is full: the rest goes into the next)
            A.src[m] = sg.src_offset; A.dst[m] = sg.dst_offset; A.n[m] = sg.nbytes;
            chunks += c;
            A.chunk0[++m] = (int)chunks;
        }
        if (m == 0) continue;
        A.n_segs = m;
        hipLaunchKernelGGL(lfd_copy_segments_kernel, dim3((unsigned)chunks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), A,
                           static_cast<const unsigned char*>(src), static_cast<unsigned char*>(dst));
        if (hipGetLastError() != hipSuccess) return LFD_ERR_HIP;
    }
    return LFD_OK;
}

int lfd_quantise_rgb(lfd_context* ctx, const float* rgb, int64_t n, uint8_t* out) {
    if (!ctx) return fail(nullptr, LFD_ERR_INVALID, "null context");
    if (ctx->is_host) return fail(ctx, LFD_ERR_STATE, "device entry point called on a host context");
    if (n < 0 || (n > 0 && (!rgb || !out))) return fail(ctx, LFD_ERR_INVALID, "bad arguments");
    if (n == 0) return LFD_OK;
    LFD_HIP(ctx, hipSetDevice(ctx->device));
    const long long n3 = 3 * (long long)n;
    const unsigned grid = (unsigned)std::min<long long>((n3 + 255) / 256, 4096);
    hipLaunchKernelGGL(lfd_quantise_rgb_kernel, dim3(grid), dim3(256), 0, ctx->stream, rgb, n3, out);
    LFD_HIP(ctx, hipGetLastError());
    return LFD_OK;
}

// ---- host-side helpers -----------------------------------------------------------------------
int lfd_identity_axis(int32_t n, float* out) {
    if (n <= 0 || !out) return LFD_ERR_INVALID;
    const LfdAxis a = lfd_make_axis(n);
    for (int j = 0; j < n; ++j) out[j] = lfd_axis_value(a, j);
    return LFD_OK;
}

float lfd_parallax_dot_threshold(float min_deg) {
    // ang(d) = (float)acosf(clip(d)) * (180/pi as f32) is non-increasing in d; find the largest f32 d
    // in [-1, 1] with ang(d) >= min_deg by bisection over the ordered f32 bit patterns.
    const float k = 57.295776f;   // 180.0f / NPY_PIf, the constant np.degrees uses for f32
    auto ok = [&](float d) { return acosf(d) * k >= min_deg; };
    if (!ok(-1.0f)) return -2.0f;             // nothing passes (min_deg > 180)
    if (ok(1.0f)) return 1.0f;                // everything in range passes (min_deg <= 0)
    auto key = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i >= 0 ? (int64_t)i : -(int64_t)(i & 0x7fffffff); };
    auto unkey = [](int64_t kx) { int32_t i = kx >= 0 ? (int32_t)kx : (int32_t)(0x80000000u | (uint32_t)(-kx)); float f; std::memcpy(&f, &i, 4); return f; };
    int64_t lo = key(-1.0f), hi = key(1.0f);   // ok(lo) true, ok(hi) false
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (ok(unkey(mid))) lo = mid; else hi = mid;
    }
    return unkey(lo);
}

int lfd_host_fundamental(const float* K1, const float* R1, const float* t1, const float* K2, const float* R2,
                         const float* t2, float* F_out) {
    if (!K1 || !R1 || !t1 || !K2 || !R2 || !t2 || !F_out) return LFD_ERR_INVALID;
    lfd_fundamental(K1, R1, t1, K2, R2, t2, F_out);
    return LFD_OK;
}

static void unpack_cam(const float* v, LfdCam& c) {
    std::memcpy(c.K, v, 9 * 4); std::memcpy(c.R, v + 9, 9 * 4); std::memcpy(c.t, v + 18, 3 * 4);
    std::memcpy(c.P, v + 21, 12 * 4); std::memcpy(c.C, v + 33, 3 * 4);
    c.w = (int32_t)v[36]; c.h = (int32_t)v[37]; c.pad[0] = c.pad[1] = 0;
}

int lfd_host_null_vector(const float* A16, double* out4) {
    if (!A16 || !out4) return -LFD_ERR_INVALID;
    return lfd_null_vector(A16, out4);
}

int lfd_host_eval_correspondence(const float* cam1, const float* cam2, float xa_norm, float ya_norm, float xb_norm,
                                 float yb_norm, int32_t w_match, int32_t h_match, const lfd_params* params, float* out8) {
    if (!cam1 || !cam2 || !params || !out8 || w_match <= 1 || h_match <= 1) return LFD_ERR_INVALID;
    LfdCam a, b;
    unpack_cam(cam1, a); unpack_cam(cam2, b);
    LfdRefConst rc; LfdPairConst pc;
    lfd_make_ref_const(a, w_match, h_match, rc);
    lfd_make_pair_const(a, b, 1, w_match, h_match, pc);
    lfd_batch bb; std::memset(&bb, 0, sizeof(bb));
    bb.w_match = w_match; bb.h_match = h_match;
    LfdKernelParams kp;
    lfd_fill_kernel_params(&bb, params, kp);
    LfdCellResult res;
    lfd_eval_correspondence(rc, pc, xa_norm, ya_norm, xb_norm, yb_norm, kp, res);
    out8[0] = res.x; out8[1] = res.y; out8[2] = res.z; out8[3] = 0.0f; out8[4] = 0.0f; out8[5] = 0.0f;
    out8[6] = res.err; out8[7] = (float)res.keep;
    return LFD_OK;
}

}  // extern "C"
