// CPU twin of the hot path (include/lfd_densify.h, "CPU twin" section): the same per-cell routine as the kernels -
// the HOST build of lfd_geometry.hpp - driven over host arrays by a pool of threads.  It exists for three things:
// upstream's CPU-only configuration (BASELINE config 1: plumbing without a GPU), the `cpu_baseline` leg of bench.py
// (this build's own C++ restatement timed on the host cores) and CPU-side parity tests.  It is never
// reached from a device context and no device entry point falls back to it: a caller opts in with lfd_create_host().
//
// Differences from the device build of the same source, all inside lfd_geometry.hpp's #if blocks: IEEE division and
// square root where the kernels use v_rcp_f32 / v_sqrt_f32 (1 ulp) and a plain 1.0/d where they Newton-refine v_rcp_f64.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "lfd_context.hpp"

void lfd_fill_kernel_params(const lfd_batch* b, const lfd_params* p, LfdKernelParams& kp);   // lfd_api.hip

namespace {

struct HostRef {             // what block_prologue stages per reference on the device
    LfdRefConst rc;
    LfdPairConst pc[LFD_MAX_SLOTS];
};

struct HostLaunch {
    const lfd_batch* b;
    LfdKernelParams kp;
    std::vector<float> axis_x, axis_y;      // the matcher's linspace when the caller passes none
    const float* ax;
    const float* ay;
    float mask_sx, mask_sy;
    int HW;
    bool exact_colour;
};

}  // namespace

// The host context's worker threads: started once by lfd_create_host, parked on a condition variable between calls, joined by
// lfd_destroy.  A call hands them a chunk count and a function; chunks are claimed with one atomic counter (the calling thread
// works too), so the cost per call is one wake-up instead of n_threads thread creations.
struct LfdHostPool {
    explicit LfdHostPool(int n_threads) {
        const int extra = std::max(0, n_threads - 1);         // the caller is a worker as well
        workers.reserve((size_t)extra);
        for (int t = 0; t < extra; ++t) workers.emplace_back([this]() { loop(); });
    }
    ~LfdHostPool() {
        { std::lock_guard<std::mutex> g(m); stop = true; ++generation; }
        wake.notify_all();
        for (auto& th : workers) th.join();
    }
    void run(int n_chunks, int max_threads, const std::function<void(int)>& fn) {
        if (n_chunks <= 0) return;
        const int helpers = std::min({(int)workers.size(), std::max(0, max_threads - 1), n_chunks - 1});
        if (helpers == 0) { for (int c = 0; c < n_chunks; ++c) fn(c); return; }
        {
            std::lock_guard<std::mutex> g(m);
            job = &fn; total = n_chunks; next.store(0); wanted = helpers; joined = 0; finished = 0; ++generation;
        }
        wake.notify_all();
        for (int c = next.fetch_add(1); c < n_chunks; c = next.fetch_add(1)) fn(c);
        std::unique_lock<std::mutex> g(m);
        wanted = joined;                                      // no late joiner may start on this job any more
        done.wait(g, [this]() { return finished == joined; });
        job = nullptr;
    }

private:
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn = nullptr;
            int n = 0;
            {
                std::unique_lock<std::mutex> g(m);
                wake.wait(g, [&]() { return generation != seen; });
                seen = generation;
                if (stop) return;
                if (job == nullptr || joined >= wanted) continue;
                ++joined; fn = job; n = total;
            }
            for (int c = next.fetch_add(1); c < n; c = next.fetch_add(1)) (*fn)(c);
            { std::lock_guard<std::mutex> g(m); ++finished; }
            done.notify_one();
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake, done;
    const std::function<void(int)>* job = nullptr;
    std::atomic<int> next{0};
    int total = 0, wanted = 0, joined = 0, finished = 0;
    unsigned long long generation = 0;
    bool stop = false;
};

void lfd_host_pool_destroy(LfdHostPool* p) { delete p; }      // lfd_destroy (lfd_api.hip)

namespace {

template <class Fn>
void parallel_chunks(lfd_context* ctx, int n_chunks, Fn fn) {
    const std::function<void(int)> f(fn);
    if (ctx->host_pool) ctx->host_pool->run(n_chunks, ctx->host_threads, f);
    else for (int c = 0; c < n_chunks; ++c) f(c);
}

int validate_host(lfd_context* ctx, const lfd_batch* b, const lfd_params* p) {
    if (!ctx) return lfd_fail(nullptr, LFD_ERR_INVALID, "null context");
    if (!ctx->is_host) return lfd_fail(ctx, LFD_ERR_STATE, "the *_host entry points need a context made by lfd_create_host");
    if (!b || !p) return lfd_fail(ctx, LFD_ERR_INVALID, "null batch/params");
    if (ctx->host_cams.empty()) return lfd_fail(ctx, LFD_ERR_STATE, "lfd_upload_cameras must be called first");
    const int n_cams = (int)ctx->host_cams.size();
    if (b->n_refs <= 0) return lfd_fail(ctx, LFD_ERR_INVALID, "n_refs must be > 0");
    if (b->k <= 0 || b->k > LFD_MAX_SLOTS) return lfd_fail(ctx, LFD_ERR_INVALID, "k must be in [1, LFD_MAX_SLOTS]");
    if (b->H <= 0 || b->W <= 0 || b->w_match <= 1 || b->h_match <= 1) return lfd_fail(ctx, LFD_ERR_INVALID, "bad grid / match size");
    if ((long long)b->H * b->W > 0x7fffffffLL) return lfd_fail(ctx, LFD_ERR_INVALID, "grid too large");
    if (b->warp_channels != 2 && b->warp_channels != 4) return lfd_fail(ctx, LFD_ERR_INVALID, "warp_channels must be 2 or 4");
    if (!b->ref_cam || !b->n_slots || !b->nbr_cam || !b->cert || !b->warp || !b->image) return lfd_fail(ctx, LFD_ERR_INVALID, "null table in batch");
    if ((b->axis_x == nullptr) != (b->axis_y == nullptr)) return lfd_fail(ctx, LFD_ERR_INVALID, "axis_x and axis_y must both be given or both be null");
    for (int r = 0; r < b->n_refs; ++r) {
        if (b->ref_cam[r] < 0 || b->ref_cam[r] >= n_cams) return lfd_fail(ctx, LFD_ERR_INVALID, "ref_cam out of range");
        if (b->n_slots[r] < 1 || b->n_slots[r] > b->k) return lfd_fail(ctx, LFD_ERR_INVALID, "n_slots must be in [1, k]");
        if (!b->image[r]) return lfd_fail(ctx, LFD_ERR_INVALID, "null image pointer");
        for (int j = 0; j < b->n_slots[r]; ++j) {
            const size_t s = (size_t)r * b->k + j;
            if (b->nbr_cam[s] < 0 || b->nbr_cam[s] >= n_cams) return lfd_fail(ctx, LFD_ERR_INVALID, "nbr_cam out of range");
            if (!b->cert[s] || !b->warp[s]) return lfd_fail(ctx, LFD_ERR_INVALID, "null cert/warp pointer in a valid slot");
        }
    }
    return LFD_OK;
}

void prepare_host(const lfd_batch* b, const lfd_params* p, HostLaunch& L) {
    L.b = b;
    lfd_fill_kernel_params(b, p, L.kp);
    L.HW = b->H * b->W;
    L.mask_sx = (float)b->w_match / (float)b->W;
    L.mask_sy = (float)b->h_match / (float)b->H;
    L.exact_colour = (p->flags & LFD_FLAG_EXACT_COLOUR) != 0;
    if (b->axis_x) { L.ax = b->axis_x; L.ay = b->axis_y; }
    else {
        L.axis_x.resize((size_t)b->W); L.axis_y.resize((size_t)b->H);
        const LfdAxis ax = lfd_make_axis(b->W), ay = lfd_make_axis(b->H);
        for (int j = 0; j < b->W; ++j) L.axis_x[(size_t)j] = lfd_axis_value(ax, j);
        for (int j = 0; j < b->H; ++j) L.axis_y[(size_t)j] = lfd_axis_value(ay, j);
        L.ax = L.axis_x.data(); L.ay = L.axis_y.data();
    }
}

void make_ref(const lfd_context* ctx, const HostLaunch& L, int r, HostRef& R) {
    const lfd_batch* b = L.b;
    const LfdCam& ca = ctx->host_cams[(size_t)b->ref_cam[r]];
    lfd_make_ref_const(ca, b->w_match, b->h_match, R.rc);
    for (int j = 0; j < b->n_slots[r]; ++j) {
        const size_t s = (size_t)r * b->k + j;
        lfd_make_pair_const(ca, ctx->host_cams[(size_t)b->nbr_cam[s]], b->nbr_cam[s], b->w_match, b->h_match, R.pc[j],
                            b->fundamental ? b->fundamental + s * 9 : nullptr);
    }
}

// certainty of slot j at one cell after the prologue of core/pipeline.py:407-430 (cell_cert of lfd_kernels.hip)
float host_cell_cert(const HostLaunch& L, int r, int j, int cell, int x, int y) {
    const lfd_batch* b = L.b;
    const size_t s = (size_t)r * b->k + j;
    float c = lfd_cert_floor(b->cert[s][cell], L.kp.certainty_thresh);
    const uint8_t* ma = b->mask_a ? b->mask_a[r] : nullptr;
    if (ma) c = c * (float)ma[(size_t)lfd_nearest_src(y, L.mask_sy, b->h_match) * b->w_match + lfd_nearest_src(x, L.mask_sx, b->w_match)];
    const uint8_t* mb = b->mask_b ? b->mask_b[s] : nullptr;
    if (mb) {
        const float* wv = b->warp[s] + (size_t)cell * b->warp_channels + (b->warp_channels - 2);
        const int ix = lfd_grid_nearest(wv[0], b->W), iy = lfd_grid_nearest(wv[1], b->H);
        float m = 0.0f;
        if (ix >= 0 && iy >= 0)
            m = (float)mb[(size_t)lfd_nearest_src(iy, L.mask_sy, b->h_match) * b->w_match + lfd_nearest_src(ix, L.mask_sx, b->w_match)];
        c = c * m;
    }
    return c;
}

// torch.max(dim=0): first maximum wins, a NaN beats any number (first NaN)
void host_cell_best(const HostLaunch& L, int r, int cell, float& best, int& bj) {
    const int y = cell / L.b->W, x = cell - y * L.b->W;
    best = host_cell_cert(L, r, 0, cell, x, y);
    bj = 0;
    for (int j = 1; j < L.b->n_slots[r]; ++j) {
        const float c = host_cell_cert(L, r, j, cell, x, y);
        if ((c > best) || ((c != c) && !(best != best))) { best = c; bj = j; }
    }
}

struct HostPoint { float x, y, z, r, g, b, err; int cell; int slot; };

// one cell through arg-max -> geometry -> colour; returns false when the cell does not survive
bool host_eval_cell(const HostLaunch& L, const HostRef& R, int r, int cell, HostPoint& o, int& bj_out, bool need_weight = false) {
    const lfd_batch* b = L.b;
    float best; int bj;
    host_cell_best(L, r, cell, best, bj);
    bj_out = bj;
    // dense mode: only cells upstream's sampler could ever draw - a weight that is not <= 0 after floor and masks (core/sampling.py:24-27, 41-43 upstream:
    // p = weights / sum, the coverage pass stops at weights <= 0); a masked-out cell is not a candidate.  Indexed mode: the caller selected.
    if (need_weight && best <= 0.0f) return false;
    const float* wp = b->warp[(size_t)r * b->k + bj] + (size_t)cell * b->warp_channels;
    float xan, yan, xbn, ybn;
    if (b->warp_channels == 4) { xan = wp[0]; yan = wp[1]; xbn = wp[2]; ybn = wp[3]; }
    else { const int y = cell / b->W, x = cell - y * b->W; xan = L.ax[x]; yan = L.ay[y]; xbn = wp[0]; ybn = wp[1]; }
    LfdCellResult res;
    lfd_eval_correspondence(R.rc, R.pc[bj], xan, yan, xbn, ybn, L.kp, res);
    if (!res.keep) return false;
    float rgb[3];
    if (L.exact_colour) lfd_bilinear_rgb(b->image[r], b->w_match, b->h_match, res.xa_px, res.ya_px, 1.0f, 1.0f, rgb);
    else lfd_bilinear_rgb_f32(b->image[r], b->w_match, b->h_match, res.xa_px, res.ya_px, rgb);
    o.x = res.x; o.y = res.y; o.z = res.z; o.r = rgb[0]; o.g = rgb[1]; o.b = rgb[2]; o.err = res.err; o.cell = cell; o.slot = bj;
    return true;
}

void store_point(const lfd_points* out, long long pos, const HostPoint& p) {
    if (pos >= out->capacity) return;           // beyond capacity: counted, not written (as on the device)
    out->xyz[pos * 3 + 0] = p.x; out->xyz[pos * 3 + 1] = p.y; out->xyz[pos * 3 + 2] = p.z;
    out->rgb[pos * 3 + 0] = p.r; out->rgb[pos * 3 + 1] = p.g; out->rgb[pos * 3 + 2] = p.b;
    out->err[pos] = p.err;
    if (out->cell) out->cell[pos] = p.cell;
    if (out->slot) out->slot[pos] = (uint8_t)p.slot;
}

int check_host_points(lfd_context* ctx, const lfd_points* out, const int64_t* ref_offsets) {
    if (!out || !out->xyz || !out->rgb || !out->err) return lfd_fail(ctx, LFD_ERR_INVALID, "null output buffers");
    if (out->capacity < 0) return lfd_fail(ctx, LFD_ERR_INVALID, "negative capacity");
    if (!ref_offsets) return lfd_fail(ctx, LFD_ERR_INVALID, "ref_offsets is required");
    return LFD_OK;
}

constexpr int kChunk = 4096;     // cells per work item

}  // namespace

extern "C" {

int lfd_create_host(int32_t n_threads, lfd_context** out) {
    if (!out) return lfd_fail(nullptr, LFD_ERR_INVALID, "out is null");
    lfd_context* ctx = new lfd_context();
    ctx->is_host = true;
    const unsigned hw = std::thread::hardware_concurrency();
    ctx->host_threads = n_threads > 0 ? n_threads : (hw ? (int)hw : 1);
    if (ctx->host_threads > 1) ctx->host_pool = new LfdHostPool(ctx->host_threads);
    *out = ctx;
    return LFD_OK;
}

int lfd_host_threads(const lfd_context* ctx) { return (ctx && ctx->is_host) ? ctx->host_threads : 0; }

int lfd_aggregate_host(lfd_context* ctx, const lfd_batch* b, const lfd_params* p, float* best_cert, uint8_t* best_slot) {
    int rc = validate_host(ctx, b, p);
    if (rc != LFD_OK) return rc;
    if (!best_cert) return lfd_fail(ctx, LFD_ERR_INVALID, "best_cert is null");
    HostLaunch L;
    prepare_host(b, p, L);
    const int chunks_per_ref = (L.HW + kChunk - 1) / kChunk;
    parallel_chunks(ctx, b->n_refs * chunks_per_ref, [&](int c) {
        const int r = c / chunks_per_ref, c0 = (c - r * chunks_per_ref) * kChunk, c1 = std::min(c0 + kChunk, L.HW);
        for (int cell = c0; cell < c1; ++cell) {
            float best; int bj;
            host_cell_best(L, r, cell, best, bj);
            best_cert[(size_t)r * L.HW + cell] = best;
            if (best_slot) best_slot[(size_t)r * L.HW + cell] = (uint8_t)bj;
        }
    });
    return LFD_OK;
}

int lfd_triangulate_dense_host(lfd_context* ctx, const lfd_batch* b, const lfd_params* p, const lfd_points* out,
                               int64_t* ref_offsets, int32_t* seg_counts) {
    int rc = validate_host(ctx, b, p);
    if (rc != LFD_OK) return rc;
    rc = check_host_points(ctx, out, ref_offsets);
    if (rc != LFD_OK) return rc;
    HostLaunch L;
    prepare_host(b, p, L);
    const int chunks_per_ref = (L.HW + kChunk - 1) / kChunk;
    std::vector<HostRef> refs((size_t)b->n_refs);
    for (int r = 0; r < b->n_refs; ++r) make_ref(ctx, L, r, refs[(size_t)r]);
    // Every chunk parks its survivors at its own fixed place of a staging area that belongs to the context (chunk c at
    // c * kChunk: no allocation, no growth, nothing shared between threads); a prefix over the chunk counts then gives every
    // chunk its place in the output - references in batch order, cells in raster order, the device's look-back scan - and a
    // second parallel sweep moves the records there.  References are taken in groups so that the staging area stays bounded.
    const int refs_per_group = std::max(1, (int)(((size_t)256 << 20) / (sizeof(HostPoint) * (size_t)chunks_per_ref * kChunk)));
    const int group_chunks = std::min(b->n_refs, refs_per_group) * chunks_per_ref;
    if (ctx->host_stage.size() < (size_t)group_chunks * kChunk * sizeof(HostPoint)) ctx->host_stage.resize((size_t)group_chunks * kChunk * sizeof(HostPoint));
    HostPoint* stage = reinterpret_cast<HostPoint*>(ctx->host_stage.data());
    std::vector<int> kept((size_t)group_chunks);
    std::vector<int> per_slot(seg_counts ? (size_t)group_chunks * LFD_MAX_SLOTS : 0);
    std::vector<long long> start((size_t)group_chunks + 1);
    if (seg_counts) std::memset(seg_counts, 0, sizeof(int32_t) * (size_t)b->n_refs * b->k);
    long long total = 0;
    for (int r0 = 0; r0 < b->n_refs; r0 += refs_per_group) {
        const int nr = std::min(refs_per_group, b->n_refs - r0), nc = nr * chunks_per_ref;
        parallel_chunks(ctx, nc, [&](int c) {
            const int r = r0 + c / chunks_per_ref, c0 = (c % chunks_per_ref) * kChunk, c1 = std::min(c0 + kChunk, L.HW);
            HostPoint* v = stage + (size_t)c * kChunk;
            int n = 0, bj;
            for (int cell = c0; cell < c1; ++cell)
                if (host_eval_cell(L, refs[(size_t)r], r, cell, v[n], bj, true)) ++n;
            kept[(size_t)c] = n;
            if (seg_counts) {
                int* cnt = per_slot.data() + (size_t)c * LFD_MAX_SLOTS;
                for (int j = 0; j < LFD_MAX_SLOTS; ++j) cnt[j] = 0;
                for (int i = 0; i < n; ++i) cnt[v[i].slot] += 1;
            }
        });
        start[0] = total;
        for (int c = 0; c < nc; ++c) start[(size_t)c + 1] = start[(size_t)c] + kept[(size_t)c];
        for (int r = 0; r < nr; ++r) ref_offsets[r0 + r] = start[(size_t)r * chunks_per_ref];
        total = start[(size_t)nc];
        parallel_chunks(ctx, nc, [&](int c) {
            long long pos = start[(size_t)c];
            const HostPoint* v = stage + (size_t)c * kChunk;
            for (int i = 0; i < kept[(size_t)c]; ++i) store_point(out, pos++, v[i]);
        });
        if (seg_counts)
            for (int c = 0; c < nc; ++c)
                for (int j = 0; j < b->k; ++j) seg_counts[(size_t)(r0 + c / chunks_per_ref) * b->k + j] += per_slot[(size_t)c * LFD_MAX_SLOTS + j];
    }
    ref_offsets[b->n_refs] = total;
    if (total > out->capacity) return lfd_fail(ctx, LFD_ERR_CAPACITY, "output capacity too small (counts are valid)");
    return LFD_OK;
}

int lfd_triangulate_indexed_host(lfd_context* ctx, const lfd_batch* b, const lfd_params* p, const int64_t* sel_idx,
                                 const int64_t* sel_offsets, const lfd_points* out, int64_t* ref_offsets, int32_t* seg_counts,
                                 int32_t* seg_order) {
    int rc = validate_host(ctx, b, p);
    if (rc != LFD_OK) return rc;
    if (!sel_idx || !sel_offsets) return lfd_fail(ctx, LFD_ERR_INVALID, "sel_idx / sel_offsets are required");
    if (sel_offsets[0] != 0) return lfd_fail(ctx, LFD_ERR_INVALID, "sel_offsets[0] must be 0");
    for (int r = 0; r < b->n_refs; ++r)
        if (sel_offsets[r + 1] < sel_offsets[r]) return lfd_fail(ctx, LFD_ERR_INVALID, "sel_offsets must be non-decreasing");
    rc = check_host_points(ctx, out, ref_offsets);
    if (rc != LFD_OK) return rc;
    HostLaunch L;
    prepare_host(b, p, L);
    L.exact_colour = true;       // the upstream-equivalent mode always blends in f64, like lfd_triangulate_indexed
    long long total = 0;
    for (int r = 0; r < b->n_refs; ++r) {
        HostRef R;
        make_ref(ctx, L, r, R);
        const long long s0 = sel_offsets[r], n_sel = sel_offsets[r + 1] - s0;
        std::vector<HostPoint> pts((size_t)n_sel);
        std::vector<int8_t> code((size_t)n_sel, (int8_t)-1);          // -1 dropped index, else slot | 0x40 when kept
        const int n_chunks = (int)((n_sel + kChunk - 1) / kChunk);
        parallel_chunks(ctx, n_chunks, [&](int c) {
            const long long i0 = (long long)c * kChunk, i1 = std::min<long long>(i0 + kChunk, n_sel);
            for (long long i = i0; i < i1; ++i) {
                const long long cl = sel_idx[s0 + i];
                if (cl < 0 || cl >= L.HW) continue;                   // invalid selection index: dropped
                int bj = 0;
                const bool keep = host_eval_cell(L, R, r, (int)cl, pts[(size_t)i], bj);
                code[(size_t)i] = (int8_t)(bj | (keep ? 0x40 : 0));
            }
        });
        // groups in order of first appearance while scanning sel_idx (core/pipeline.py:685-688), members in sel_idx order
        int order[LFD_MAX_SLOTS], n_groups = 0;
        long long count[LFD_MAX_SLOTS] = {0};
        bool seen[LFD_MAX_SLOTS] = {false};
        for (long long i = 0; i < n_sel; ++i) {
            if (code[(size_t)i] < 0) continue;
            const int j = code[(size_t)i] & 0x3f;
            if (!seen[j]) { seen[j] = true; order[n_groups++] = j; }
            if (code[(size_t)i] & 0x40) count[j] += 1;
        }
        long long begin[LFD_MAX_SLOTS] = {0}, acc = total;
        int g_out = 0;
        for (int g = 0; g < n_groups; ++g) {
            const int j = order[g];
            begin[j] = acc; acc += count[j];
            if (count[j] && seg_order) seg_order[(size_t)r * b->k + g_out] = j;
            if (count[j]) ++g_out;
        }
        if (seg_order) for (; g_out < b->k; ++g_out) seg_order[(size_t)r * b->k + g_out] = -1;
        if (seg_counts) for (int j = 0; j < b->k; ++j) seg_counts[(size_t)r * b->k + j] = (j < b->n_slots[r]) ? (int32_t)count[j] : 0;
        for (long long i = 0; i < n_sel; ++i) {
            if (code[(size_t)i] < 0 || !(code[(size_t)i] & 0x40)) continue;
            HostPoint& pt = pts[(size_t)i];
            store_point(out, begin[pt.slot]++, pt);
        }
        ref_offsets[r] = total;
        total = acc;
    }
    ref_offsets[b->n_refs] = total;
    if (total > out->capacity) return lfd_fail(ctx, LFD_ERR_CAPACITY, "output capacity too small (counts are valid)");
    return LFD_OK;
}

}  // extern "C"
