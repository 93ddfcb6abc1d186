// HIP kernels of the dense-initialisation hot path (gfx950 / MI355X, wave64).
//
//   lfd_aggregate_kernel      P1+F1   certainty floor, masks, per-cell arg-max        (HBM-bound stream)
//   lfd_dense_kernel          P1..F10 the fused kernel over the whole H x W grid      (HBM / f64-VALU)
//   lfd_indexed_kernel        F2..F10 upstream-equivalent: only the selected cells, upstream's order
//
// Layout in HBM: certainty is one f32 plane [H*W] per (reference, neighbour slot); the warp is one
// [H*W*C] f32 plane per slot (C=2: xB,yB; C=4: xA,yA,xB,yB); the reference image is u8 HWC at match
// resolution.  Planes are addressed through a small descriptor table so RoMa's output tensors are
// consumed in place (no stacking copy).  Camera blocks and the k fundamental matrices of a
// reference are derived once per workgroup and live in LDS.
//
// Per-correspondence arithmetic, no contraction: MFMA is not used on purpose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfd_device.hpp"

namespace {

constexpr int kBlock = LFD_DENSE_BLOCK;       // 256 threads = 4 waves
constexpr int kCpt = LFD_DENSE_CPT;           // consecutive cells per thread
constexpr int kTile = kBlock * kCpt;          // cells per workgroup

typedef unsigned long long u64;

// ---- tile-state word for the decoupled look-back: [63:62] status, [61:40] launch epoch, [39:0] value
constexpr u64 kStEmpty = 0ull, kStAggregate = 1ull, kStPrefix = 2ull;
constexpr u64 kValueMask = (1ull << LFD_VALUE_BITS) - 1ull;
__device__ __forceinline__ u64 pack_state(u64 st, unsigned epoch, u64 v) {
    return (st << 62) | ((u64)epoch << LFD_VALUE_BITS) | (v & kValueMask);
}
// status of a word as seen by launch `epoch`: words written by earlier launches count as empty
__device__ __forceinline__ u64 state_status(u64 s, unsigned epoch) {
    return (((s >> LFD_VALUE_BITS) & LFD_EPOCH_MASK) == epoch) ? (s >> 62) : kStEmpty;
}

__device__ __forceinline__ void state_store(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 state_load(const u64* p) {
    return __hip_atomic_load(const_cast<u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off, 64);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        v += ((u64)hi << 32) | lo;
    }
    return v;
}

// Exclusive prefix of `my_total` over all tiles with a smaller ticket.  Called by wave 0 only; every
// lane returns the same value.  Each tile publishes ONE 8-byte word {status,value} with a relaxed
// agent-scope store (the data is the flag, so no fence is needed); predecessors are guaranteed to
// be running because tickets are handed out by an atomic counter.
__device__ u64 lookback_exclusive(const LfdLaunch& L, unsigned tile, u64 my_total) {
    u64* state = L.tile_state;
    const unsigned epoch = L.epoch;
    const int lane = lane_id();
    unsigned spins = 0;
    if (tile == 0) {
        if (lane == 0) state_store(state, pack_state(kStPrefix, epoch, my_total));
        return 0;
    }
    if (lane == 0) state_store(state + tile, pack_state(kStAggregate, epoch, my_total));
    u64 excl = 0;
    long long base = (long long)tile - 1;
    while (true) {
        const long long j = base - lane;
        u64 s;
        if (j >= 0) {
            s = state_load(state + j);
            while (__any(state_status(s, epoch) == kStEmpty)) {
                if (++spins > LFD_SPIN_LIMIT) {          // never hang the GPU: report and bail out
                    if (lane == 0) atomicExch(L.status, LFD_LAUNCH_TIMEOUT);
                    return 0;
                }
                __builtin_amdgcn_s_sleep(1);
                if (state_status(s, epoch) == kStEmpty) s = state_load(state + j);
            }
        } else {
            s = pack_state(kStPrefix, epoch, 0);   // virtual tile -1: prefix 0
        }
        const u64 is_prefix = __ballot(state_status(s, epoch) == kStPrefix);
        const u64 val = s & kValueMask;
        if (is_prefix) {
            const int first = __ffsll((long long)is_prefix) - 1;   // nearest predecessor with a full prefix
            excl += wave_sum_u64(lane <= first ? val : 0ull);
            break;
        }
        excl += wave_sum_u64(val);
        base -= 64;
    }
    if (lane == 0) state_store(state + tile, pack_state(kStPrefix, epoch, excl + my_total));
    return excl;
}

// ---- workgroup prologue: descriptors + per-pair constants into LDS -------------------------------
struct BlockShared {
    LfdPairConst pc[LFD_MAX_SLOTS];
    LfdRefConst rc;
    LfdSlotDesc slot[LFD_MAX_SLOTS];
    LfdRefDesc ref;
};

__device__ __forceinline__ void block_prologue(const LfdLaunch& L, int r, BlockShared& S) {
    // One level of global loads, then one barrier: the descriptors and the per-pair constants of all k
    // slots of reference r are fetched together (rows of unused slots are never read afterwards).
    const int tid = (int)threadIdx.x;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(L.pair_const + (size_t)r * L.k);
        unsigned* dst = reinterpret_cast<unsigned*>(S.pc);
        const int nw = L.k * (int)(sizeof(LfdPairConst) / 4);
        for (int i = tid; i < nw; i += (int)blockDim.x) dst[i] = src[i];
        const unsigned* ssrc = reinterpret_cast<const unsigned*>(L.slots + (size_t)r * L.k);
        unsigned* sdst = reinterpret_cast<unsigned*>(S.slot);
        const int nsw = L.k * (int)(sizeof(LfdSlotDesc) / 4);
        for (int i = tid; i < nsw; i += (int)blockDim.x) sdst[i] = ssrc[i];
        const unsigned* rsrc = reinterpret_cast<const unsigned*>(L.ref_const + r);
        unsigned* rdst = reinterpret_cast<unsigned*>(&S.rc);
        if (tid < (int)(sizeof(LfdRefConst) / 4)) rdst[tid] = rsrc[tid];
        const unsigned* dsrc = reinterpret_cast<const unsigned*>(L.refs + r);
        unsigned* ddst = reinterpret_cast<unsigned*>(&S.ref);
        if (tid >= 64 && tid < 64 + (int)(sizeof(LfdRefDesc) / 4)) ddst[tid - 64] = dsrc[tid - 64];
    }
    __syncthreads();
}

// cell -> (row, column) without an integer division: float estimate + one-step correction (exact for
// cell < 2^24 * ... any grid this library accepts: cell < 2^31, W < 2^16)
__device__ __forceinline__ void lfd_divmod(int cell, int W, float inv_w, int& y, int& x) {
    int q = (int)((float)cell * inv_w);
    int r = cell - q * W;
    if (r < 0) { --q; r += W; }
    if (r >= W) { ++q; r -= W; }
    if (r < 0) { --q; r += W; }
    if (r >= W) { ++q; r -= W; }
    y = q; x = r;
}

// certainty of one slot at one cell after the prologue of core/pipeline.py:407-430
__device__ __forceinline__ float cell_cert(const LfdLaunch& L, const BlockShared& S, int j, int cell, int x, int y,
                                           float raw, float mask_a_val) {
    float c = lfd_cert_floor(raw, L.kp.certainty_thresh);
    if (S.ref.mask_a) c = c * mask_a_val;
    const uint8_t* mb = S.slot[j].mask_b;
    if (mb) {
        const float* wp = S.slot[j].warp + (size_t)cell * L.warp_channels + (L.warp_channels - 2);
        const int ix = lfd_grid_nearest(wp[0], L.W);
        const int iy = lfd_grid_nearest(wp[1], L.H);
        float m = 0.0f;
        if (ix >= 0 && iy >= 0)
            m = (float)mb[(size_t)lfd_nearest_src(iy, L.mask_sy, L.h_match) * L.w_match + lfd_nearest_src(ix, L.mask_sx, L.w_match)];
        c = c * m;
    }
    return c;
}

__device__ __forceinline__ float cell_mask_a(const LfdLaunch& L, const BlockShared& S, int x, int y) {
    if (!S.ref.mask_a) return 1.0f;
    return (float)S.ref.mask_a[(size_t)lfd_nearest_src(y, L.mask_sy, L.h_match) * L.w_match + lfd_nearest_src(x, L.mask_sx, L.w_match)];
}

// torch.max(dim=0): first maximum wins, a NaN beats any number (first NaN)
__device__ __forceinline__ void argmax_step(float c, int j, float& best, int& bj) {
    const bool take = (c > best) || ((c != c) && !(best != best));
    if (take) { best = c; bj = j; }
}

__device__ __forceinline__ void cell_best(const LfdLaunch& L, const BlockShared& S, int cell, float& best, int& bj) {
    int y, x;
    lfd_divmod(cell, L.W, L.inv_w, y, x);
    const float ma = cell_mask_a(L, S, x, y);
    const int ns = S.ref.n_slots;
    best = cell_cert(L, S, 0, cell, x, y, S.slot[0].cert[cell], ma);
    bj = 0;
    for (int j = 1; j < ns; ++j) argmax_step(cell_cert(L, S, j, cell, x, y, S.slot[j].cert[cell], ma), j, best, bj);
}

// winner's warp -> normalised coordinates of the correspondence
__device__ __forceinline__ void cell_coords(const LfdLaunch& L, const BlockShared& S, int cell, int bj, float& xan,
                                            float& yan, float& xbn, float& ybn) {
    const float* wp = S.slot[bj].warp;
    if (L.warp_channels == 4) {
        const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)cell * 4);
        xan = v.x; yan = v.y; xbn = v.z; ybn = v.w;
    } else {
        const float2 v = *reinterpret_cast<const float2*>(wp + (size_t)cell * 2);
        int y, x;
        lfd_divmod(cell, L.W, L.inv_w, y, x);
        xan = L.axis_x[x]; yan = L.axis_y[y];
        xbn = v.x; ybn = v.y;
    }
}

}  // namespace

// =================================================================================================
// F5: per-(reference, neighbour) constants, once per batch (skipped when the batch is unchanged)
// =================================================================================================
extern "C" __global__ void lfd_pair_setup_kernel(LfdLaunch L, LfdRefConst* __restrict__ ref_out,
                                                 LfdPairConst* __restrict__ pair_out, LfdFastRef* __restrict__ fast_out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int n_pairs = L.n_refs * L.k;
    if (i < n_pairs) {
        const int r = i / L.k, j = i - r * L.k;
        if (j < L.refs[r].n_slots)
            lfd_make_pair_const(L.cams[L.refs[r].cam], L.cams[L.slots[i].cam], L.slots[i].cam, L.w_match, L.h_match, pair_out[i]);
    } else if (i < n_pairs + L.n_refs) {
        const int r = i - n_pairs;
        lfd_make_ref_const(L.cams[L.refs[r].cam], L.w_match, L.h_match, ref_out[r]);
        if (fast_out) {           // k <= 4: the record the persistent kernel reads with scalar loads
            LfdFastRef f;
            const int ns = L.refs[r].n_slots;
            for (int j = 0; j < 4; ++j) {
                const int jj = (j < ns) ? j : 0;
                f.cert[j] = L.slots[(size_t)r * L.k + jj].cert;
                f.warp[j] = L.slots[(size_t)r * L.k + jj].warp;
            }
            f.image = L.refs[r].image;
            f.n_slots = ns;
            f.pad = 0;
            lfd_make_ref_const(L.cams[L.refs[r].cam], L.w_match, L.h_match, f.rc);
            fast_out[r] = f;
        }
    }
}

// =================================================================================================
// P1 + F1: aggregate
// =================================================================================================
extern "C" __global__ void __launch_bounds__(256) lfd_aggregate_kernel(LfdLaunch L, float* __restrict__ best_cert,
                                                                       uint8_t* __restrict__ best_slot) {
    __shared__ BlockShared S;
    const int r = (int)blockIdx.y;
    const int tid = (int)threadIdx.x;
    if (tid == 0) S.ref = L.refs[r];
    if (tid < L.k) S.slot[tid] = L.slots[(size_t)r * L.k + tid];
    __syncthreads();
    const int HW = L.H * L.W;
    const bool plain = !S.ref.mask_a && (HW & 3) == 0;
    bool any_mask_b = false;
    for (int j = 0; j < S.ref.n_slots; ++j) any_mask_b |= (S.slot[j].mask_b != nullptr);
    const int ns = S.ref.n_slots;
    for (int base = ((int)blockIdx.x * 256 + tid) * 4; base < HW; base += (int)gridDim.x * 256 * 4) {
        if (plain && !any_mask_b && base + 3 < HW) {
            float4 best = *reinterpret_cast<const float4*>(S.slot[0].cert + base);
            const float th = L.kp.certainty_thresh;
            best.x = lfd_cert_floor(best.x, th); best.y = lfd_cert_floor(best.y, th);
            best.z = lfd_cert_floor(best.z, th); best.w = lfd_cert_floor(best.w, th);
            int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            for (int j = 1; j < ns; ++j) {
                const float4 c = *reinterpret_cast<const float4*>(S.slot[j].cert + base);
                argmax_step(lfd_cert_floor(c.x, th), j, best.x, b0);
                argmax_step(lfd_cert_floor(c.y, th), j, best.y, b1);
                argmax_step(lfd_cert_floor(c.z, th), j, best.z, b2);
                argmax_step(lfd_cert_floor(c.w, th), j, best.w, b3);
            }
            *reinterpret_cast<float4*>(best_cert + (size_t)r * HW + base) = best;
            if (best_slot) {
                const unsigned packed = (unsigned)b0 | ((unsigned)b1 << 8) | ((unsigned)b2 << 16) | ((unsigned)b3 << 24);
                *reinterpret_cast<unsigned*>(best_slot + (size_t)r * HW + base) = packed;
            }
        } else {
            for (int e = 0; e < 4 && base + e < HW; ++e) {
                float best; int bj;
                cell_best(L, S, base + e, best, bj);
                best_cert[(size_t)r * HW + base + e] = best;
                if (best_slot) best_slot[(size_t)r * HW + base + e] = (uint8_t)bj;
            }
        }
    }
}

// ---- wide look-back (both dense kernels) ------------------------------------------------------------
// A window of LFD_LB_ROWS x 64 predecessors is read per round trip.  The first window's loads are issued
// early (lookback_issue) so that their latency is covered by the work issued after them; lookback_finish
// consumes them and only falls back to polling when a needed predecessor has not published yet.
#ifndef LFD_LB_ROWS
#define LFD_LB_ROWS 4
#endif
struct LookbackWindow { u64 s[LFD_LB_ROWS]; };

__device__ __forceinline__ void lookback_issue(const LfdLaunch& L, unsigned tile, LookbackWindow& w) {
    const int lane = lane_id();
    const long long base = (long long)tile - 1;
#pragma unroll
    for (int q = 0; q < LFD_LB_ROWS; ++q) {
        const long long j = base - (long long)(q * 64 + lane);
        w.s[q] = state_load(L.tile_state + (j >= 0 ? j : 0));
        if (j < 0) w.s[q] = pack_state(kStPrefix, L.epoch, 0);   // virtual tiles < 0: prefix 0
    }
}

// returns true when the window settled the prefix (done) or was fully consumed (continue further back);
// false when a needed predecessor is still empty
__device__ __forceinline__ bool lookback_consume(const LookbackWindow& w, unsigned epoch, u64& acc, bool& done) {
    const int lane = lane_id();
    int qp = LFD_LB_ROWS, first = 64;
    bool blocked = false;
#pragma unroll
    for (int q = 0; q < LFD_LB_ROWS; ++q) {
        if (qp == LFD_LB_ROWS && !blocked) {
            const u64 st = state_status(w.s[q], epoch);
            const u64 pm = __ballot(st == kStPrefix);
            const u64 em = __ballot(st == kStEmpty);
            if (pm) {
                const int f = __ffsll((long long)pm) - 1;
                if (em & ((f == 0) ? 0ull : (~0ull >> (64 - f)))) blocked = true;
                else { qp = q; first = f; }
            } else if (em) {
                blocked = true;
            }
        }
    }
    if (blocked) return false;
#pragma unroll
    for (int q = 0; q < LFD_LB_ROWS; ++q)
        if (q < qp || (q == qp && lane <= first)) acc += w.s[q] & kValueMask;
    done = qp < LFD_LB_ROWS;
    return true;
}

__device__ __forceinline__ u64 lookback_finish(const LfdLaunch& L, unsigned tile, u64 my_total, LookbackWindow& w) {
    const unsigned epoch = L.epoch;
    const int lane = lane_id();
    if (tile == 0) return 0;           // published as a prefix right away
    u64 acc = 0;                       // per-lane partial sum of the aggregates taken so far
    unsigned base_tile = tile;         // the window in w covers [base_tile-1 ... base_tile-64*ROWS]
    unsigned spins = 0;
    bool done = false;
    for (;;) {
        if (lookback_consume(w, epoch, acc, done)) {
            if (done) break;
            base_tile -= 64 * LFD_LB_ROWS;
        } else {                       // a needed predecessor has not published yet: poll again
            if (++spins > LFD_SPIN_LIMIT) {
                if (lane == 0) atomicExch(L.status, LFD_LAUNCH_TIMEOUT);
                return 0;
            }
            // back off: every poll is LFD_LB_ROWS x 512 B of uncached traffic, and hundreds of tiles may be polling
            if (spins < 4) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);
        }
        lookback_issue(L, base_tile, w);
    }
    const u64 excl = wave_sum_u64(acc);
    if (lane == 0) state_store(L.tile_state + tile, pack_state(kStPrefix, epoch, excl + my_total));
    return excl;
}

// publish + resolve in one go (ticketed kernel: the count is published when the look-back starts)
__device__ __forceinline__ u64 lookback_exclusive_wide(const LfdLaunch& L, unsigned tile, u64 my_total) {
    if (lane_id() == 0) state_store(L.tile_state + tile, pack_state(tile == 0 ? kStPrefix : kStAggregate, L.epoch, my_total));
    if (tile == 0) return 0;
    LookbackWindow w;
    lookback_issue(L, tile, w);
    return lookback_finish(L, tile, my_total, w);
}

// =================================================================================================
// fused dense kernel
// =================================================================================================
// One 1024-cell tile per workgroup; tiles are numbered by an atomic ticket so that every tile a
// workgroup can wait for in the look-back belongs to a workgroup that is already running.  (A
// persistent-grid variant - one ticket per workgroup, tiles b, b+G, ... - was measured slower: with
// only a few resident workgroups per CU their phases line up and the look-back / barrier bubbles are
// no longer covered by other workgroups' arithmetic.)  The spin is bounded and reports
// LFD_LAUNCH_TIMEOUT instead of hanging.
//
// A thread owns 4 consecutive cells.  Their inputs, and later their outputs, are parked in the
// thread's own LDS slots, so the geometry loop carries no per-cell register arrays; survivors are
// then copied out through an order map with coalesced 16-byte stores.
struct DenseStage {                // per-tile results, indexed by the cell's slot inside the tile
    float xyz[3 * kTile];
    float pxy[2 * kTile];          // reference position in match pixels: colours are sampled at copy-out
    float err[kTile];
    unsigned short order[kTile];   // order[i] = tile slot of the i-th survivor (raster order)
    unsigned char slot[kTile];
};

struct __attribute__((packed, aligned(4))) LfdF3 { float a, b, c; };

extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_kernel(LfdLaunch L) {
    __shared__ BlockShared S;
    __shared__ DenseStage stage;
    __shared__ unsigned s_ticket;
    __shared__ unsigned s_wave_cnt[kBlock / 64];
    __shared__ unsigned s_slot_cnt[LFD_MAX_SLOTS];
    __shared__ u64 s_tile_excl;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // Tickets: a workgroup learns its tile from an atomic counter, so every tile a look-back can wait for is
    // already being worked on.  Returning atomics on ONE address complete at ~12 ns each on MI355X
    // (profiles/microbench/latency.hip: 16384 tickets = 0.2 ms), so the counter is split in LFD_TICKET_LANES
    // interleaved sequences on separate cache lines: workgroup b draws from sequence b % LANES, whose k-th
    // ticket is tile k*LANES + b % LANES.  Workgroups are dealt round-robin to the 8 XCDs, so each sequence
    // is served by one XCD and always has resident workgroups.
    const unsigned seq = blockIdx.x % LFD_TICKET_LANES;
    if (tid == 0) {
        const unsigned long long k = atomicAdd(L.ticket_lanes + (size_t)seq * 16, 1ull) - L.ticket_base_lane[seq];
        s_ticket = (unsigned)k * LFD_TICKET_LANES + seq;
    }
    __syncthreads();
    const unsigned tile = s_ticket;
    const unsigned n_tiles = (unsigned)L.n_refs * (unsigned)L.tiles_per_ref;
    const int HW = L.H * L.W;
    if (tile < n_tiles) {
        const int r = (int)(tile / (unsigned)L.tiles_per_ref);
        const int tile_in_ref = (int)(tile - (unsigned)r * (unsigned)L.tiles_per_ref);
        if (tid < LFD_MAX_SLOTS) s_slot_cnt[tid] = 0;
        block_prologue(L, r, S);                 // ends with a barrier (also covers s_slot_cnt)
        const int ns = S.ref.n_slots;
        bool any_mask = S.ref.mask_a != nullptr;
        for (int j = 0; j < ns; ++j) any_mask |= (S.slot[j].mask_b != nullptr);
        const int tile_cell0 = tile_in_ref * kTile;
        const int cell0 = tile_cell0 + tid * kCpt;

        // ---- stage 1: certainty floor + arg-max over the neighbour slots (coalesced 16-B loads) -----
        int bj[kCpt];
        if (!any_mask && (HW & 3) == 0 && kCpt == 4 && cell0 + 3 < HW) {
            const float th = L.kp.certainty_thresh;
            float4 best = *reinterpret_cast<const float4*>(S.slot[0].cert + cell0);
            best.x = lfd_cert_floor(best.x, th); best.y = lfd_cert_floor(best.y, th);
            best.z = lfd_cert_floor(best.z, th); best.w = lfd_cert_floor(best.w, th);
            bj[0] = bj[1] = bj[2] = bj[3] = 0;
            for (int j = 1; j < ns; ++j) {
                const float4 c = *reinterpret_cast<const float4*>(S.slot[j].cert + cell0);
                argmax_step(lfd_cert_floor(c.x, th), j, best.x, bj[0]);
                argmax_step(lfd_cert_floor(c.y, th), j, best.y, bj[1]);
                argmax_step(lfd_cert_floor(c.z, th), j, best.z, bj[2]);
                argmax_step(lfd_cert_floor(c.w, th), j, best.w, bj[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < kCpt; ++e) {
                bj[e] = 0;
                if (cell0 + e < HW) { float b; cell_best(L, S, cell0 + e, b, bj[e]); }
            }
        }

        // ---- stage 2: winner's warp (8 or 16 B per cell), all four loads in flight together; the
        //      coordinates are parked in this thread's own LDS slots (the slots later receive the cell's
        //      outputs), so the geometry loop below carries no per-cell register arrays ------------------
        unsigned bj_packed = 0;                   // 4 x 8 bits: the winning slot of each of this thread's cells
        if (kCpt == 4 && (L.W & 3) == 0 && cell0 + 3 < HW) {
            // W % 4 == 0: the four cells share a row, so one cell -> (row, column) conversion serves all
            float xa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ya = 0.0f;
            if (L.warp_channels != 4) {
                int y0, x0;
                lfd_divmod(cell0, L.W, L.inv_w, y0, x0);
                const float4 ax = *reinterpret_cast<const float4*>(L.axis_x + x0);
                xa[0] = ax.x; xa[1] = ax.y; xa[2] = ax.z; xa[3] = ax.w;
                ya = L.axis_y[y0];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* wp = S.slot[bj[e]].warp;
                float xan, yan, xbn, ybn;
                if (L.warp_channels == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)(unsigned)(cell0 + e) * 4);
                    xan = v.x; yan = v.y; xbn = v.z; ybn = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(wp + (size_t)(unsigned)(cell0 + e) * 2);
                    xan = xa[e]; yan = ya; xbn = v.x; ybn = v.y;
                }
                const int sl = tid * kCpt + e;
                stage.xyz[3 * sl + 0] = xan; stage.xyz[3 * sl + 1] = yan; stage.xyz[3 * sl + 2] = xbn;
                stage.err[sl] = ybn;
                bj_packed |= (unsigned)bj[e] << (8 * e);
            }
        } else {
#pragma unroll
            for (int e = 0; e < kCpt; ++e) {
                float xan = 0.0f, yan = 0.0f, xbn = 0.0f, ybn = 0.0f;
                if (cell0 + e < HW) cell_coords(L, S, cell0 + e, bj[e], xan, yan, xbn, ybn);
                const int sl = tid * kCpt + e;
                stage.xyz[3 * sl + 0] = xan; stage.xyz[3 * sl + 1] = yan; stage.xyz[3 * sl + 2] = xbn;
                stage.err[sl] = ybn;
                bj_packed |= (unsigned)bj[e] << (8 * e);
            }
        }
        *reinterpret_cast<unsigned*>(&stage.slot[tid * kCpt]) = bj_packed;

        // ---- stage 3: per-correspondence geometry + colour; survivors overwrite their slot -------------
        unsigned keep_bits = 0;
#pragma unroll 1
        for (int e = 0; e < kCpt; ++e) {
            // re-read the camera constants from LDS every cell instead of pinning ~60 registers on them
            asm volatile("" ::: "memory");
            const int sl = tid * kCpt + e;
            const float xan = stage.xyz[3 * sl + 0], yan = stage.xyz[3 * sl + 1], xbn = stage.xyz[3 * sl + 2];
            const float ybn = stage.err[sl];
            const int bje = (int)((bj_packed >> (8 * e)) & 0xffu);
            LfdCellResult res;
            res.keep = 0; res.x = res.y = res.z = res.err = res.xa_px = res.ya_px = 0.0f;
#if defined(LFD_ABLATE_EVAL)
            if (cell0 + e < HW) { res.keep = xbn > -0.9f; res.x = xan; res.y = yan; res.z = xbn; res.err = ybn; res.xa_px = 1.0f; res.ya_px = 1.0f; }
#else
            if (cell0 + e < HW) lfd_eval_correspondence(S.rc, S.pc[bje], xan, yan, xbn, ybn, L.kp, res);
#endif
            if (res.keep) {
                stage.xyz[3 * sl + 0] = res.x; stage.xyz[3 * sl + 1] = res.y; stage.xyz[3 * sl + 2] = res.z;
                stage.pxy[2 * sl + 0] = res.xa_px; stage.pxy[2 * sl + 1] = res.ya_px;
                stage.err[sl] = res.err;
                keep_bits |= 1u << e;
            }
        }
        // survivors per neighbour slot (outside the divergent loop): ballots over the kept cells of each slot
        if (L.seg_counts) {
            for (int j = 0; j < ns; ++j) {
                unsigned c = 0;
#pragma unroll
                for (int e = 0; e < kCpt; ++e)
                    c += (unsigned)__popcll(__ballot(((keep_bits >> e) & 1u) && ((bj_packed >> (8 * e)) & 0xffu) == (unsigned)j));
                if (lane == 0 && c) atomicAdd(&s_slot_cnt[j], c);
            }
        }

        // ---- stage 4: ordered compaction: thread -> wave -> workgroup -> grid (look-back) ----------------
        const unsigned my_cnt = __popc(keep_bits);
        unsigned incl = my_cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned n = __shfl_up(incl, off, 64);
            if (lane >= off) incl += n;
        }
        if (lane == 63) s_wave_cnt[wave] = incl;
        __syncthreads();                          // s_wave_cnt written
        unsigned wave_off = 0, block_total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            if (w < wave) wave_off += s_wave_cnt[w];
            block_total += s_wave_cnt[w];
        }
        {
            unsigned lpos = wave_off + (incl - my_cnt);
#pragma unroll
            for (int e = 0; e < kCpt; ++e)
                if ((keep_bits >> e) & 1u) stage.order[lpos++] = (unsigned short)(tid * kCpt + e);
        }
        if (wave == 0) {
#if defined(LFD_ABLATE_LOOKBACK)
            const u64 excl = (u64)tile * kTile;
#else
            const u64 excl = lookback_exclusive(L, tile, block_total);    // 64-wide window: measured faster here than 256
#endif
            if (lane == 0) {
                s_tile_excl = excl;
                if (tile_in_ref == 0) L.ref_offsets[r] = (long long)excl;
                if (tile == n_tiles - 1u) L.ref_offsets[L.n_refs] = (long long)(excl + block_total);
            }
        }
        __syncthreads();                          // staging complete, prefix known
        if (L.seg_counts && tid < ns && s_slot_cnt[tid]) atomicAdd(&L.seg_counts[(size_t)r * L.k + tid], (int)s_slot_cnt[tid]);

        // ---- stage 5: coalesced copy-out ----------------------------------------------------------------
        {
            const long long base = (long long)s_tile_excl;
            long long room = L.capacity - base;          // beyond capacity: counted, not written
            int n = (int)block_total;
            if (room < (long long)n) n = room > 0 ? (int)room : 0;
#if !defined(LFD_ABLATE_STORES)
            // one survivor record per thread: consecutive threads write consecutive records, so every
            // wave-wide store covers one contiguous span of the output arrays
            LfdF3* gx = reinterpret_cast<LfdF3*>(L.xyz + 3 * base);
            LfdF3* gc = reinterpret_cast<LfdF3*>(L.rgb + 3 * base);
            float* ge = L.err + base;
            // two records per thread per step: the image rows of both are in flight before either colour is evaluated
            const uint8_t* image = S.ref.image;
            for (int i0 = tid; i0 < n; i0 += 2 * kBlock) {
                int sl[2];
                float px[2], py[2];
                LfdTapRows taps[2];
                unsigned sh0[2], sh1[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * kBlock;
                    sl[u] = (int)stage.order[i < n ? i : n - 1];
                    px[u] = stage.pxy[2 * sl[u] + 0]; py[u] = stage.pxy[2 * sl[u] + 1];
                    taps[u] = lfd_bilinear_fetch(image, L.w_match, L.h_match, px[u], py[u], sh0[u], sh1[u]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * kBlock;
                    float rgb[3];
#if defined(LFD_ABLATE_COLOUR)
                    rgb[0] = px[u]; rgb[1] = py[u]; rgb[2] = (float)(taps[u].r0 + taps[u].r1 + sh0[u] + sh1[u]);
#else
                    lfd_bilinear_eval(taps[u], sh0[u], sh1[u], L.w_match, L.h_match, px[u], py[u], rgb);
#endif
                    if (i < n) {
                        LfdF3 p, c;
                        p.a = stage.xyz[3 * sl[u] + 0]; p.b = stage.xyz[3 * sl[u] + 1]; p.c = stage.xyz[3 * sl[u] + 2];
                        c.a = rgb[0]; c.b = rgb[1]; c.c = rgb[2];
                        gx[i] = p;
                        gc[i] = c;
                        ge[i] = stage.err[sl[u]];
                        if (L.cell) L.cell[base + i] = tile_cell0 + sl[u];
                        if (L.slot) L.slot[base + i] = stage.slot[sl[u]];
                    }
                }
            }
#endif
        }
    }
}

// =================================================================================================
// fused dense kernel, fast path: persistent, software-pipelined (no masks, k <= 4)
// =================================================================================================
// Measured on MI355X (profiles/r1/ablation.txt): with one tile per workgroup the kernel is bound by
// the chain of dependent round trips every tile pays (ticket -> descriptors -> certainty -> warp ->
// look-back), not by arithmetic or bandwidth.  This variant removes the chain from the critical path:
//   * persistent grid, tiles dealt statically (tile = block + i*grid): no ticket; every workgroup the
//     look-back can wait for is resident because the grid is sized from the occupancy of the kernel;
//   * while tile n is being evaluated, the certainty planes and per-pair constants of tile n+1 are
//     already in flight (registers -> LDS double buffer);
//   * the look-back reads a window of 256 predecessors per round trip instead of 64;
//   * a wave owns 256 consecutive cells, 64 per step, so survivors are compacted with one ballot per
//     step straight into the wave's staging area (raster order, no order map) and copied out linearly.
#if defined(LFD_PHASE_TIMING)
__device__ unsigned long long lfd_phase_acc[16];
#ifndef LFD_PHASE_MASK
#define LFD_PHASE_MASK 0xfff
#endif
#define LFD_PHASE_MARK(idx) do { if ((LFD_PHASE_MASK >> (idx)) & 1) { const long long t_ = (long long)wall_clock64(); t_acc[idx] += (unsigned)(t_ - t_phase); t_phase = t_; } } while (0)
#else
#define LFD_PHASE_MARK(idx) do { } while (0)
#endif

#define LFD_CONST_AS __attribute__((address_space(4)))
// Loads through a constant-address-space pointer with a uniform address are selected as scalar loads
// (s_load_*): the per-tile descriptors then cost no vector-memory round trip and land in SGPRs.
template <class T>
__device__ __forceinline__ const T LFD_CONST_AS* lfd_const_as(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const T LFD_CONST_AS*)p;
#pragma clang diagnostic pop
}

constexpr int kFastWaves = LFD_DENSE_FAST_THREADS / 64;    // waves per workgroup of the persistent kernel
constexpr int kFastTile = LFD_DENSE_FAST_TILE;              // cells per tile (256 per wave)
constexpr int kFastThreads = LFD_DENSE_FAST_THREADS;

template <int K>
struct FastShared {
    LfdPairConst pc[2][K];                     // [constants buffer][slot]
    // survivors of a tile wait in LDS for one tile period (deferred look-back): two staging buffers.
    // Colours are not staged: they are sampled when the records are copied out.
    float xyz[2][kFastWaves][3 * 256];         // [staging buffer][wave][record]
    float err[2][kFastWaves][256];
    unsigned char cellq[2][kFastWaves][256];   // survivor -> cell inside the wave's 256-cell chunk
    unsigned char slot[2][kFastWaves][256];
    unsigned slot_cnt[2][K];
    unsigned wave_cnt[2][kFastWaves];
    unsigned claim[4];                         // ring of claimed tiles (0xffffffff: sequence exhausted)
    u64 tile_excl;
};

// MODE 0: warp = [xB,yB], default A-grid axes (closed form); 1: warp = [xB,yB], axes given by the caller;
//      2: warp = [xA,yA,xB,yB]
//
// Persistent, software-pipelined dense kernel.  What the one-tile-per-workgroup kernel pays in sequence for
// every tile (ticket -> descriptors -> certainty -> warp -> geometry -> look-back -> copy-out) is spread over
// three consecutive tiles of a resident workgroup:
//   tile n+2  claimed (ticket in flight, nobody waits for it)
//   tile n+1  certainty planes and per-pair constants in flight (registers -> LDS double buffer)
//   tile n    arg-max, warp, geometry; survivors compacted per wave into staging buffer n&1; count published
//   tile n-1  retired: prefix resolved by a look-back whose loads were issued before anything else of this
//             iteration, colours sampled, records copied out of staging buffer (n-1)&1
// Tickets come from LFD_TICKET_LANES interleaved counters (see lfd_dense_kernel); a claimed tile is always
// evaluated without waiting for anything, so every count a look-back polls for is on its way.
template <int K, int MODE>
__device__ __forceinline__ void lfd_dense_fast_body(const LfdLaunch& L, FastShared<K>& sm) {
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tpr = (unsigned)L.tiles_per_ref;
    const unsigned n_tiles = (unsigned)L.n_refs * tpr;
    const int HW = L.H * L.W;
    const float th = L.kp.certainty_thresh;
    constexpr int kConstWords = 36 * K;          // LfdPairConst = 36 dwords
    constexpr unsigned kNone = 0xffffffffu;
    static_assert(sizeof(LfdPairConst) == 144, "LfdPairConst layout");
    static_assert(kConstWords <= kFastThreads, "one prefetched word per thread");
    const LfdFastRef LFD_CONST_AS* fast = lfd_const_as(L.fast);
    const unsigned seq = blockIdx.x % LFD_TICKET_LANES;

    auto claim = [&]() -> unsigned {            // lane 0 of wave 0 only
        const unsigned long long k = atomicAdd(L.ticket_lanes + (size_t)seq * 16, 1ull) - L.ticket_base_lane[seq];
        const unsigned long long t = k * LFD_TICKET_LANES + seq;
        return t < (unsigned long long)n_tiles ? (unsigned)t : kNone;
    };
    auto split = [&](unsigned t, unsigned& rr, unsigned& tt) {       // tile -> (reference, tile in reference), uniform
        const unsigned q = t / tpr;
        rr = __builtin_amdgcn_readfirstlane(q);
        tt = __builtin_amdgcn_readfirstlane(t - q * tpr);
    };

    // one word of the per-pair constants per thread (unconditional load, predicated LDS store)
    const int cw = tid < kConstWords ? tid : 0;
    auto fetch_const_word = [&](unsigned rr) -> unsigned {
        return reinterpret_cast<const unsigned*>(L.pair_const + (size_t)rr * K)[cw];
    };
    auto store_const_word = [&](int b, unsigned v) {
        if (tid < kConstWords) reinterpret_cast<unsigned*>(sm.pc[b])[tid] = v;
    };

    float cn[K][4];                       // raw certainty of this thread's 4 cells, all slots
    auto fetch_cert = [&](unsigned rr, unsigned tt) {
        const int c0 = (int)tt * kFastTile + wave * 256 + lane;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float* cp = fast[rr].cert[j];       // slots >= n_slots alias slot 0: always loadable
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cell = c0 + 64 * e;
                cn[j][e] = cp[cell < HW ? cell : HW - 1];
            }
        }
    };

    // ---- pipeline prologue: claim three tiles, constants and certainties of the first ------------------
    if (tid == 0) { sm.claim[0] = claim(); sm.claim[1] = claim(); sm.claim[2] = claim(); }
    if (tid < 2 * K) { sm.slot_cnt[0][tid % K] = 0; sm.slot_cnt[1][tid % K] = 0; }
    __syncthreads();
    unsigned tile = sm.claim[0];
    if (tile == kNone) return;                // (uniform) nothing left for this workgroup
    unsigned r, tin;
    split(tile, r, tin);
    store_const_word(0, fetch_const_word(r));
    fetch_cert(r, tin);
    __syncthreads();
    int buf = 0;
    unsigned it = 0;                          // iteration = position in the claim ring
    bool has_cur = true, have_prev = false;
    unsigned p_tile = 0, p_r = 0, p_tin = 0, p_total = 0, p_wave_off = 0, p_run = 0;

    while (has_cur || have_prev) {
        const unsigned next = has_cur ? sm.claim[(it + 1) & 3] : kNone;
        const bool has_next = next != kNone;
        unsigned r_next = r, tin_next = tin;
        if (has_next) split(next, r_next, tin_next);
        unsigned claimed = kNone;
        if (tid == 0 && has_next) claimed = claim();          // tile n+3's ticket: consumed at the end of the iteration
        const int ns = fast[r].n_slots;
        const int tile_cell0 = (int)tin * kFastTile;
        const int cell0 = tile_cell0 + wave * 256 + lane;     // + 64*e
        const int pbuf = buf ^ 1;

        // ---- retire (tile n-1), issue half: look-back window first, then the image rows of this wave's survivors ----
        LookbackWindow lbw;
        int rcell[4];
        float rxa[4], rya[4];
        LfdTapRows taps[4];
        unsigned sh0[4], sh1[4];
        if (have_prev) {
            if (wave == 0 && p_tile != 0) lookback_issue(L, p_tile, lbw);
            const uint8_t* image = fast[p_r].image;
            const int rchunk0 = (int)p_tin * kFastTile + wave * 256;
            const int nn = (int)p_run > 0 ? (int)p_run : 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = g * 64 + lane;
                const int ii = i < nn ? i : nn - 1;                 // clamped: loads stay unconditional
                rcell[g] = rchunk0 + (int)sm.cellq[pbuf][wave][ii];
                float xan, yan;
                if (MODE == 2) {
                    const int sj = (int)sm.slot[pbuf][wave][ii];
                    const float* wp = fast[p_r].warp[0];
#pragma unroll
                    for (int j = 1; j < K; ++j) wp = (sj == j) ? fast[p_r].warp[j] : wp;
                    const unsigned cc = (unsigned)(rcell[g] < HW ? rcell[g] : HW - 1);
                    const float2 v = *reinterpret_cast<const float2*>(wp + (size_t)cc * 4);
                    xan = v.x; yan = v.y;
                } else {
                    int y, x;
                    lfd_divmod(rcell[g] < HW ? rcell[g] : 0, L.W, L.inv_w, y, x);
                    if (MODE == 0) { xan = lfd_axis_value(L.ax, x); yan = lfd_axis_value(L.ay, y); }
                    else { xan = L.axis_x[x]; yan = L.axis_y[y]; }
                }
                rxa[g] = lfd_match_px(xan, L.kp.wm1); rya[g] = lfd_match_px(yan, L.kp.hm1);
                if (MODE == 0) taps[g] = lfd_bilinear_fetch(image, L.w_match, L.h_match, rxa[g], rya[g], sh0[g], sh1[g]);
            }
            if (MODE != 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) taps[g] = lfd_bilinear_fetch(image, L.w_match, L.h_match, rxa[g], rya[g], sh0[g], sh1[g]);
            }
        }

        // ---- stage 1 (tile n): certainty floor + arg-max over the slots; stage 2a: the winner's warp --------------
        unsigned bj_packed = 0;
        float wv[4][4];
        if (has_cur) {
            int bj[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float best = lfd_cert_floor(cn[0][e], th);
#pragma unroll
                for (int j = 1; j < K; ++j)
                    if (j < ns) argmax_step(lfd_cert_floor(cn[j][e], th), j, best, bj[e]);
                bj_packed |= (unsigned)bj[e] << (8 * e);
            }
            const float* wbase[K];
#pragma unroll
            for (int j = 0; j < K; ++j) wbase[j] = fast[r].warp[j];
            int y = 0, x = 0;
            if (MODE == 1) lfd_divmod(cell0 < HW ? cell0 : 0, L.W, L.inv_w, y, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cell = cell0 + 64 * e;
                const unsigned cc = (unsigned)(cell < HW ? cell : HW - 1);
                const float* wp = wbase[0];
#pragma unroll
                for (int j = 1; j < K; ++j) wp = (bj[e] == j) ? wbase[j] : wp;
                if (MODE == 2) {
                    const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)cc * 4);
                    wv[e][0] = v.x; wv[e][1] = v.y; wv[e][2] = v.z; wv[e][3] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(wp + (size_t)cc * 2);
                    wv[e][2] = v.x; wv[e][3] = v.y;
                    if (MODE == 1) {
                        wv[e][0] = L.axis_x[x]; wv[e][1] = L.axis_y[y];
                        x += 64;
                        while (x >= L.W) { x -= L.W; ++y; }
                    }
                }
            }
        }

        // ---- prefetch (tile n+1): per-pair constants (one word per thread) and certainty planes; issued last, ----
        //      consumed a whole tile period later
        unsigned pre = 0;
        if (has_next) {
            pre = fetch_const_word(r_next);
            fetch_cert(r_next, tin_next);
        }

        // ---- retire (tile n-1), finish half: prefix, colours, copy-out --------------------------------------------
        if (have_prev) {
            if (wave == 0) {
#if defined(LFD_ABLATE_LOOKBACK)
                const u64 excl = (u64)p_tile * kFastTile;
#else
                const u64 excl = lookback_finish(L, p_tile, p_total, lbw);
#endif
                if (lane == 0) {
                    sm.tile_excl = excl;
                    if (p_tin == 0) L.ref_offsets[p_r] = (long long)excl;
                    if (p_tile == n_tiles - 1u) L.ref_offsets[L.n_refs] = (long long)(excl + p_total);
                }
            }
            __syncthreads();                          // prefix known
            const long long base = (long long)sm.tile_excl + (long long)p_wave_off;
            long long room = L.capacity - base;          // beyond capacity: counted, not written
            int n = (int)p_run;
            if (room < (long long)n) n = room > 0 ? (int)room : 0;
            LfdF3* gx = reinterpret_cast<LfdF3*>(L.xyz + 3 * base);
            LfdF3* gc = reinterpret_cast<LfdF3*>(L.rgb + 3 * base);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = g * 64 + lane;
                float rgb[3];
#if defined(LFD_ABLATE_COLOUR)
                rgb[0] = rxa[g]; rgb[1] = rya[g]; rgb[2] = (float)(taps[g].r0 + taps[g].r1 + sh0[g] + sh1[g]);
#else
                lfd_bilinear_eval(taps[g], sh0[g], sh1[g], L.w_match, L.h_match, rxa[g], rya[g], rgb);
#endif
#if defined(LFD_ABLATE_STORES)
                if (i < n && rgb[0] == -12345.0f) {
#else
                if (i < n) {
#endif
                    LfdF3 c; c.a = rgb[0]; c.b = rgb[1]; c.c = rgb[2];
                    LfdF3 q; q.a = sm.xyz[pbuf][wave][3 * i + 0]; q.b = sm.xyz[pbuf][wave][3 * i + 1]; q.c = sm.xyz[pbuf][wave][3 * i + 2];
                    gx[i] = q;
                    gc[i] = c;
                    L.err[base + i] = sm.err[pbuf][wave][i];
                    if (L.cell) L.cell[base + i] = rcell[g];
                    if (L.slot) L.slot[base + i] = sm.slot[pbuf][wave][i];
                }
            }
            have_prev = false;
        }
        if (!has_cur) break;

        // ---- stage 2b (tile n): park the correspondences in the wave's staging slots of these cells --------------
        {
            int y = 0, x = 0;
            if (MODE == 0) lfd_divmod(cell0 < HW ? cell0 : 0, L.W, L.inv_w, y, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sl = e * 64 + lane;
                if (MODE == 0) {      // default axes: closed form, nothing was loaded for them
                    wv[e][0] = lfd_axis_value(L.ax, x); wv[e][1] = lfd_axis_value(L.ay, y);
                    x += 64;
                    while (x >= L.W) { x -= L.W; ++y; }
                }
                sm.xyz[buf][wave][3 * sl + 0] = wv[e][0]; sm.xyz[buf][wave][3 * sl + 1] = wv[e][1]; sm.xyz[buf][wave][3 * sl + 2] = wv[e][2];
                sm.err[buf][wave][sl] = wv[e][3];
            }
        }

        // ---- stage 3 (tile n): geometry, 64 cells of the wave per step, survivors compacted in place ---------
        LfdRefConst rc;
#pragma unroll
        for (int i = 0; i < 12; ++i) rc.P[i] = fast[r].rc.P[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) rc.C[i] = fast[r].rc.C[i];
        rc.sx = fast[r].rc.sx; rc.sy = fast[r].rc.sy; rc.pad = 0.0f;
        unsigned keep_bits = 0;
        unsigned run = 0;                       // survivors of this wave so far (uniform)
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
            asm volatile("" ::: "memory");      // per-pair constants are re-read from LDS per step instead of pinned in registers
            const int sl = e * 64 + lane;
            const float xan = sm.xyz[buf][wave][3 * sl + 0], yan = sm.xyz[buf][wave][3 * sl + 1], xbn = sm.xyz[buf][wave][3 * sl + 2];
            const float ybn = sm.err[buf][wave][sl];
            const int bje = (int)((bj_packed >> (8 * e)) & 0xffu);
            LfdCellResult res;
            res.keep = 0; res.x = res.y = res.z = res.err = res.xa_px = res.ya_px = 0.0f;
#if defined(LFD_ABLATE_EVAL)
            if (cell0 + 64 * e < HW) { res.keep = xbn > -0.9f; res.x = xan + rc.P[0]; res.y = yan + sm.pc[buf][bje].P[1]; res.z = xbn; res.err = ybn; }
#else
            if (cell0 + 64 * e < HW) lfd_eval_correspondence(rc, sm.pc[buf][bje], xan, yan, xbn, ybn, L.kp, res);
#endif
            const u64 km = __ballot(res.keep != 0);
            if (res.keep) {
                const unsigned pos = run + __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                sm.xyz[buf][wave][3 * pos + 0] = res.x; sm.xyz[buf][wave][3 * pos + 1] = res.y; sm.xyz[buf][wave][3 * pos + 2] = res.z;
                sm.err[buf][wave][pos] = res.err;
                sm.cellq[buf][wave][pos] = (unsigned char)sl;
                sm.slot[buf][wave][pos] = (unsigned char)bje;
                keep_bits |= 1u << e;
            }
            run += (unsigned)__popcll(km);
        }
        if (L.seg_counts) {
            for (int j = 0; j < ns; ++j) {
                unsigned c = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    c += (unsigned)__popcll(__ballot(((keep_bits >> e) & 1u) && ((bj_packed >> (8 * e)) & 0xffu) == (unsigned)j));
                if (lane == 0 && c) atomicAdd(&sm.slot_cnt[buf][j], c);
            }
        }
        if (lane == 0) sm.wave_cnt[buf][wave] = run;
        if (has_next) store_const_word(buf ^ 1, pre);
        if (tid == 0) sm.claim[(it + 3) & 3] = claimed;       // (kNone once the sequence is exhausted)
        __syncthreads();                          // wave counts, slot counts, next constants, next claim

        // ---- stage 4 (tile n): workgroup offsets; the count is published now, the prefix is resolved one tile later ----
        unsigned wave_off = 0, block_total = 0;
#pragma unroll
        for (int w = 0; w < kFastWaves; ++w) {
            const unsigned c = sm.wave_cnt[buf][w];
            if (w < wave) wave_off += c;
            block_total += c;
        }
        if (wave == 0) {
            if (lane == 0) state_store(L.tile_state + tile, pack_state(tile == 0 ? kStPrefix : kStAggregate, L.epoch, block_total));
            if (L.seg_counts && lane < ns) {
                const unsigned c = sm.slot_cnt[buf][lane];
                if (c) atomicAdd(&L.seg_counts[(size_t)r * K + lane], (int)c);
                sm.slot_cnt[buf][lane] = 0;
            }
        }
        have_prev = true;
        p_tile = tile; p_r = r; p_tin = tin; p_total = block_total; p_wave_off = wave_off; p_run = run;
        has_cur = has_next;
        tile = next; r = r_next; tin = tin_next; buf ^= 1; ++it;
    }
}

// the number of claims a launch makes depends on how the tiles fell to the workgroups, so the last workgroup
// to leave puts the ticket sequences back to zero for the next launch on the stream
__device__ __forceinline__ void lfd_fast_epilogue(const LfdLaunch& L) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(L.exit_count, 1u);
        if (done == gridDim.x - 1u) {
            for (int s = 0; s < LFD_TICKET_LANES; ++s)
                __hip_atomic_store(L.ticket_lanes + (size_t)s * 16, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(L.exit_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#define LFD_DENSE_FAST_KERNEL(KK, MM)                                                                          \
    extern "C" __global__ void __launch_bounds__(kFastThreads, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_fast_kernel_k##KK##_m##MM(LfdLaunch L) { \
        __shared__ FastShared<KK> sm;                                                                          \
        lfd_dense_fast_body<KK, MM>(L, sm);                                                                    \
        lfd_fast_epilogue(L);                                                                                  \
    }
LFD_DENSE_FAST_KERNEL(1, 0) LFD_DENSE_FAST_KERNEL(1, 1) LFD_DENSE_FAST_KERNEL(1, 2)
LFD_DENSE_FAST_KERNEL(2, 0) LFD_DENSE_FAST_KERNEL(2, 1) LFD_DENSE_FAST_KERNEL(2, 2)
LFD_DENSE_FAST_KERNEL(3, 0) LFD_DENSE_FAST_KERNEL(3, 1) LFD_DENSE_FAST_KERNEL(3, 2)
LFD_DENSE_FAST_KERNEL(4, 0) LFD_DENSE_FAST_KERNEL(4, 1) LFD_DENSE_FAST_KERNEL(4, 2)

// =================================================================================================
// upstream-equivalent indexed kernel: one workgroup per reference
// =================================================================================================
// Pass A evaluates every selected cell (results parked in a scratch area, keep/slot codes in LDS or
// scratch), counts survivors per slot and finds each slot's first appearance; the groups are then
// ordered by first appearance (core/pipeline.py:685-688) and pass B moves survivors to their final
// place: group after group, members in selection order.
extern "C" __global__ void __launch_bounds__(LFD_INDEXED_BLOCK) lfd_indexed_kernel(LfdLaunch L, const long long* __restrict__ sel_idx,
                                                                                   const long long* __restrict__ sel_offsets,
                                                                                   float* __restrict__ scratch, uint8_t* __restrict__ codes,
                                                                                   int32_t* __restrict__ seg_order) {
    __shared__ BlockShared S;
    __shared__ unsigned s_ticket;
    __shared__ unsigned s_cnt[LFD_MAX_SLOTS];          // survivors per slot
    __shared__ unsigned s_first[LFD_MAX_SLOTS];        // first selection position of each slot
    __shared__ unsigned s_start[LFD_MAX_SLOTS];        // output start of each slot's group
    __shared__ unsigned s_run[LFD_MAX_SLOTS];          // running count during pass B
    __shared__ unsigned s_wcnt[LFD_INDEXED_BLOCK / 64][LFD_MAX_SLOTS];
    __shared__ u64 s_excl;
    __shared__ unsigned s_total;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nwaves = LFD_INDEXED_BLOCK / 64;
    if (tid == 0) s_ticket = (unsigned)(atomicAdd(L.ticket, 1ull) - L.ticket_base);
    if (tid < LFD_MAX_SLOTS) { s_cnt[tid] = 0; s_first[tid] = 0xffffffffu; s_run[tid] = 0; }
    __syncthreads();
    const int r = (int)s_ticket;
    block_prologue(L, r, S);
    const int ns = S.ref.n_slots;
    const long long sel_begin = sel_offsets[r], sel_end = sel_offsets[r + 1];
    const int n_sel = (int)(sel_end - sel_begin);
    const int HW = L.H * L.W;

    // ---- pass A ---------------------------------------------------------------------------------
    for (int i = tid; i < n_sel; i += LFD_INDEXED_BLOCK) {
        const long long cl = sel_idx[sel_begin + i];
        unsigned code = 0xffu;                       // invalid selection index: dropped
        if (cl >= 0 && cl < HW) {
            const int cell = (int)cl;
            float best; int bj;
            cell_best(L, S, cell, best, bj);
            float xan, yan, xbn, ybn;
            cell_coords(L, S, cell, bj, xan, yan, xbn, ybn);
            LfdCellResult res;
            lfd_eval_correspondence(S.rc, S.pc[bj], xan, yan, xbn, ybn, L.kp, res);
            atomicMin(&s_first[bj], (unsigned)i);
            code = (unsigned)bj | (res.keep ? 0x80u : 0u);
            if (res.keep) {
                float rgb[3];
                lfd_bilinear_rgb(S.ref.image, L.w_match, L.h_match, res.xa_px, res.ya_px, 1.0f, 1.0f, rgb);
                float* o = scratch + (size_t)(sel_begin + i) * 8;
                o[0] = res.x; o[1] = res.y; o[2] = res.z; o[3] = res.err; o[4] = rgb[0]; o[5] = rgb[1]; o[6] = rgb[2];
                atomicAdd(&s_cnt[bj], 1u);
            }
        }
        codes[sel_begin + i] = (uint8_t)code;
    }
    __syncthreads();

    // ---- group order = slots sorted by first appearance; exclusive starts ---------------------------
    if (tid == 0) {
        unsigned total = 0;
        int g = 0;
        bool used[LFD_MAX_SLOTS];
        for (int j = 0; j < LFD_MAX_SLOTS; ++j) used[j] = false;
        for (int round = 0; round < ns; ++round) {
            int pick = -1; unsigned fp = 0xffffffffu;
            for (int j = 0; j < ns; ++j) if (!used[j] && s_first[j] < fp) { fp = s_first[j]; pick = j; }
            if (pick < 0) break;
            used[pick] = true;
            s_start[pick] = total;
            total += s_cnt[pick];
            if (s_cnt[pick]) { if (seg_order) seg_order[(size_t)r * L.k + g] = pick; ++g; }
        }
        if (seg_order) for (; g < L.k; ++g) seg_order[(size_t)r * L.k + g] = -1;
        s_total = total;
    }
    if (L.seg_counts && tid < L.k) L.seg_counts[(size_t)r * L.k + tid] = (tid < ns) ? (int)s_cnt[tid] : 0;
    __syncthreads();
    if (wave == 0) {
        const u64 excl = lookback_exclusive(L, (unsigned)r, s_total);
        if (lane == 0) {
            s_excl = excl;
            L.ref_offsets[r] = (long long)excl;
            if (r == L.n_refs - 1) L.ref_offsets[L.n_refs] = (long long)(excl + s_total);
        }
    }
    __syncthreads();

    // ---- pass B: stable scatter, rounds of one workgroup width in selection order ------------------------
    const long long out_base = (long long)s_excl;
    for (int round0 = 0; round0 < n_sel; round0 += LFD_INDEXED_BLOCK) {
        const int i = round0 + tid;
        unsigned code = 0xffu;
        if (i < n_sel) code = codes[sel_begin + i];
        const bool kept = (code != 0xffu) && (code & 0x80u);
        const int j = (int)(code & 0x7fu);
        unsigned rank_in_wave = 0;
        for (int jj = 0; jj < ns; ++jj) {
            const u64 m = __ballot(kept && j == jj);
            if (lane == 0) s_wcnt[wave][jj] = (unsigned)__popcll(m);
            if (kept && j == jj) rank_in_wave = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (kept) {
            unsigned before = s_run[j];
            for (int w = 0; w < wave; ++w) before += s_wcnt[w][j];
            const long long pos = out_base + s_start[j] + before + rank_in_wave;
            if (pos < L.capacity) {
                const float* o = scratch + (size_t)(sel_begin + i) * 8;
                L.xyz[pos * 3 + 0] = o[0]; L.xyz[pos * 3 + 1] = o[1]; L.xyz[pos * 3 + 2] = o[2];
                L.err[pos] = o[3];
                L.rgb[pos * 3 + 0] = o[4]; L.rgb[pos * 3 + 1] = o[5]; L.rgb[pos * 3 + 2] = o[6];
                if (L.cell) L.cell[pos] = (int32_t)sel_idx[sel_begin + i];
                if (L.slot) L.slot[pos] = (uint8_t)j;
            }
        }
        __syncthreads();
        if (tid < ns) {
            unsigned add = 0;
            for (int w = 0; w < nwaves; ++w) add += s_wcnt[w][tid];
            s_run[tid] += add;
        }
        __syncthreads();
    }
}
