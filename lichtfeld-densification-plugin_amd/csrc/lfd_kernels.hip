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
__device__ u64 lookback_exclusive(u64* state, unsigned epoch, unsigned tile, u64 my_total) {
    const int lane = lane_id();
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
    const int tid = (int)threadIdx.x;
    if (tid == 0) S.ref = L.refs[r];
    if (tid < L.k) S.slot[tid] = L.slots[(size_t)r * L.k + tid];
    __syncthreads();
    // per-pair constants (P, C, pixel scales, F) were derived once per batch by lfd_pair_setup_kernel;
    // stage this reference's rows in LDS with coalesced dword copies
    const int ns = S.ref.n_slots;
    {
        const unsigned* src = reinterpret_cast<const unsigned*>(L.pair_const + (size_t)r * L.k);
        unsigned* dst = reinterpret_cast<unsigned*>(S.pc);
        const int nw = ns * (int)(sizeof(LfdPairConst) / 4);
        for (int i = tid; i < nw; i += (int)blockDim.x) dst[i] = src[i];
        const unsigned* rsrc = reinterpret_cast<const unsigned*>(L.ref_const + r);
        unsigned* rdst = reinterpret_cast<unsigned*>(&S.rc);
        if (tid < (int)(sizeof(LfdRefConst) / 4)) rdst[tid] = rsrc[tid];
    }
    __syncthreads();
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
    const int y = cell / L.W, x = cell - y * L.W;
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
        const int y = cell / L.W, x = cell - y * L.W;
        xan = L.axis_x[x]; yan = L.axis_y[y];
        xbn = v.x; ybn = v.y;
    }
}

}  // namespace

// =================================================================================================
// F5: per-(reference, neighbour) constants, once per batch (skipped when the batch is unchanged)
// =================================================================================================
extern "C" __global__ void lfd_pair_setup_kernel(LfdLaunch L, LfdRefConst* __restrict__ ref_out,
                                                 LfdPairConst* __restrict__ pair_out) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int n_pairs = L.n_refs * L.k;
    if (i < n_pairs) {
        const int r = i / L.k, j = i - r * L.k;
        if (j < L.refs[r].n_slots)
            lfd_make_pair_const(L.cams[L.refs[r].cam], L.cams[L.slots[i].cam], L.slots[i].cam, L.w_match, L.h_match, pair_out[i]);
    } else if (i < n_pairs + L.n_refs) {
        const int r = i - n_pairs;
        lfd_make_ref_const(L.cams[L.refs[r].cam], L.w_match, L.h_match, ref_out[r]);
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

// =================================================================================================
// fused dense kernel
// =================================================================================================
extern "C" __global__ void __launch_bounds__(kBlock, 3) lfd_dense_kernel(LfdLaunch L) {
    __shared__ BlockShared S;
    __shared__ unsigned s_ticket;
    __shared__ unsigned s_wave_cnt[kBlock / 64];
    __shared__ unsigned s_slot_cnt[LFD_MAX_SLOTS];
    __shared__ u64 s_tile_excl;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ticket = (unsigned)(atomicAdd(L.ticket, 1ull) - L.ticket_base);
    if (tid < LFD_MAX_SLOTS) s_slot_cnt[tid] = 0;
    __syncthreads();
    const unsigned ticket = s_ticket;
    const int r = (int)(ticket / (unsigned)L.tiles_per_ref);
    const int tile_in_ref = (int)(ticket - (unsigned)r * (unsigned)L.tiles_per_ref);
    block_prologue(L, r, S);

    const int HW = L.H * L.W;
    const int ns = S.ref.n_slots;
    const int cell0 = tile_in_ref * kTile + tid * kCpt;
    bool any_mask = S.ref.mask_a != nullptr;
    for (int j = 0; j < ns; ++j) any_mask |= (S.slot[j].mask_b != nullptr);

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

    // ---- stage 2: winner's warp (8 or 16 B per cell), issued together so the loads overlap --------
    float xan[kCpt], yan[kCpt], xbn[kCpt], ybn[kCpt];
#pragma unroll
    for (int e = 0; e < kCpt; ++e) {
        xan[e] = yan[e] = xbn[e] = ybn[e] = 0.0f;
        if (cell0 + e < HW) cell_coords(L, S, cell0 + e, bj[e], xan[e], yan[e], xbn[e], ybn[e]);
    }

    // ---- stage 3: per-correspondence geometry ---------------------------------------------------
    float ox[kCpt], oy[kCpt], oz[kCpt], oe[kCpt], opx[kCpt], opy[kCpt];
    unsigned keep_bits = 0;
#pragma unroll
    for (int e = 0; e < kCpt; ++e) {
        LfdCellResult res;
        res.keep = 0; res.x = res.y = res.z = res.err = res.xa_px = res.ya_px = 0.0f;
        if (cell0 + e < HW) lfd_eval_correspondence(S.rc, S.pc[bj[e]], xan[e], yan[e], xbn[e], ybn[e], L.kp, res);
        ox[e] = res.x; oy[e] = res.y; oz[e] = res.z; oe[e] = res.err; opx[e] = res.xa_px; opy[e] = res.ya_px;
        keep_bits |= (res.keep ? 1u : 0u) << e;
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
    // survivors per neighbour slot (order-independent integer sums)
    for (int j = 0; j < ns; ++j) {
        unsigned c = 0;
#pragma unroll
        for (int e = 0; e < kCpt; ++e) c += ((keep_bits >> e) & 1u) && (bj[e] == j);
        c = (unsigned)wave_sum_u64(c);
        if (lane == 0 && c) atomicAdd(&s_slot_cnt[j], c);
    }
    __syncthreads();
    unsigned wave_off = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
        if (w < wave) wave_off += s_wave_cnt[w];
        block_total += s_wave_cnt[w];
    }
    if (wave == 0) {
        const u64 excl = lookback_exclusive(L.tile_state, L.epoch, ticket, block_total);
        if (lane == 0) {
            s_tile_excl = excl;
            if (tile_in_ref == 0) L.ref_offsets[r] = (long long)excl;
            if (ticket == (unsigned)(L.n_refs * L.tiles_per_ref) - 1u) L.ref_offsets[L.n_refs] = (long long)(excl + block_total);
        }
    }
    if (L.seg_counts && tid < ns && s_slot_cnt[tid]) atomicAdd(&L.seg_counts[(size_t)r * L.k + tid], (int)s_slot_cnt[tid]);
    __syncthreads();

    // ---- stage 5: colour for survivors + write --------------------------------------------------------
    long long pos = (long long)s_tile_excl + wave_off + (incl - my_cnt);
    const float sx_img = 1.0f, sy_img = 1.0f;   // the image handed over is already at match resolution
#pragma unroll
    for (int e = 0; e < kCpt; ++e) {
        if (!((keep_bits >> e) & 1u)) continue;
        if (pos < L.capacity) {
            float rgb[3];
            lfd_bilinear_rgb(S.ref.image, L.w_match, L.h_match, opx[e], opy[e], sx_img, sy_img, rgb);
            L.xyz[pos * 3 + 0] = ox[e]; L.xyz[pos * 3 + 1] = oy[e]; L.xyz[pos * 3 + 2] = oz[e];
            L.rgb[pos * 3 + 0] = rgb[0]; L.rgb[pos * 3 + 1] = rgb[1]; L.rgb[pos * 3 + 2] = rgb[2];
            L.err[pos] = oe[e];
            if (L.cell) L.cell[pos] = cell0 + e;
            if (L.slot) L.slot[pos] = (uint8_t)bj[e];
        }   // beyond capacity: counted, not written (the caller compares the total with capacity)
        ++pos;
    }
}

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
        const u64 excl = lookback_exclusive(L.tile_state, L.epoch, (unsigned)r, s_total);
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
