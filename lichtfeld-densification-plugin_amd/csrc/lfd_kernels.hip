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

// Every plane, mask, image and axis handed to the kernels lives in device (global) memory, but the pointers reach
// the threads through LDS descriptors, so the compiler would have to use FLAT loads (address-space check per
// access, both wait counters).  Casting to the global address space selects global_load_* instead.
#define LFD_GLOBAL_AS __attribute__((address_space(1)))
typedef float lfd_f32x4 __attribute__((ext_vector_type(4)));
typedef float lfd_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned lfd_u32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
template <class T>
__device__ __forceinline__ const T LFD_GLOBAL_AS* lfd_global(const T* p) { return (const T LFD_GLOBAL_AS*)p; }
__device__ __forceinline__ float4 load_f32x4(const float* p) {
#if LFD_NT_LOADS
    const lfd_f32x4 v = __builtin_nontemporal_load((const lfd_f32x4 LFD_GLOBAL_AS*)p);
#else
    const lfd_f32x4 v = *(const lfd_f32x4 LFD_GLOBAL_AS*)p;
#endif
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 load_f32x2(const float* p) {
#if LFD_NT_LOADS
    const lfd_f32x2 v = __builtin_nontemporal_load((const lfd_f32x2 LFD_GLOBAL_AS*)p);
#else
    const lfd_f32x2 v = *(const lfd_f32x2 LFD_GLOBAL_AS*)p;
#endif
    return make_float2(v.x, v.y);
}
#pragma clang diagnostic pop

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
    // A window of 64 * kPer predecessors per memory round trip: every lane reads kPer consecutive words (nearest first).
    // The tiles that are themselves waiting here have published their count but no prefix yet; with N of them in flight a
    // tile walks back through N / window round trips before it meets a prefix, and that walk - not the arithmetic - sets the
    // life of a tile once N is in the hundreds (1792 resident tiles, about a third of them in this loop): a wide window.
    constexpr int kPer = LFD_LOOKBACK_PER_LANE;
    u64 excl = 0;
    long long base = (long long)tile - 1;
#if LFD_LOOKBACK_WATCH_ONE
    // While the predecessors are still computing, ONE lane watches ONE word - the nearest predecessor's.  Tile-state words are
    // read past the caches (every XCD writes them), and a wave that re-reads its whole window every microsecond, times the
    // ~500 waves that wait here at any moment, is measurable traffic on the fabric (longer sleeps between the polls alone
    // were worth 4 %).  Tiles publish roughly in ticket order, so when the nearest one has, the window read below rarely
    // finds a gap.  The watch sleeps ~4 us between two looks (measured: 0.25 us 0.308 ms, 1 us 0.304, 3 us 0.299, 4 us 0.297,
    // 8 us 0.300, 16 us 0.318: a tile waits 12 us on average, detecting the end of the wait 2 us late costs less than looking).
    {
        u64 w = 0;
        bool waiting = lane == 0;
        if (waiting) w = state_load(state + base);
        while (__any(waiting && state_status(w, epoch) == kStEmpty)) {
            if (++spins > (LFD_SPIN_LIMIT >> 3)) {       // (~4 us per look: the same few seconds as the window's limit)
                if (lane == 0) atomicExch(L.status, LFD_LAUNCH_TIMEOUT);
                return 0;
            }
            __builtin_amdgcn_s_sleep(LFD_WATCH_SLEEP);
            if (waiting && state_status(w, epoch) == kStEmpty) w = state_load(state + base);
        }
    }
#endif
    while (true) {
        const long long j0 = base - (long long)lane * kPer;
        u64 s[kPer];
#pragma unroll
        for (int p = 0; p < kPer; ++p) {
            if (lane >= LFD_LOOKBACK_LANES) s[p] = pack_state(kStAggregate, epoch, 0);      // lanes beyond the window: nothing to add
            else s[p] = (j0 - p >= 0) ? state_load(state + (j0 - p)) : pack_state(kStPrefix, epoch, 0);   // virtual tile -1: prefix 0
        }
        for (;;) {
            bool empty = false;
#pragma unroll
            for (int p = 0; p < kPer; ++p) empty = empty || state_status(s[p], epoch) == kStEmpty;
            if (!__any(empty)) break;
            if (++spins > LFD_SPIN_LIMIT) {          // never hang the GPU: report and bail out
                if (lane == 0) atomicExch(L.status, LFD_LAUNCH_TIMEOUT);
                return 0;
            }
            __builtin_amdgcn_s_sleep(LFD_POLL_SLEEP);
#pragma unroll
            for (int p = 0; p < kPer; ++p)
                if (state_status(s[p], epoch) == kStEmpty) s[p] = state_load(state + (j0 - p));
        }
        // this lane's words up to and including its nearest prefix
        u64 sum = 0;
        bool has = false;
#pragma unroll
        for (int p = 0; p < kPer; ++p) {
            if (!has) { sum += s[p] & kValueMask; has = state_status(s[p], epoch) == kStPrefix; }
        }
        const u64 with_prefix = __ballot(has);
        if (with_prefix) {
            const int first = __ffsll((long long)with_prefix) - 1;   // the lane that holds the nearest predecessor with a full prefix
            excl += wave_sum_u64(lane <= first ? sum : 0ull);
            break;
        }
        excl += wave_sum_u64(sum);
        base -= LFD_LOOKBACK_LANES * kPer;
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

// the same in two halves, so that the loads can be in flight while something else (the ticket atomic) is issued:
// prologue_issue requests this thread's words of the four blocks, prologue_commit parks them in LDS (no barrier)
struct PrologueRegs { unsigned pw[(LFD_MAX_SLOTS * sizeof(LfdPairConst) / 4 + LFD_DENSE_BLOCK - 1) / LFD_DENSE_BLOCK]; unsigned sw, rw; int k; };

__device__ __forceinline__ void prologue_issue(const LfdLaunch& L, int r, PrologueRegs& P) {
    // every load is unconditional (clamped index): straight-line code that the compiler cannot sink below the ticket atomic
    // into the guarded LDS stores of prologue_commit
    const int tid = (int)threadIdx.x;
    constexpr int kPw = (int)(sizeof(P.pw) / sizeof(P.pw[0]));
    const unsigned LFD_GLOBAL_AS* src = lfd_global(reinterpret_cast<const unsigned*>(L.pair_const + (size_t)r * L.k));
    const int nw = L.k * (int)(sizeof(LfdPairConst) / 4);
#pragma unroll
    for (int q = 0; q < kPw; ++q) { const int i = tid + q * (int)blockDim.x; P.pw[q] = src[i < nw ? i : nw - 1]; }
    const unsigned LFD_GLOBAL_AS* ssrc = lfd_global(reinterpret_cast<const unsigned*>(L.slots + (size_t)r * L.k));
    const int nsw = L.k * (int)(sizeof(LfdSlotDesc) / 4);
    P.sw = ssrc[tid < nsw ? tid : nsw - 1];
    // reference constants (18 words) by threads 0..17, reference descriptor (8 words) by threads 64..71: one load per thread
    const unsigned LFD_GLOBAL_AS* rsrc = lfd_global(reinterpret_cast<const unsigned*>(L.ref_const + r));
    const unsigned LFD_GLOBAL_AS* dsrc = lfd_global(reinterpret_cast<const unsigned*>(L.refs + r));
    constexpr int kRc = (int)(sizeof(LfdRefConst) / 4), kRd = (int)(sizeof(LfdRefDesc) / 4);
    const bool second = tid >= 64;
    const int i2 = second ? (tid - 64 < kRd ? tid - 64 : kRd - 1) : (tid < kRc ? tid : kRc - 1);
    P.rw = (second ? dsrc : rsrc)[i2];
    P.k = L.k;
    __builtin_amdgcn_sched_barrier(0);         // the requests are out before anything below is issued (and nothing waits for them here)
}

__device__ __forceinline__ void prologue_commit(const LfdLaunch& L, const PrologueRegs& P, BlockShared& S) {
    const int tid = (int)threadIdx.x;
    constexpr int kPw = (int)(sizeof(P.pw) / sizeof(P.pw[0]));
    unsigned* dst = reinterpret_cast<unsigned*>(S.pc);
    const int nw = L.k * (int)(sizeof(LfdPairConst) / 4);
#pragma unroll
    for (int q = 0; q < kPw; ++q) { const int i = tid + q * (int)blockDim.x; if (i < nw) dst[i] = P.pw[q]; }
    const int nsw = L.k * (int)(sizeof(LfdSlotDesc) / 4);
    if (tid < nsw) reinterpret_cast<unsigned*>(S.slot)[tid] = P.sw;
    if (tid < (int)(sizeof(LfdRefConst) / 4)) reinterpret_cast<unsigned*>(&S.rc)[tid] = P.rw;
    if (tid >= 64 && tid < 64 + (int)(sizeof(LfdRefDesc) / 4)) reinterpret_cast<unsigned*>(&S.ref)[tid - 64] = P.rw;
}

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

// the same for a small offset from a known (row, column): local = column0 + offset < 2^22, so the float estimate
// is within one of the quotient and a single correction each way is enough
__device__ __forceinline__ void lfd_divmod_local(int local, int W, float inv_w, int w_log2, int& dy, int& x) {
    if (w_log2 >= 0) { dy = local >> w_log2; x = local & (W - 1); return; }     // power-of-two rows (uniform branch): two instructions
    int q = (int)((float)local * inv_w);
    int r = local - __mul24(q, W);      // q <= 2^22 / W, W < 2^16: the 24-bit multiply is exact (and full rate; v_mul_lo_u32 is quarter rate)
    if (r < 0) { --q; r += W; }
    if (r >= W) { ++q; r -= W; }
    dy = q; x = r;
}

// certainty of one slot at one cell after the prologue of core/pipeline.py:407-430
__device__ __forceinline__ float cell_cert(const LfdLaunch& L, const BlockShared& S, int j, int cell, int x, int y,
                                           float raw, float mask_a_val) {
    float c = lfd_cert_floor(raw, L.kp.certainty_thresh);
    if (S.ref.mask_a) c = c * mask_a_val;
    const uint8_t* mb = S.slot[j].mask_b;
    if (mb) {
        const float2 wv = load_f32x2(S.slot[j].warp + (size_t)cell * L.warp_channels + (L.warp_channels - 2));
        const int ix = lfd_grid_nearest(wv.x, L.W);
        const int iy = lfd_grid_nearest(wv.y, L.H);
        float m = 0.0f;
        if (ix >= 0 && iy >= 0)
            m = (float)lfd_global(mb)[(size_t)lfd_nearest_src(iy, L.mask_sy, L.h_match) * L.w_match + lfd_nearest_src(ix, L.mask_sx, L.w_match)];
        c = c * m;
    }
    return c;
}

__device__ __forceinline__ float cell_mask_a(const LfdLaunch& L, const BlockShared& S, int x, int y) {
    if (!S.ref.mask_a) return 1.0f;
    return (float)lfd_global(S.ref.mask_a)[(size_t)lfd_nearest_src(y, L.mask_sy, L.h_match) * L.w_match + lfd_nearest_src(x, L.mask_sx, L.w_match)];
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
    best = cell_cert(L, S, 0, cell, x, y, lfd_global(S.slot[0].cert)[cell], ma);
    bj = 0;
    for (int j = 1; j < ns; ++j) argmax_step(cell_cert(L, S, j, cell, x, y, lfd_global(S.slot[j].cert)[cell], ma), j, best, bj);
}

// the same for four consecutive cells of one row when masks are present (W % 4 == 0): certainties and warps come in
// 16-byte loads, the row / column arithmetic is shared; per cell the operations and their order are those of cell_cert
__device__ __forceinline__ void cells4_best_masked(const LfdLaunch& L, const BlockShared& S, int cell0, int y, int x0,
                                                   float best[4], int bj[4]) {
    const float th = L.kp.certainty_thresh;
    float ma[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    const bool has_a = S.ref.mask_a != nullptr;
    if (has_a) {
        const uint8_t LFD_GLOBAL_AS* row = lfd_global(S.ref.mask_a) + (size_t)lfd_nearest_src(y, L.mask_sy, L.h_match) * L.w_match;
#pragma unroll
        for (int e = 0; e < 4; ++e) ma[e] = (float)row[lfd_nearest_src(x0 + e, L.mask_sx, L.w_match)];
    }
    const int ns = S.ref.n_slots;
    for (int j = 0; j < ns; ++j) {
        const float4 c4 = load_f32x4(S.slot[j].cert + cell0);
        float c[4] = {lfd_cert_floor(c4.x, th), lfd_cert_floor(c4.y, th), lfd_cert_floor(c4.z, th), lfd_cert_floor(c4.w, th)};
        if (has_a) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = c[e] * ma[e];
        }
        const uint8_t* mb = S.slot[j].mask_b;
        if (mb) {
            float wx[4], wy[4];
            const float* wp = S.slot[j].warp;
            if (L.warp_channels == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float4 v = load_f32x4(wp + (size_t)(unsigned)(cell0 + e) * 4); wx[e] = v.z; wy[e] = v.w; }
            } else {
                const float4 a = load_f32x4(wp + (size_t)(unsigned)cell0 * 2), b = load_f32x4(wp + (size_t)(unsigned)cell0 * 2 + 4);
                wx[0] = a.x; wy[0] = a.y; wx[1] = a.z; wy[1] = a.w; wx[2] = b.x; wy[2] = b.y; wx[3] = b.z; wy[3] = b.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ix = lfd_grid_nearest(wx[e], L.W);
                const int iy = lfd_grid_nearest(wy[e], L.H);
                float m = 0.0f;
                if (ix >= 0 && iy >= 0)
                    m = (float)lfd_global(mb)[(size_t)lfd_nearest_src(iy, L.mask_sy, L.h_match) * L.w_match + lfd_nearest_src(ix, L.mask_sx, L.w_match)];
                c[e] = c[e] * m;
            }
        }
        if (j == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { best[e] = c[e]; bj[e] = 0; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) argmax_step(c[e], j, best[e], bj[e]);
        }
    }
}

// winner's warp -> normalised coordinates of the correspondence
__device__ __forceinline__ void cell_coords(const LfdLaunch& L, const BlockShared& S, int cell, int bj, float& xan,
                                            float& yan, float& xbn, float& ybn) {
    const float* wp = S.slot[bj].warp;
    if (L.warp_channels == 4) {
        const float4 v = load_f32x4(wp + (size_t)cell * 4);
        xan = v.x; yan = v.y; xbn = v.z; ybn = v.w;
    } else {
        const float2 v = load_f32x2(wp + (size_t)cell * 2);
        int y, x;
        lfd_divmod(cell, L.W, L.inv_w, y, x);
        xan = lfd_global(L.axis_x)[x]; yan = lfd_global(L.axis_y)[y];
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
        if (j < L.refs[r].n_slots) {
            // (with the caller's F - upstream's fundamental_from_world2cam result - nothing is derived here: half of this kernel's work)
            lfd_make_pair_const(L.cams[L.refs[r].cam], L.cams[L.slots[i].cam], L.slots[i].cam, L.w_match, L.h_match, pair_out[i],
                                L.fund_override ? L.fund_override + (size_t)i * 9 : nullptr);
        }
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
            float4 best = load_f32x4(S.slot[0].cert + base);
            const float th = L.kp.certainty_thresh;
            best.x = lfd_cert_floor(best.x, th); best.y = lfd_cert_floor(best.y, th);
            best.z = lfd_cert_floor(best.z, th); best.w = lfd_cert_floor(best.w, th);
            int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            for (int j = 1; j < ns; ++j) {
                const float4 c = load_f32x4(S.slot[j].cert + base);
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
        } else if ((L.W & 3) == 0 && (HW & 3) == 0 && base + 3 < HW) {      // masks, four cells of one row at a time
            int y, x0;
            lfd_divmod(base, L.W, L.inv_w, y, x0);
            float b4[4]; int j4[4];
            cells4_best_masked(L, S, base, y, x0, b4, j4);
            *reinterpret_cast<float4*>(best_cert + (size_t)r * HW + base) = make_float4(b4[0], b4[1], b4[2], b4[3]);
            if (best_slot) {
                const unsigned packed = (unsigned)j4[0] | ((unsigned)j4[1] << 8) | ((unsigned)j4[2] << 16) | ((unsigned)j4[3] << 24);
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
#define LFD_CONST_AS __attribute__((address_space(4)))
// Loads through a constant-address-space pointer with a uniform address are selected as scalar loads
// (s_load_*): they cost no vector-memory or LDS traffic and land in SGPRs.
template <class T>
__device__ __forceinline__ const T LFD_CONST_AS* lfd_const_as(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const T LFD_CONST_AS*)p;
#pragma clang diagnostic pop
}

// 12 * slot as a 24-bit multiply (full rate).  Written as an instruction because the optimiser turns every other spelling of it
// back into v_mul_lo_u32, which runs at a quarter of the rate - once per cell in the geometry loop.
__device__ __forceinline__ int lfd_slot_bytes12(int sl) {
    int o;
    asm("v_mul_u32_u24 %0, %1, 12" : "=v"(o) : "v"(sl));
    return o;
}

struct DenseStage {                // per-tile results, indexed by the cell's slot inside the tile
    float xyz[3 * kTile];
    float err[kTile];
    unsigned char slot[kTile];     // (the order map - order[i] = tile slot of the i-th survivor, raster order - lives in BlockShared::pc, see below)
};

struct __attribute__((packed, aligned(4))) LfdF3 { float a, b, c; };

// records a thread of the copy-out waves handles: the tile's survivors are shared out over the waves that do NOT run the look-back
constexpr int kCopyThreadsOrdered = kBlock - 64;

// phase stamps of the dense kernel (profiling builds only): lane 0 of waves 0 and 1 record the shader clock at the phase
// boundaries of their tile; profiles/dense_phases.py turns them into a per-phase latency table
#if defined(LFD_DENSE_TIMING)
// (written straight to memory, keyed by the workgroup index - tickets and workgroup indices run in step to within a few tiles: twelve
// stamps kept in registers would cost the timing build the eighth wave per SIMD)
#define LFD_STAMP(i) do { if (L.phase_stamps && lane == 0 && wave < 2) L.phase_stamps[((size_t)blockIdx.x * 2 + wave) * 12 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define LFD_STAMP(i) do { } while (0)
#endif

// kUnordered (lfd_triangulate_dense_segments): no look-back.  A tile claims room in its reference's region of the output with one
// atomic on the reference's cursor and records {offset, count} in the tile table; the consumers (lfd_order_segments, lfd_pack_*_segments)
// walk the table and emit raster order.  Everything up to the retirement is the same code.
// kPly (lfd_triangulate_dense_ply): the survivors leave as 15-byte PLY vertex records (xyz f32 LE + the colour quantised like upstream's
// to_uint8_rgb) instead of the 28-byte structure of arrays: the packer's pass over the cloud disappears for streamed / exchanged output.
template <bool kExactColour, bool kUnordered = false, bool kPly = false>
__device__ __forceinline__ void lfd_dense_body(const LfdLaunch& L) {
    // one LDS block, the per-pair constants first: S.pc[slot] is then addressed as slot * sizeof(LfdPairConst) + an instruction offset,
    // with no base to keep in a vector register across the geometry loop
    struct DenseShared {
        BlockShared S;
        DenseStage stage;
        u64 tile_excl;
        unsigned ticket;
        unsigned wave_cnt[kBlock / 64];
        unsigned slot_cnt[LFD_MAX_SLOTS];
    };
    __shared__ DenseShared sh;
    BlockShared& S = sh.S;
    DenseStage& stage = sh.stage;
    // The order map (2 KB) lives where the per-pair constants were: those are last read in the geometry loop, the map is written behind
    // the barrier that follows it.  With it the block is 20.4 KB: eight workgroups fit the 160 KB of a CU (22.4 KB: seven).
    static_assert(sizeof(sh.S.pc) >= kTile * sizeof(unsigned short), "the order map does not fit the per-pair constant block");
    unsigned short* const stage_order = reinterpret_cast<unsigned short*>(sh.S.pc);
    unsigned& s_ticket = sh.ticket;
    unsigned (&s_wave_cnt)[kBlock / 64] = sh.wave_cnt;
    unsigned (&s_slot_cnt)[LFD_MAX_SLOTS] = sh.slot_cnt;
    u64& s_tile_excl = sh.tile_excl;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
#if defined(LFD_DENSE_TIMING)
#endif
    LFD_STAMP(0);
    // Tickets: a workgroup learns its tile from an atomic counter, so every tile a look-back can wait for is
    // already being worked on.  Returning atomics on ONE address complete at ~12 ns each on MI355X
    // (profiles/microbench/latency.hip: 16384 tickets = 0.2 ms), so the counter is split in LFD_TICKET_LANES
    // interleaved sequences on separate cache lines: workgroup b draws from sequence b % LANES, whose k-th
    // ticket is tile k*LANES + b % LANES.  Once workgroups 0..N-1 have started, the claimed tiles are exactly
    // 0..N-1 (whichever workgroup of a class drew which ticket), so the lowest unfinished tile always belongs to
    // a running workgroup, as with a single counter; the 8 classes also match the round-robin dealing of
    // workgroups to the 8 XCDs.
    const unsigned seq = blockIdx.x % LFD_TICKET_LANES;
    const unsigned n_tiles = (unsigned)L.n_refs * (unsigned)L.tiles_per_ref;
    const int HW = L.H * L.W;

    // ---- front end: ONE memory round trip for the ticket and the constants ---------------------------------------------
    // Workgroup b nearly always draws a ticket of the same reference as tile b (tickets and workgroup indices run in step to
    // within a few tiles; a reference has hundreds of tiles), so the constants of THAT reference are requested before the
    // ticket is known, together with the ticket atomic, instead of after it.  If the ticket belongs to another reference
    // the block is fetched again the plain way: results never depend on the guess.  (Requesting the certainty planes of
    // tile b the same way does not pay: measured, only 7 % of the workgroups draw exactly ticket b.)
#if LFD_STAGGER_UNITS > 0
    // The workgroups that are resident when the launch starts all begin at once and would run through their phases in step -
    // every tile fetching, then every tile computing (the first generation of tiles lives 48 us, the later ones 33).  The k-th
    // workgroup a CU receives waits k x ~1.5 us before it draws its ticket (a heuristic on the dispatch order: harmless where it
    // does not hold).  Measured 0.2966 -> 0.292 ms (profiles/history.md (r2/ablation.txt)).
    if (blockIdx.x < (unsigned)LFD_DENSE_WAVES_PER_SIMD * 256u) {      // (one resident workgroup per wave slot of a SIMD, 256 CUs)
        const unsigned slot = blockIdx.x >> 8;
        for (unsigned i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(LFD_STAGGER_UNITS);
    }
#endif
    const unsigned tile_guess = blockIdx.x;
    const int r_guess = __builtin_amdgcn_readfirstlane((int)(tile_guess / (unsigned)L.tiles_per_ref));
#if LFD_FRONT_PRIO > 0
    __builtin_amdgcn_s_setprio(LFD_FRONT_PRIO);     // the few instructions between the front end's memory requests go ahead of other waves' arithmetic
#endif
    PrologueRegs pro;
    if (tile_guess < n_tiles) prologue_issue(L, r_guess, pro);        // constants of the guessed reference: loads in flight
    if (tid == 0) {
        const unsigned long long k = atomicAdd(L.ticket_lanes + (size_t)seq * 16, 1ull) - L.ticket_base_lane[seq];
        s_ticket = (unsigned)k * LFD_TICKET_LANES + seq;
    }
    if (tid < LFD_MAX_SLOTS) s_slot_cnt[tid] = 0;
    if (tile_guess < n_tiles) prologue_commit(L, pro, S);             // ... and into LDS
    __syncthreads();
    LFD_STAMP(1);
    const unsigned tile = s_ticket;
    if (tile == 0 && (L.seg_counts || kUnordered)) {         // zero the per-(reference, slot) counters (and the references' cursors) for this launch, then raise the flag
        if (L.seg_counts) for (int i = tid; i < L.n_refs * L.k; i += kBlock) L.seg_counts[i] = 0;
        if (kUnordered) for (int i = tid; i < L.n_refs; i += kBlock) L.ref_cursor[i] = 0ull;
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(L.seg_ready, L.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tile < n_tiles) {
        const int r = __builtin_amdgcn_readfirstlane((int)(tile / (unsigned)L.tiles_per_ref));
        const int tile_in_ref = (int)(tile - (unsigned)r * (unsigned)L.tiles_per_ref);
        // the reference camera's block is the same for every lane: scalar loads keep it in SGPRs instead of LDS reads
        LfdRefConst rc;
        {
            const LfdRefConst LFD_CONST_AS* rcp = lfd_const_as(L.ref_const) + r;
#pragma unroll
            for (int i = 0; i < 12; ++i) rc.P[i] = rcp->P[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) rc.C[i] = rcp->C[i];
            rc.sx = rcp->sx; rc.sy = rcp->sy; rc.pad = 0.0f;
        }
        if (r != r_guess || tile_guess >= n_tiles) {      // the guess was wrong: this tile's constants the plain way (uniform branch)
            __syncthreads();                 // nobody reads the guessed block any more
            block_prologue(L, r, S);         // ends with a barrier
        }
        LFD_STAMP(2);
#if defined(LFD_DENSE_TIMING)
        if (L.phase_stamps && lane == 0 && wave < 2) {      // did the guess hold?  (low bits of stamp 0)
            unsigned long long* s0 = L.phase_stamps + ((size_t)blockIdx.x * 2 + wave) * 12;
            *s0 = (*s0 & ~3ull) | (r == r_guess ? 1ull : 0ull) | (tile == tile_guess ? 2ull : 0ull);
        }
#endif
        const int ns = S.ref.n_slots;
        const bool any_mask = S.ref.any_mask != 0;
        const int tile_cell0 = tile_in_ref * kTile;
        const int cell0 = tile_cell0 + tid * kCpt;
        int tile_y0, tile_x0;                    // (row, column) of the tile's first cell: uniform, kept in SGPRs
        {
            int ty, tx;
            lfd_divmod(tile_cell0, L.W, L.inv_w, ty, tx);
            tile_y0 = __builtin_amdgcn_readfirstlane(ty); tile_x0 = __builtin_amdgcn_readfirstlane(tx);
        }

        // ---- stage 1: certainty floor + arg-max over the neighbour slots (coalesced 16-B loads) -----
        int bj[kCpt];
        // cells whose best certainty is <= 0 (masked out; a floor of 0 with a certainty of 0): not candidates - upstream's sampler cannot draw
        // them (core/sampling.py:24-27, 41-43 upstream).  Four bits that live until the coordinates are parked: a dead cell is parked with a NaN
        // neighbour coordinate, which every branch of the per-cell routine rejects (Sampson `<`, reprojection `<=`, isfinite) - the geometry loop
        // carries no flag and no test for it.
        unsigned dead = 0;
#if LFD_DENSE_ALL_WARPS
        // Up to four neighbours, two-channel warps, no masks: the warps of ALL slots are requested together with the certainty
        // planes (dense 16-byte loads) and the winner's is picked in registers, instead of a second, dependent round trip for the
        // winner's warp alone (8-byte loads at a 32-byte stride).  Costs 8 (k - 1) more bytes per cell of HBM traffic - the
        // kernel is bound by its chain of memory round trips, not by bandwidth (profiles/history.md (r2/phases_*.txt)).
        float spec_xb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, spec_yb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        bool have_warps = false;
        if (!any_mask && (HW & 3) == 0 && kCpt == 4 && cell0 + 3 < HW && ns <= LFD_DENSE_ALL_WARPS && L.warp_channels == 2) {
            const float th = L.kp.certainty_thresh;
            float4 c[4], wa[4], wb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < ns) {
                    c[j] = load_f32x4(S.slot[j].cert + cell0);
                    wa[j] = load_f32x4(S.slot[j].warp + (size_t)(unsigned)cell0 * 2);           // cells 0, 1: xB yB xB yB
                    wb[j] = load_f32x4(S.slot[j].warp + (size_t)(unsigned)cell0 * 2 + 4);       // cells 2, 3
                }
            }
            float4 best = c[0];
            best.x = lfd_cert_floor(best.x, th); best.y = lfd_cert_floor(best.y, th);
            best.z = lfd_cert_floor(best.z, th); best.w = lfd_cert_floor(best.w, th);
            bj[0] = bj[1] = bj[2] = bj[3] = 0;
            spec_xb[0] = wa[0].x; spec_yb[0] = wa[0].y; spec_xb[1] = wa[0].z; spec_yb[1] = wa[0].w;
            spec_xb[2] = wb[0].x; spec_yb[2] = wb[0].y; spec_xb[3] = wb[0].z; spec_yb[3] = wb[0].w;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                if (j < ns) {
                    int t;
                    t = bj[0]; argmax_step(lfd_cert_floor(c[j].x, th), j, best.x, bj[0]); if (bj[0] != t) { spec_xb[0] = wa[j].x; spec_yb[0] = wa[j].y; }
                    t = bj[1]; argmax_step(lfd_cert_floor(c[j].y, th), j, best.y, bj[1]); if (bj[1] != t) { spec_xb[1] = wa[j].z; spec_yb[1] = wa[j].w; }
                    t = bj[2]; argmax_step(lfd_cert_floor(c[j].z, th), j, best.z, bj[2]); if (bj[2] != t) { spec_xb[2] = wb[j].x; spec_yb[2] = wb[j].y; }
                    t = bj[3]; argmax_step(lfd_cert_floor(c[j].w, th), j, best.w, bj[3]); if (bj[3] != t) { spec_xb[3] = wb[j].z; spec_yb[3] = wb[j].w; }
                }
            }
            have_warps = true;
            if (!(th > 0.0f))          // (a floored certainty is >= the floor: with a positive floor - a scalar test - no cell of an unmasked reference is dead)
                dead = (unsigned)(best.x <= 0.0f) | ((unsigned)(best.y <= 0.0f) << 1) | ((unsigned)(best.z <= 0.0f) << 2) | ((unsigned)(best.w <= 0.0f) << 3);
        } else
#endif
        if (!any_mask && (HW & 3) == 0 && kCpt == 4 && cell0 + 3 < HW) {
            const float th = L.kp.certainty_thresh;
            float4 best;
            bj[0] = bj[1] = bj[2] = bj[3] = 0;
            {                                    // four planes in flight at a time: ONE dependent round trip per four neighbours (a
                                                 // load-then-compare loop over the slots costs one round trip per neighbour)
                best = load_f32x4(S.slot[0].cert + cell0);
                best.x = lfd_cert_floor(best.x, th); best.y = lfd_cert_floor(best.y, th);
                best.z = lfd_cert_floor(best.z, th); best.w = lfd_cert_floor(best.w, th);
                for (int j0 = 1; j0 < ns; j0 += 4) {
                    float4 c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (j0 + u < ns) c[u] = load_f32x4(S.slot[j0 + u].cert + cell0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j0 + u < ns) {
                            argmax_step(lfd_cert_floor(c[u].x, th), j0 + u, best.x, bj[0]);
                            argmax_step(lfd_cert_floor(c[u].y, th), j0 + u, best.y, bj[1]);
                            argmax_step(lfd_cert_floor(c[u].z, th), j0 + u, best.z, bj[2]);
                            argmax_step(lfd_cert_floor(c[u].w, th), j0 + u, best.w, bj[3]);
                        }
                    }
                }
            }
            if (!(th > 0.0f))
                dead = (unsigned)(best.x <= 0.0f) | ((unsigned)(best.y <= 0.0f) << 1) | ((unsigned)(best.z <= 0.0f) << 2) | ((unsigned)(best.w <= 0.0f) << 3);
        } else if ((L.W & 3) == 0 && (HW & 3) == 0 && kCpt == 4 && cell0 + 3 < HW) {      // masks present
            int dy, x0;
            lfd_divmod_local(tile_x0 + tid * kCpt, L.W, L.inv_w, L.w_log2, dy, x0);
            float b4[4];
            cells4_best_masked(L, S, cell0, tile_y0 + dy, x0, b4, bj);
#pragma unroll
            for (int e = 0; e < 4; ++e) dead |= (unsigned)(b4[e] <= 0.0f) << e;
        } else {
#pragma unroll
            for (int e = 0; e < kCpt; ++e) {
                bj[e] = 0;
                if (cell0 + e < HW) { float b; cell_best(L, S, cell0 + e, b, bj[e]); dead |= (unsigned)(b <= 0.0f) << e; }
            }
        }

        LFD_STAMP(3);
        // ---- stage 2: winner's warp (8 or 16 B per cell), all four loads in flight together; the
        //      coordinates are parked in this thread's own LDS slots (the slots later receive the cell's
        //      outputs), so the geometry loop below carries no per-cell register arrays ------------------
        unsigned bj_packed = 0;                   // 4 x 8 bits: the winning slot of each of this thread's cells
#if LFD_WARP_BY_LANE
        // A thread's four cells are CONSECUTIVE (the certainty planes want 16-byte loads), so the winner's warp read cell by cell is a wave-wide
        // load of 8 bytes at a 32-byte stride - sixteen 128-byte lines per plane and load, up to 48 with three planes, four loads per wave - in a
        // memory pipeline whose queue is what a tile's round trips wait in (MI355X_MICROARCH: the price of a hop sits in the CU's own queue).  Here the
        // winners' slots go through LDS to the lanes of the SAME wave and lane l fetches the cells l, l + 64, l + 128, l + 192 of the wave's 256:
        // consecutive lanes, consecutive cells - four lines per plane and load - and parks the coordinates in THOSE cells' slots (word stride 3 and 1
        // instead of 12 and 4).  LDS operations of one wave complete in order: no barrier.  Same bits.
        const bool wave_whole = kCpt == 4 && (L.W & 3) == 0 && tile_cell0 + (wave + 1) * 64 * kCpt <= HW;
#if LFD_DENSE_ALL_WARPS
        const bool by_lane = wave_whole && !have_warps;
#else
        const bool by_lane = wave_whole;
#endif
        if (by_lane) {
            // (thread index, lane and wave taken from the hardware register again, as values of their own: copies of the kernel's `tid` / `lane` used here
            // would stay in registers across the geometry loop, which has none to spare)
            int t2 = (int)threadIdx.x;
            asm volatile("" : "+v"(t2));
            const int ln = t2 & 63, wv = t2 >> 6;
#pragma unroll
            for (int e = 0; e < 4; ++e) bj_packed |= (unsigned)bj[e] << (8 * e);
            *reinterpret_cast<unsigned*>(&stage.slot[t2 * kCpt]) = bj_packed;
            if (L.warp_channels != 4) {          // the A-grid coordinates of the thread's own cells: computed (or two small loads), parked at once
                int dy, x0;
                lfd_divmod_local(tile_x0 + t2 * kCpt, L.W, L.inv_w, L.w_log2, dy, x0);
                const int y0 = tile_y0 + dy;
                float xa[4], ya;
                if (L.axis_identity) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xa[e] = lfd_axis_value(L.ax, x0 + e);
                    ya = lfd_axis_value(L.ay, y0);
                } else {
                    const float4 ax = load_f32x4(L.axis_x + x0);
                    xa[0] = ax.x; xa[1] = ax.y; xa[2] = ax.z; xa[3] = ax.w;
                    ya = lfd_global(L.axis_y)[y0];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { stage.xyz[3 * (t2 * kCpt + e) + 0] = xa[e]; stage.xyz[3 * (t2 * kCpt + e) + 1] = ya; }
            }
            // (what one lane wrote another lane of the wave reads: in order in the hardware, and kept in order by the compiler)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int wbase = wv * 64 * kCpt + ln;
            int sj[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sj[e] = (int)stage.slot[wbase + 64 * e];
            if (L.warp_channels == 4) {          // upstream's concatenated [xA, yA, xB, yB]: 16 bytes per cell, consecutive lanes read one contiguous KB per plane
                float4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = load_f32x4(S.slot[sj[e]].warp + (size_t)(unsigned)(tile_cell0 + wbase + 64 * e) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sl = wbase + 64 * e;
                    stage.xyz[3 * sl + 0] = v[e].x; stage.xyz[3 * sl + 1] = v[e].y; stage.xyz[3 * sl + 2] = v[e].z;
                    stage.err[sl] = v[e].w;
                }
            } else {
                float2 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = load_f32x2(S.slot[sj[e]].warp + (size_t)(unsigned)(tile_cell0 + wbase + 64 * e) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int sl = wbase + 64 * e;
                    stage.xyz[3 * sl + 2] = v[e].x;
                    stage.err[sl] = v[e].y;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else
#endif
        if (kCpt == 4 && (L.W & 3) == 0 && cell0 + 3 < HW) {
            // W % 4 == 0: the four cells share a row, so one cell -> (row, column) conversion serves all
            float xa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ya = 0.0f;
            if (L.warp_channels != 4) {
                int dy, x0;
                lfd_divmod_local(tile_x0 + tid * kCpt, L.W, L.inv_w, L.w_log2, dy, x0);
                const int y0 = tile_y0 + dy;
                if (L.axis_identity) {     // the matcher's linspace, computed instead of loaded
#pragma unroll
                    for (int e = 0; e < 4; ++e) xa[e] = lfd_axis_value(L.ax, x0 + e);
                    ya = lfd_axis_value(L.ay, y0);
                } else {
                    const float4 ax = load_f32x4(L.axis_x + x0);
                    xa[0] = ax.x; xa[1] = ax.y; xa[2] = ax.z; xa[3] = ax.w;
                    ya = lfd_global(L.axis_y)[y0];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* wp = S.slot[bj[e]].warp;
                float xan, yan, xbn, ybn;
                if (L.warp_channels == 4) {
                    const float4 v = load_f32x4(wp + (size_t)(unsigned)(cell0 + e) * 4);
                    xan = v.x; yan = v.y; xbn = v.z; ybn = v.w;
#if LFD_DENSE_ALL_WARPS
                } else if (have_warps) {
                    xan = xa[e]; yan = ya; xbn = spec_xb[e]; ybn = spec_yb[e];
#endif
                } else {
                    const float2 v = load_f32x2(wp + (size_t)(unsigned)(cell0 + e) * 2);
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
        if (dead) {                               // (rare: masks, or a floor of 0)
#pragma unroll
            for (int e = 0; e < kCpt; ++e)
                if ((dead >> e) & 1u) stage.xyz[3 * (tid * kCpt + e) + 2] = __builtin_nanf("");
        }
        *reinterpret_cast<unsigned*>(&stage.slot[tid * kCpt]) = bj_packed;

        LFD_STAMP(4);
#if LFD_FRONT_PRIO > 0
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- stage 3: per-correspondence geometry + colour; survivors overwrite their slot -------------
        unsigned keep_bits = 0;
#pragma unroll 1
        for (int ev = 0; ev < kCpt; ++ev) {
            // re-read the camera constants from LDS every cell instead of pinning ~60 registers on them
            asm volatile("" ::: "memory");
            const int e = __builtin_amdgcn_readfirstlane(ev);      // the cell counter is the same for every lane: a scalar register
            const int sl = tid * kCpt + e;
            float* sxyz = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(stage.xyz) + lfd_slot_bytes12(sl));
            const float xan = sxyz[0], yan = sxyz[1], xbn = sxyz[2];
            const float ybn = stage.err[sl];
            const int bje = (int)((bj_packed >> (8 * e)) & 0xffu);
            LfdCellResult res;
            res.keep = 0; res.x = res.y = res.z = res.err = res.xa_px = res.ya_px = 0.0f;
#if defined(LFD_ABLATE_EVAL)
            if (cell0 + e < HW) { res.keep = xbn > -0.9f; res.x = xan; res.y = yan; res.z = xbn; res.err = ybn; res.xa_px = 1.0f; res.ya_px = 1.0f; }
#else
            if (cell0 + e < HW) lfd_eval_correspondence(rc, S.pc[bje], xan, yan, xbn, ybn, L.kp, res);
#endif
            if (res.keep) {
                sxyz[0] = res.x; sxyz[1] = res.y; sxyz[2] = res.z;
                stage.err[sl] = res.err;
                keep_bits |= 1u << e;
            }
        }
        LFD_STAMP(5);
#if defined(LFD_BACK_PRIO)
        __builtin_amdgcn_s_setprio(LFD_BACK_PRIO);
#endif
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
        // survivors of the lower lanes (thread-major raster order): one ballot per cell index, counted below the lane
        const unsigned my_cnt = __popc(keep_bits);
        unsigned before = 0, wave_total = 0;
#pragma unroll
        for (int e = 0; e < kCpt; ++e) {
            const u64 m = __ballot((keep_bits >> e) & 1u);
            before += __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            wave_total += (unsigned)__popcll(m);
        }
        const unsigned incl = before + my_cnt;
        if (lane == 0) s_wave_cnt[wave] = wave_total;
        __syncthreads();                          // s_wave_cnt written
        LFD_STAMP(6);
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
                if ((keep_bits >> e) & 1u) stage_order[lpos++] = (unsigned short)(tid * kCpt + e);
        }
        __syncthreads();                          // the order map is complete
        LFD_STAMP(7);
        if (L.seg_counts && tid < ns && s_slot_cnt[tid]) {
            // the workgroup of tile 0 zeroed the array and raised seg_ready when the launch began (no memset launch)
            // (only the first few workgroups of a launch ever have to poll.  The counters are only touched by device-scope
            // atomics, so a relaxed read of the flag is enough.)
            unsigned spins = 0;
            unsigned ready = __hip_atomic_load(L.seg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ready != L.epoch) {
                if (++spins > LFD_SPIN_LIMIT) { atomicExch(L.status, LFD_LAUNCH_TIMEOUT); break; }
                __builtin_amdgcn_s_sleep(8);
                ready = __hip_atomic_load(L.seg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            atomicAdd(&L.seg_counts[(size_t)r * L.k + tid], (int)s_slot_cnt[tid]);
        }

        // ---- stage 5: ordered retirement.  Wave 0 runs the look-back (it publishes the tile's count at once and then
        //      waits for its predecessors' counts); MEANWHILE waves 1..3 evaluate the colours of the tile's survivors
        //      into registers - that work needs the tile-local order only, not the global offset - so the wait of the
        //      look-back is covered by the workgroup's own arithmetic instead of idling all four waves.  After the
        //      barrier the same three waves write their records: consecutive threads write consecutive records, so
        //      every wave-wide store covers one contiguous span of the output arrays. -----------------------------
        // (unordered retirement: nobody runs a look-back, so all four waves share the colour work and the stores)
        constexpr int kCopyThreads = kUnordered ? kBlock : kCopyThreadsOrdered;
        constexpr int kCopyRecords = (kTile + kCopyThreads - 1) / kCopyThreads;
        float rgb[kCopyRecords][3];
        int ctid = (int)threadIdx.x;               // thread index among the copy-out waves; taken from the hardware register again
        asm volatile("" : "+v"(ctid));             // here, so that no copy of it occupies a register across the geometry loop
        if (!kUnordered) ctid -= 64;
        u64 claimed = 0;
        if (kUnordered && tid == 0) {
            // one returning atomic per tile on the reference's cursor (the tiles of a reference are in flight together, a few hundred per
            // reference: ~12 ns each on one address), behind the flag that says tile 0's workgroup has zeroed the cursors.  Issued before
            // the colour work, consumed after it: its round trip is covered
            unsigned spins = 0;
            unsigned ready = __hip_atomic_load(L.seg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ready != L.epoch) {
                if (++spins > LFD_SPIN_LIMIT) { atomicExch(L.status, LFD_LAUNCH_TIMEOUT); break; }
                __builtin_amdgcn_s_sleep(8);
                ready = __hip_atomic_load(L.seg_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            claimed = atomicAdd(L.ref_cursor + r, (u64)block_total);
        }
        if (!kUnordered && wave == 0) {
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
        if (kUnordered || wave != 0) {
#if !defined(LFD_ABLATE_STORES)
            const uint8_t* image = S.ref.image;
            const int n_loc = (int)block_total;
            if (!kExactColour && L.colour_cols) {
                // analytic A-grid, f32 blend: a cell's reference position, first-tap offset and weight factors come from the
                // column / row tables (two 16-byte loads instead of ~60 instructions of axis, floor, clamp and offset arithmetic)
                const unsigned n_bytes = __umul24((unsigned)L.w_match, (unsigned)L.h_match) * 3u;
#pragma unroll
                for (int u0 = 0; u0 < kCopyRecords; u0 += LFD_COPY_UNROLL) {
                    LfdColourCol cc[LFD_COPY_UNROLL];
                    LfdColourRow cr[LFD_COPY_UNROLL];
                    LfdTapRows taps[LFD_COPY_UNROLL];
                    unsigned sh0[LFD_COPY_UNROLL], sh1[LFD_COPY_UNROLL];
                    if (ctid + u0 * kCopyThreads < n_loc) {       // wave-uniform except in the tile's last partial wave
#pragma unroll
                        for (int v = 0; v < LFD_COPY_UNROLL; ++v) {
                            if (u0 + v >= kCopyRecords) break;
                            const int i = ctid + (u0 + v) * kCopyThreads;
                            const int sl = (int)stage_order[i < n_loc ? i : n_loc - 1];
                            int dy, x;
                            lfd_divmod_local(tile_x0 + sl, L.W, L.inv_w, L.w_log2, dy, x);
                            const lfd_u32x4 c4 = *reinterpret_cast<const lfd_u32x4 LFD_GLOBAL_AS*>(lfd_global(L.colour_cols) + x);
                            const lfd_u32x4 r4 = *reinterpret_cast<const lfd_u32x4 LFD_GLOBAL_AS*>(lfd_global(L.colour_rows) + (tile_y0 + dy));
                            cc[v].off = c4.x; cc[v].ax = __uint_as_float(c4.y); cc[v].bx = __uint_as_float(c4.z); cc[v].clamped = c4.w;
                            cr[v].off0 = r4.x; cr[v].off1 = r4.y; cr[v].ay = __uint_as_float(r4.z); cr[v].by = __uint_as_float(r4.w);
                            taps[v] = lfd_bilinear_fetch_tab(image, n_bytes, cc[v], cr[v], sh0[v], sh1[v]);
                        }
#pragma unroll
                        for (int v = 0; v < LFD_COPY_UNROLL; ++v) {
                            if (u0 + v >= kCopyRecords) break;
                            lfd_bilinear_eval_tab(taps[v], sh0[v], sh1[v], cc[v], cr[v], rgb[u0 + v]);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int u0 = 0; u0 < kCopyRecords; u0 += LFD_COPY_UNROLL) {
                // LFD_COPY_UNROLL records per step: their image rows are in flight before the first colour is evaluated
                float px[LFD_COPY_UNROLL], py[LFD_COPY_UNROLL];
                LfdTapRows taps[LFD_COPY_UNROLL];
                unsigned sh0[LFD_COPY_UNROLL], sh1[LFD_COPY_UNROLL];
                if (ctid + u0 * kCopyThreads < n_loc) {       // wave-uniform except in the tile's last partial wave
#pragma unroll
                    for (int v = 0; v < LFD_COPY_UNROLL; ++v) {
                        if (u0 + v >= kCopyRecords) break;
                        const int i = ctid + (u0 + v) * kCopyThreads;
                        const int sl = (int)stage_order[i < n_loc ? i : n_loc - 1];
                        {   // reference position in match pixels, from the A-grid coordinates of the cell (not staged: LDS is
                            // one of the two things that limit the number of resident workgroups)
                            const int cell = tile_cell0 + sl;
                            float xan, yan;
                            if (L.warp_channels == 4) {
                                const float2 w2 = load_f32x2(S.slot[stage.slot[sl]].warp + (size_t)cell * 4);
                                xan = w2.x; yan = w2.y;
                            } else {
                                int dy, x;
                                lfd_divmod_local(tile_x0 + sl, L.W, L.inv_w, L.w_log2, dy, x);
                                const int y = tile_y0 + dy;
                                if (L.axis_identity) { xan = lfd_axis_value(L.ax, x); yan = lfd_axis_value(L.ay, y); }
                                else { xan = lfd_global(L.axis_x)[x]; yan = lfd_global(L.axis_y)[y]; }
                            }
                            px[v] = lfd_match_px(xan, L.kp.wm1); py[v] = lfd_match_px(yan, L.kp.hm1);
                        }
                        taps[v] = lfd_bilinear_fetch(image, L.w_match, L.h_match, px[v], py[v], sh0[v], sh1[v]);
                    }
#pragma unroll
                    for (int v = 0; v < LFD_COPY_UNROLL; ++v) {
                        if (u0 + v >= kCopyRecords) break;
#if defined(LFD_ABLATE_COLOUR)
                        rgb[u0 + v][0] = px[v]; rgb[u0 + v][1] = py[v]; rgb[u0 + v][2] = (float)(taps[v].r0 + taps[v].r1 + sh0[v] + sh1[v]);
#else
                        if (kExactColour) lfd_bilinear_eval(taps[v], sh0[v], sh1[v], L.w_match, L.h_match, px[v], py[v], rgb[u0 + v]);
                        else lfd_bilinear_eval_f32(taps[v], sh0[v], sh1[v], L.w_match, L.h_match, px[v], py[v], rgb[u0 + v]);
#endif
                    }
                }
            }
            }
#endif
        }
        if (kUnordered && tid == 0) {
            s_tile_excl = (u64)(unsigned)r * (u64)(unsigned)HW + claimed;
            LfdTileSeg seg; seg.offset = (int32_t)claimed; seg.count = (int32_t)block_total;
            L.tile_table[tile] = seg;
        }
        LFD_STAMP(8);
        __syncthreads();                          // prefix known, colours in registers
        LFD_STAMP(9);

#if !defined(LFD_ABLATE_STORES)
        if (kUnordered || wave != 0) {
            // the tile's offset is the same for every lane: kept in scalar registers, so that a record's address is a scalar base
            // plus a 32-bit per-lane byte offset (no 64-bit vector multiply-adds: those run at a quarter of the rate)
            const u64 excl = s_tile_excl;
            const long long base = (long long)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(excl >> 32)) << 32) |
                                               (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)excl));
            long long room = L.capacity - base;          // beyond capacity: counted, not written
            int n = (int)block_total;
            if (room < (long long)n) n = room > 0 ? (int)room : 0;
            if (kPly) {
                // one record per thread and step: 12 bytes of position + 3 bytes of colour at byte 15 (base + i) - unaligned 12 / 2 / 1-byte
                // stores; a wave's 64 records are 960 contiguous bytes of the file payload
                unsigned char* gp = L.ply + 15 * base;
#pragma unroll
                for (int u = 0; u < kCopyRecords; ++u) {
                    const int i = ctid + u * kCopyThreads;
                    if (i < n) {
                        const int sl = (int)stage_order[i];
                        const float* sxyz = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(stage.xyz) + lfd_slot_bytes12(sl));
                        struct __attribute__((packed, aligned(1))) F3u { float a, b, c; };
                        struct __attribute__((packed, aligned(1))) U16u { unsigned short v; };
                        F3u p; p.a = sxyz[0]; p.b = sxyz[1]; p.c = sxyz[2];
                        const unsigned o15 = (unsigned)i * 15u;
                        const unsigned q0 = lfd_quantise_u8(rgb[u][0]), q1 = lfd_quantise_u8(rgb[u][1]), q2 = lfd_quantise_u8(rgb[u][2]);
                        *reinterpret_cast<F3u*>(gp + o15) = p;
                        U16u rg; rg.v = (unsigned short)(q0 | (q1 << 8));
                        *reinterpret_cast<U16u*>(gp + o15 + 12u) = rg;
                        gp[o15 + 14u] = (unsigned char)q2;
                        if (L.cell) L.cell[base + i] = tile_cell0 + sl;
                        if (L.slot) L.slot[base + i] = stage.slot[sl];
                    }
                }
            } else {
            unsigned char* gx = reinterpret_cast<unsigned char*>(L.xyz + 3 * base);
            unsigned char* gc = reinterpret_cast<unsigned char*>(L.rgb + 3 * base);
            unsigned char* ge = reinterpret_cast<unsigned char*>(L.err + base);
#pragma unroll
            for (int u = 0; u < kCopyRecords; ++u) {
                const int i = ctid + u * kCopyThreads;
                if (i < n) {
                    const int sl = (int)stage_order[i];
                    LfdF3 p, c;
                    const float* sxyz = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(stage.xyz) + lfd_slot_bytes12(sl));
                    p.a = sxyz[0]; p.b = sxyz[1]; p.c = sxyz[2];
                    c.a = rgb[u][0]; c.b = rgb[u][1]; c.c = rgb[u][2];
                    const unsigned o12 = (unsigned)lfd_slot_bytes12(i), o4 = (unsigned)i * 4u;
#if LFD_NT_STORES
                    // (component stores: the compiler merges them into one dwordx3 with the nt bit)
                    float* fx = reinterpret_cast<float*>(gx + o12);
                    float* fc = reinterpret_cast<float*>(gc + o12);
                    __builtin_nontemporal_store(p.a, fx); __builtin_nontemporal_store(p.b, fx + 1); __builtin_nontemporal_store(p.c, fx + 2);
                    __builtin_nontemporal_store(c.a, fc); __builtin_nontemporal_store(c.b, fc + 1); __builtin_nontemporal_store(c.c, fc + 2);
                    __builtin_nontemporal_store(stage.err[sl], reinterpret_cast<float*>(ge + o4));
#else
                    *reinterpret_cast<LfdF3*>(gx + o12) = p;
                    *reinterpret_cast<LfdF3*>(gc + o12) = c;
                    *reinterpret_cast<float*>(ge + o4) = stage.err[sl];
#endif
                    if (L.cell) L.cell[base + i] = tile_cell0 + sl;
                    if (L.slot) L.slot[base + i] = stage.slot[sl];
                }
            }
        }
#endif
            }
#if defined(LFD_DENSE_TIMING)
        LFD_STAMP(10);
        if (L.phase_stamps && lane == 0 && wave < 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores retired: the wave could end here
            L.phase_stamps[((size_t)blockIdx.x * 2 + wave) * 12 + 11] = __builtin_readcyclecounter();
        }
#endif
    }
}

extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_kernel(LfdLaunch L) { lfd_dense_body<false>(L); }
// the same kernel with upstream's f64 colour arithmetic (bit-identical rgb; lfd_params.flags & LFD_FLAG_EXACT_COLOUR)
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_exact_kernel(LfdLaunch L) { lfd_dense_body<true>(L); }
// unordered retirement (lfd_triangulate_dense_segments), both colour forms
// file-payload output (lfd_triangulate_dense_ply), both colour forms
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_ply_kernel(LfdLaunch L) { lfd_dense_body<false, false, true>(L); }
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_ply_exact_kernel(LfdLaunch L) { lfd_dense_body<true, false, true>(L); }
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_segments_kernel(LfdLaunch L) { lfd_dense_body<false, true>(L); }
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_segments_exact_kernel(LfdLaunch L) { lfd_dense_body<true, true>(L); }
// ... and both at once (lfd_triangulate_dense_ply_segments): the file payload without a look-back, for a consumer that wants the point SET of every
// reference, not its raster order
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_ply_segments_kernel(LfdLaunch L) { lfd_dense_body<false, true, true>(L); }
extern "C" __global__ void __launch_bounds__(kBlock, LFD_DENSE_WAVES_PER_SIMD) lfd_dense_ply_segments_exact_kernel(LfdLaunch L) { lfd_dense_body<true, true, true>(L); }

// =================================================================================================
// indexed mode, pass A on the whole chip: every selected cell is evaluated by its own thread (256 cells per workgroup,
// grid = chunks x references); per-(reference, slot) survivor counts and first appearances are collected in `tab`
// ([ref][slot]{count, 0xffffffff - first}, zeroed before the launch) for lfd_indexed_kernel, which then only orders and
// scatters.
// =================================================================================================
// sel_offsets: [n_refs + 1] (reference r owns [off[r], off[r+1])) or, with off_pairs, [2 * n_refs] = {begin, end} per reference
// (the fused multi-reference call: every reference's selection sits at a fixed stride, its length known only on the device)
extern "C" __global__ void __launch_bounds__(LFD_INDEXED_EVAL_BLOCK) lfd_indexed_eval_kernel(LfdLaunch L, const long long* __restrict__ sel_idx,
                                                                                             const long long* __restrict__ sel_offsets,
                                                                                             float* __restrict__ scratch, uint8_t* __restrict__ codes,
                                                                                             unsigned* __restrict__ tab, int off_pairs) {
    __shared__ BlockShared S;
    __shared__ unsigned s_cnt[LFD_MAX_SLOTS];
    __shared__ unsigned s_first[LFD_MAX_SLOTS];
    const int tid = (int)threadIdx.x;
    const int r = (int)blockIdx.y;
    const long long sel_begin = off_pairs ? sel_offsets[2 * r] : sel_offsets[r], sel_end = off_pairs ? sel_offsets[2 * r + 1] : sel_offsets[r + 1];
    const int n_sel = (int)(sel_end - sel_begin);
    const int i = (int)blockIdx.x * LFD_INDEXED_EVAL_BLOCK + tid;
    if ((int)blockIdx.x * LFD_INDEXED_EVAL_BLOCK >= n_sel) return;
    if (tid < LFD_MAX_SLOTS) { s_cnt[tid] = 0; s_first[tid] = 0xffffffffu; }
    block_prologue(L, r, S);
    const int HW = L.H * L.W;
    if (i < n_sel) {
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
    if (tid < LFD_MAX_SLOTS) {
        unsigned* t = tab + ((size_t)r * LFD_MAX_SLOTS + tid) * 2;
        if (s_cnt[tid]) atomicAdd(t + 0, s_cnt[tid]);
        if (s_first[tid] != 0xffffffffu) atomicMax(t + 1, 0xffffffffu - s_first[tid]);
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
                                                                                   int32_t* __restrict__ seg_order, const unsigned* __restrict__ tab, int off_pairs) {
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
    const long long sel_begin = off_pairs ? sel_offsets[2 * r] : sel_offsets[r], sel_end = off_pairs ? sel_offsets[2 * r + 1] : sel_offsets[r + 1];
    const int n_sel = (int)(sel_end - sel_begin);
    const int HW = L.H * L.W;

    // ---- pass A (skipped when lfd_indexed_eval_kernel has run: `tab` then holds the per-slot counts and first appearances) ----
    if (tab) {
        if (tid < LFD_MAX_SLOTS) {
            s_cnt[tid] = tab[((size_t)r * LFD_MAX_SLOTS + tid) * 2 + 0];
            s_first[tid] = 0xffffffffu - tab[((size_t)r * LFD_MAX_SLOTS + tid) * 2 + 1];
        }
    }
    for (int i = tid; i < (tab ? 0 : n_sel); i += LFD_INDEXED_BLOCK) {
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
