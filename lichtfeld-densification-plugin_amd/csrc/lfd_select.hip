// S: coverage sampling of the aggregated certainty map, on the device
// (upstream core/sampling.py:8-53 = torch clamp/sum + legacy numpy.random.choice + argsort walk).
//
// Two kernels with identical results: lfd_select_filter_kernel runs the whole selection in one 1024-thread workgroup
// (small maps, profiling); lfd_select_filter_mw_kernel shares the streaming passes and the searches out over several workgroups
// that meet at grid barriers, with one more workgroup that PRODUCES the (sequential) MT19937 stream as an array of doubles.  A launch
// of the second kernel may cover several references ON ONE STREAM (lfd_triangulate_sampled_chain): every reference draws from where
// the one before it stopped, everything else of their selections runs side by side.  The problem is ~260k weights
// and ~9k draws per reference: latency and one CU's memory pipeline bound it, not the chip's bandwidth.
//
// Restated third-party algorithms (absent from /root/reference; NumPy 2.2.6 numpy/random/mtrand.pyx
// `RandomState.choice(a, size, replace=False, p)` and `_legacy_seeding`, randomkit's MT19937):
//   * MT19937: init_genrand(seed) (Knuth multiplier 1812433253), 624-word twist, tempering;
//     random_sample() = ((a >> 5) * 67108864 + (b >> 6)) / 2^53 from two successive outputs a, b.
//   * choice without replacement: repeat { x = random_sample(size - n_uniq); p[found] = 0;
//     cdf = cumsum(p) / cdf[-1]; new = searchsorted(cdf, x, side="right"); keep the first occurrence
//     of every distinct value, in draw order; append } until `size` distinct cells are found.
//   * the f64 cumsum is reproduced exactly although it is evaluated as a parallel scan: every p_i is
//     an f32 value >= 2^-29 or exactly 0 (certainties are floored at certainty_thresh before they get
//     here), so every partial sum below 1.0 is a multiple of 2^-52 and NO addition rounds - the
//     result does not depend on the order of the additions.  Inputs that violate the precondition are
//     detected (LFD_SELECT_INEXACT) and the caller falls back to an ordered single-lane scan.
//   * upstream's normaliser `s = weights.sum()` is a torch CPU f32 reduction whose rounding depends
//     on the host's thread count and vector ISA; here s = RN_f32(exact sum), which is what every
//     such order approximates.  Given the same s the drawn cells are identical to upstream's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfd_device.hpp"

namespace {

constexpr int kSelBlock = LFD_SELECT_BLOCK;
#ifndef LFD_PRODUCER_PIECE
#define LFD_PRODUCER_PIECE 4096
#endif
constexpr int kScanRoundDraws = 1024;     // multi-workgroup kernel: a later round with at most this many draws left searches the weights themselves

// ---- MT19937 --------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// One twist = the next 624-word state.  With separate source and destination arrays the three dependency-free segments
// need only two barriers between them: new[i] = new-or-old[(i + 397) % 624] ^ mix(old[i], old-or-new[(i + 1) % 624]).
__device__ void mt_twist_lds(const unsigned* o, unsigned* n, int tid) {
    auto mix = [](unsigned a, unsigned b) { const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu); return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
    if (tid < 227) n[tid] = o[tid + 397] ^ mix(o[tid], o[tid + 1]);                       // i in [0,227): old words only
    __syncthreads();
    if (tid < 227) { const int i = tid + 227; n[i] = n[i - 227] ^ mix(o[i], o[i + 1]); }    // i in [227,454): new[i-227] from the first segment
    __syncthreads();
    if (tid < 169) { const int i = tid + 454; n[i] = n[i - 227] ^ mix(o[i], o[i + 1]); }    // i in [454,623): new[i-227] from the second segment
    if (tid == 256) n[623] = n[396] ^ mix(o[623], n[0]);                                     // i = 623 wraps to the NEW word 0
    __syncthreads();
}

// draws[0..n) = next n legacy random_sample() doubles; mt[624] = position.  The state is staged in LDS for the whole
// call (a twist in global memory costs a memory round trip per segment) and written back at the end.
__device__ void mt_fill_doubles(unsigned* mt, double* draws, int n, int tid) {
    __shared__ unsigned s_mt[2][624];
    __shared__ int s_pos;
    __syncthreads();
    for (int i = tid; i < 624; i += kSelBlock) s_mt[0][i] = mt[i];
    if (tid == 0) s_pos = (int)mt[624];
    __syncthreads();
    int cur = 0, pos = s_pos, produced = 0;
    while (produced < n) {
        if (pos >= 624) {
            mt_twist_lds(s_mt[cur], s_mt[cur ^ 1], tid);
            cur ^= 1;
            pos = 0;
        }
        const int avail = 624 - pos;
        const int want_words = 2 * (n - produced);
        const int take = avail < want_words ? avail : want_words;
        const int pairs = take >> 1;
        for (int i = tid; i < pairs; i += kSelBlock) {
            const unsigned a = mt_temper(s_mt[cur][pos + 2 * i]), b = mt_temper(s_mt[cur][pos + 2 * i + 1]);
            draws[produced + i] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
        }
        if (take & 1) {
            // an odd number of words was left before the twist: the last one pairs with the first word of the next state
            const unsigned a = mt_temper(s_mt[cur][pos + take - 1]);
            __syncthreads();
            mt_twist_lds(s_mt[cur], s_mt[cur ^ 1], tid);
            cur ^= 1;
            if (tid == 0) {
                const unsigned b = mt_temper(s_mt[cur][0]);
                draws[produced + pairs] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
            }
            pos = 1;
            produced += pairs + 1;
        } else {
            pos += take;
            produced += pairs;
        }
        __syncthreads();       // the words just read may be overwritten by the next twist's other buffer only after everyone has read them
    }
    for (int i = tid; i < 624; i += kSelBlock) mt[i] = s_mt[cur][i];
    if (tid == 0) mt[624] = (unsigned)pos;
    __syncthreads();
}

// The whole stream as an array: doubles `from` .. `to` (absolute indices) of the legacy stream go to ring[i & mask]; the key lives in LDS across
// calls (`cur`, `pos`, `twists`: the caller's), and every key a twist produces is also kept in `snaps` (slot = number of the twist % slots): the
// producer runs AHEAD of what is consumed, so the key it ends with is not the one to commit - that one is looked up by the position the
// consumers stopped at.
struct MtCursor { int cur, pos; unsigned twists; };

__device__ void mt_twist_and_keep(unsigned (*s_mt)[624], MtCursor& c, unsigned* snaps, int snap_slots, int tid) {
    mt_twist_lds(s_mt[c.cur], s_mt[c.cur ^ 1], tid);
    c.cur ^= 1;
    ++c.twists;
    unsigned* dst = snaps + (size_t)(c.twists % (unsigned)snap_slots) * 624u;
    for (int i = tid; i < 624; i += kSelBlock) dst[i] = s_mt[c.cur][i];
}

__device__ void mt_stream_fill(unsigned (*s_mt)[624], MtCursor& c, double* ring, unsigned long long mask, unsigned long long from,
                               unsigned long long to, unsigned* snaps, int snap_slots, int tid) {
    unsigned long long produced = from;
    while (produced < to) {
        if (c.pos >= 624) { mt_twist_and_keep(s_mt, c, snaps, snap_slots, tid); c.pos = 0; }
        const int avail = 624 - c.pos;
        const unsigned long long left = to - produced;
        const int want_words = left > 312ull ? 624 : 2 * (int)left;
        const int take = avail < want_words ? avail : want_words;
        const int pairs = take >> 1;
        for (int i = tid; i < pairs; i += kSelBlock) {
            const unsigned a = mt_temper(s_mt[c.cur][c.pos + 2 * i]), b = mt_temper(s_mt[c.cur][c.pos + 2 * i + 1]);
            ring[(produced + (unsigned long long)i) & mask] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
        }
        if (take & 1) {       // the last word of this key pairs with the first word of the next one
            const unsigned a = mt_temper(s_mt[c.cur][c.pos + take - 1]);
            __syncthreads();
            mt_twist_and_keep(s_mt, c, snaps, snap_slots, tid);
            if (tid == 0) {
                const unsigned b = mt_temper(s_mt[c.cur][0]);
                ring[(produced + (unsigned long long)pairs) & mask] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
            }
            c.pos = 1;
            produced += (unsigned long long)pairs + 1ull;
        } else {
            c.pos += take;
            produced += (unsigned long long)pairs;
        }
        __syncthreads();
    }
}

// The producer workgroup of a launch.  It stays ahead of the highest index anybody has asked for - by `look` doubles while another reference is
// still to come (its first round needs exactly that many), by a later round's worth behind the launch's last reference - never more than the
// ring holds beyond what has been released, in pieces short enough to notice soon that the last reference has said where it stopped; then it
// commits the stream AT THAT POSITION: the key of the twist the position lies in, from `snaps`.
__device__ void mt_stream_producer(const LfdSelectArgs& A, int n_refs, int first_round, int tid) {
    // while another reference follows it stays M doubles ahead of the first round of the one at work (the follower looks that window up in advance)
    // (twice M: a follower marks the cells of its first round while the reference at work is in its second - the doubles of that round, which start
    // up to M behind the ones asked for, have to be there already; the ring holds four first rounds)
    const int look = n_refs > 1 ? 2 * (A.M > first_round ? A.M : first_round) : first_round;
    __shared__ unsigned s_key[2][624];
    __shared__ unsigned long long s_tgt;
    __shared__ int s_cmd, s_pos0;
    unsigned long long* produced_w = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_PRODUCED);
    unsigned long long* want_w = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_WANT);
    unsigned long long* released_w = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_RELEASED);
    unsigned* state_w = reinterpret_cast<unsigned*>(A.chain + LFD_CHAIN_STATE);
    unsigned* current_w = reinterpret_cast<unsigned*>(A.chain + LFD_CHAIN_CURRENT);
    unsigned long long* off_w = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_OFF);
    for (int i = tid; i < 624; i += kSelBlock) s_key[0][i] = A.mt[i];
    if (tid == 0) s_pos0 = (int)A.mt[624];
    __syncthreads();
    const int pos0 = s_pos0;
    MtCursor c = {0, pos0, 0u};
    const unsigned long long mask = (unsigned long long)A.ring_cap - 1ull;
    unsigned long long produced = 0ull, last = 0ull;
    while (true) {
        if (tid == 0) {
            int cmd = 3;
            unsigned spins = 0;
            while (++spins < (1u << 23)) {
                last = __hip_atomic_load(off_w + n_refs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (last != 0ull) { cmd = 2; break; }
                const unsigned long long want = __hip_atomic_load(want_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long rel = __hip_atomic_load(released_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned current = __hip_atomic_load(current_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (nothing asked for yet: the launch's first reference will want a whole first round)
                const int ahead = (want == 0ull || (int)current + 1 < n_refs) ? look : (look < kScanRoundDraws ? look : kScanRoundDraws);
                unsigned long long tgt = want + (unsigned long long)ahead;
                if (tgt > rel + (unsigned long long)A.ring_cap) tgt = rel + (unsigned long long)A.ring_cap;
                // (pieces: short behind the launch's last reference - the commit waits for the piece at hand - longer while others follow, where
                // every look at the chain's words costs the followers' lookups a microsecond)
                const unsigned long long piece = ((int)current + 1 < n_refs) ? (unsigned long long)LFD_PRODUCER_PIECE : 2048ull;
                if (tgt > produced + piece) tgt = produced + piece;
                if (tgt > produced) { cmd = 1; s_tgt = tgt; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            s_cmd = cmd;
            if (cmd == 2) s_tgt = last;
        }
        __syncthreads();
        const int cmd = s_cmd;
        const unsigned long long tgt = s_tgt;
        __syncthreads();
        if (cmd == 1) {
            mt_stream_fill(s_key, c, A.ring, mask, produced, tgt, A.snaps, A.snap_slots, tid);
            produced = tgt;
            if (tid == 0) { __threadfence(); __hip_atomic_store(produced_w, produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            continue;
        }
        if (cmd == 2 && tgt != LFD_CHAIN_BROKEN) {
            // tgt = 1 + doubles consumed by the whole launch; the word after them is word W of the stream counted from word 0 of the key the
            // launch started with.  numpy twists lazily: W <= 624 stays in that key; otherwise the key after k = (W - 1) / 624 twists, position W - 624 k
            // (1 ... 624)
            const unsigned long long Wd = (unsigned long long)pos0 + 2ull * (tgt - 1ull);
            if (Wd <= 624ull) {
                if (tid == 0) A.mt[624] = (unsigned)Wd;
            } else {
                const unsigned long long k = (Wd - 1ull) / 624ull;
                const unsigned* src = A.snaps + (size_t)(k % (unsigned long long)A.snap_slots) * 624u;
                for (int i = tid; i < 624; i += kSelBlock) A.mt[i] = src[i];
                if (tid == 0) A.mt[624] = (unsigned)(Wd - 624ull * k);
            }
            if (tid == 0) __hip_atomic_store(state_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) __hip_atomic_store(state_w, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // broken chain, or nobody spoke: nothing is committed
        return;
    }
}

// ---- workgroup-wide helpers --------------------------------------------------------------------------
__device__ double block_sum_f64(double v, double* s_tmp, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_tmp[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < kSelBlock / 64; ++w) t += s_tmp[w];
    return t;
}

__device__ int block_sum_i32(int v, int* s_tmp, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_tmp[tid >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < kSelBlock / 64; ++w) t += s_tmp[w];
    return t;
}

// exclusive scan of one int per thread, returns (exclusive prefix, workgroup total)
__device__ int block_excl_scan_i32(int v, int* s_tmp, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(incl, off, 64);
        if (lane >= off) incl += n;
    }
    __syncthreads();
    if (lane == 63) s_tmp[wave] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
    for (int w = 0; w < kSelBlock / 64; ++w) {
        if (w < wave) base += s_tmp[w];
        total += s_tmp[w];
    }
    return base + incl - v;
}

}  // namespace

// =================================================================================================
// the arguments of this workgroup's reference (blockIdx.y) in a launch that covers several references
__device__ __forceinline__ LfdSelectArgs lfd_select_args_of(LfdSelectArgs A) {
    const long long y = (long long)blockIdx.y;
    if (A.batch_info) { A.n_out = A.batch_info + 2 * y; A.status = A.n_out + 1; }
    if (y == 0) return A;
    const long long sb = A.batch_scratch_stride * y;
    A.best_cert += A.batch_cert_stride * y;
    A.weights = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(A.weights) + sb);
    A.p = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(A.p) + sb);
    A.cdf = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(A.cdf) + sb);
    A.first = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(A.first) + sb);
    A.mark += sb;
    A.draws = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(A.draws) + sb);
    A.cand = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(A.cand) + sb);
    A.found = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(A.found) + sb);
    if (A.coop) A.coop += sb;
    if (!A.batch_info) {
        A.n_out = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(A.n_out) + sb);
        A.status = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(A.status) + sb);
    }
    A.mt += A.batch_mt_stride * y;
    if (A.batch_chain_stride) {
        A.chain += A.batch_chain_stride * y;
        A.ring = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(A.ring) + A.batch_chain_stride * y);
        A.snaps = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(A.snaps) + A.batch_chain_stride * y);
    }
    A.sel_out += A.batch_out_stride * y;
    if (A.sel_offsets_out) A.sel_offsets_out += 2 * y;
    return A;
}

extern "C" __global__ void __launch_bounds__(LFD_SELECT_BLOCK) lfd_select_filter_kernel(LfdSelectArgs A_launch) {
    const LfdSelectArgs A = lfd_select_args_of(A_launch);
    __shared__ double s_d[kSelBlock / 64];
    __shared__ int s_i[kSelBlock / 64];
    __shared__ double s_chunk[kSelBlock];
    __shared__ unsigned long long s_bin[LFD_SELECT_MAX_BINS];
    __shared__ int s_n_uniq;
    __shared__ double s_span[kSelBlock / 64];     // per-wave span: sum of the p that has not been drawn yet

    const int tid = (int)threadIdx.x;
    const int H = A.H, W = A.W, N = H * W;
    const float* cert = A.best_cert;
    float* wbuf = A.weights;
    double* cdf = A.cdf;

    if (tid == 0) { *A.n_out = 0; *A.status = LFD_SELECT_OK; }
    int t_slot = 0;
#define LFD_SEL_STAMP() do { if (A.timing && tid == 0 && t_slot < 32) A.timing[t_slot] = wall_clock64(); ++t_slot; } while (0)
    LFD_SEL_STAMP();

    // ---- weights, exact sum, NaN check --------------------------------------------------------------
    // The streaming passes below move 16 bytes per lane and keep several loads in flight: one workgroup has only its
    // own 16 waves to cover the memory latency, so bytes in flight per lane are what sets its bandwidth.
    const bool vec4 = ((N & 3) == 0) && ((W & 3) == 0) && ((reinterpret_cast<uintptr_t>(cert) & 15u) == 0);
    double acc = 0.0;
    int bad = 0;
    if (vec4) {
#pragma unroll 4
        for (int g = tid; g < (N >> 2); g += kSelBlock) {
            const int i = g << 2;
            const int y = i / W, x = i - y * W;                    // W % 4 == 0: the four cells share a row
            const float4 c4 = *reinterpret_cast<const float4*>(cert + i);
            const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
            float ws[4];
            const bool row_in = y >= A.border && y <= H - 1 - A.border;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float c = cs[e];
                c = (c > A.cap) ? A.cap : c;                       // torch.clamp(max=cap); NaN stays
                const bool inside = row_in && (x + e) >= A.border && (x + e) <= W - 1 - A.border;
                const float w = c * (inside ? 1.0f : 0.0f);
                ws[e] = w;
                if (w != w) bad = 1;
                acc += (double)w;
            }
            *reinterpret_cast<float4*>(wbuf + i) = make_float4(ws[0], ws[1], ws[2], ws[3]);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < N; i += kSelBlock) {
            const int y = i / W, x = i - y * W;
            float c = cert[i];
            c = (c > A.cap) ? A.cap : c;                               // torch.clamp(max=cap); NaN stays
            const bool inside = x >= A.border && x <= W - 1 - A.border && y >= A.border && y <= H - 1 - A.border;
            const float w = c * (inside ? 1.0f : 0.0f);
            wbuf[i] = w;
            if (w != w) bad = 1;
            acc += (double)w;
        }
    }
    LFD_SEL_STAMP();      // 1: weights pass
    const double s64 = block_sum_f64(acc, s_d, tid);
    const int any_bad = block_sum_i32(bad, s_i, tid);
    const float s32 = (A.s_override > 0.0f) ? A.s_override : (float)s64;
    if (any_bad) { if (tid == 0) *A.status = LFD_SELECT_NAN; return; }
    if (!(s32 > 0.0f)) return;                                   // upstream: `if s <= 0: return empty`

    // ---- p = (weights / s) as f32, widened; exactness precondition; non-zero count -------------------------
    // Every wave owns one contiguous span of the map (the spans of the cumulative sum below) and leaves the span's sum of
    // p in LDS: the first pass of the two-pass scan is then never needed - later iterations subtract what was drawn.
    int nz = 0, inexact = 0;
    constexpr int nwaves = kSelBlock / 64;
    const int span = ((N + nwaves - 1) / nwaves + 255) & ~255;    // multiple of 256: a lane owns 4 consecutive cells per step
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int w_lo = min(wave * span, N), w_hi = min(w_lo + span, N);
        double part = 0.0;
#pragma unroll 4
        for (int i = w_lo + 4 * lane; i < w_hi; i += 256) {
            float pf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (i + 3 < w_hi) {
                const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
                pf[0] = w4.x / s32; pf[1] = w4.y / s32; pf[2] = w4.z / s32; pf[3] = w4.w / s32;
                *reinterpret_cast<float4*>(wbuf + i) = make_float4(pf[0], pf[1], pf[2], pf[3]);   // the normalised f32 weights
            } else {
                for (int e = 0; e < 4 && i + e < w_hi; ++e) { pf[e] = wbuf[i + e] / s32; wbuf[i + e] = pf[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (pf[e] > 0.0f) { ++nz; if (pf[e] < 1.862645149230957e-09f) inexact = 1; }   // 2^-29
                if (pf[e] < 0.0f) bad = 1;
            }
            part += ((double)pf[0] + (double)pf[1]) + ((double)pf[2] + (double)pf[3]);     // exact (see header) unless `inexact`
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) s_span[wave] = part;
    }
    LFD_SEL_STAMP();      // 2: p pass
    nz = block_sum_i32(nz, s_i, tid);
    inexact = block_sum_i32(inexact, s_i, tid);
    const int neg = block_sum_i32(bad, s_i, tid);
    const int size = min((int)((double)A.M * 0.85), N);           // int(M * 0.85), f64 product like Python
    if (neg) { if (tid == 0) *A.status = LFD_SELECT_NEGATIVE; return; }
    if (nz < size) { if (tid == 0) *A.status = LFD_SELECT_FEWER_NONZERO; return; }
    if (inexact) { if (tid == 0) *A.status = LFD_SELECT_INEXACT; return; }

    // ---- legacy choice(replace=False, p) -----------------------------------------------------------------------
    // p is never materialised: p[i] = mark[i] ? 0 : (double)weights[i], where mark[] flags the cells drawn so far (the
    // same array later receives the coverage picks and drives the final np.unique)
    unsigned char* mark = A.mark;
    {
        const int n16 = N >> 4;                                   // the scratch array is 256-byte aligned
        for (int g = tid; g < n16; g += kSelBlock) reinterpret_cast<uint4*>(mark)[g] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = (n16 << 4) + tid; i < N; i += kSelBlock) mark[i] = 0;
    }
    if (tid == 0) s_n_uniq = 0;
    __syncthreads();
    const int per = (N + kSelBlock - 1) / kSelBlock;              // run length of the coarse search table / final compaction
    int guard = 0;
    int n_marked = 0;                                             // found[0 .. n_marked) are flagged and subtracted already
    while (true) {
        const int n_uniq = s_n_uniq;
        if (n_uniq >= size) break;
        if (++guard > 64) { if (tid == 0) *A.status = LFD_SELECT_NO_PROGRESS; return; }
        const int need = size - n_uniq;
        mt_fill_doubles(A.mt, A.draws, need, tid);
        LFD_SEL_STAMP();  // per iteration: draws
        for (int j = n_marked + tid; j < n_uniq; j += kSelBlock) {                  // p[found] = 0
            const int c = A.found[j];
            mark[c] = 1;
            atomicAdd(&s_span[c / span], -(double)wbuf[c]);                         // exact, hence order-independent
        }
        n_marked = n_uniq;
        __syncthreads();
        // cdf = cumsum(p) (exact, see header), then /= cdf[-1].  Each wave owns one contiguous span and
        // walks it 64 elements at a time (coalesced), scanning inside the wave with shuffles.
        {
            const int lane = tid & 63, wave = tid >> 6;
            const int w_lo = min(wave * span, N), w_hi = min(w_lo + span, N);
            double carry = 0.0, total = 0.0;
            for (int w = 0; w < nwaves; ++w) { if (w < wave) carry += s_span[w]; total += s_span[w]; }
            // cdf /= cdf[-1]: 262144 IEEE f64 divisions (~35 instructions each) made this pass ALU-bound on the one CU it runs
            // on.  With r = RN(1/total), q = RN(a r), q' = RN(q + r (a - total q)) IS the correctly rounded a / total (Markstein;
            // the one exception, a divisor whose significand is all ones, takes the division)
            const double rtot = 1.0 / total;
            const bool quick_div = (__double_as_longlong(total) & 0xfffffffffffffll) != 0xfffffffffffffll;
            auto div_total = [&](double a) {
                if (quick_div) { const double q = a * rtot; return fma(fma(-total, q, a), rtot, q); }
                return a / total;
            };
            for (int base = w_lo; base < w_hi; base += 256) {
                const int i = base + 4 * lane;
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                const bool full = i + 3 < w_hi;
                if (full) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
                    const unsigned m4 = *reinterpret_cast<const unsigned*>(mark + i);
                    v0 = (m4 & 0xffu) ? 0.0 : (double)w4.x; v1 = (m4 & 0xff00u) ? 0.0 : (double)w4.y;
                    v2 = (m4 & 0xff0000u) ? 0.0 : (double)w4.z; v3 = (m4 & 0xff000000u) ? 0.0 : (double)w4.w;
                } else {
                    if (i < w_hi) v0 = mark[i] ? 0.0 : (double)wbuf[i];
                    if (i + 1 < w_hi) v1 = mark[i + 1] ? 0.0 : (double)wbuf[i + 1];
                    if (i + 2 < w_hi) v2 = mark[i + 2] ? 0.0 : (double)wbuf[i + 2];
                }
                v1 += v0; v2 += v1; v3 += v2;                  // inclusive prefix inside the lane (every sum is exact, see header)
                double v = v3;                                  // ... and across the lanes
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double n = __shfl_up(v, off, 64);
                    if (lane >= off) v += n;
                }
                const double before = carry + (v - v3);        // exact: sum of everything ahead of this lane's first cell
                if (full) {
                    *reinterpret_cast<double2*>(cdf + i) = make_double2(div_total(before + v0), div_total(before + v1));
                    *reinterpret_cast<double2*>(cdf + i + 2) = make_double2(div_total(before + v2), div_total(before + v3));
                } else {
                    if (i < w_hi) cdf[i] = div_total(before + v0);
                    if (i + 1 < w_hi) cdf[i + 1] = div_total(before + v1);
                    if (i + 2 < w_hi) cdf[i + 2] = div_total(before + v2);
                }
                carry += __shfl(v, 63, 64);
            }
        }
        __syncthreads();
        LFD_SEL_STAMP();  // cumsum
        // coarse table for a two-level search: s_chunk[k] = cdf at the end of the k-th run of `per` cells
        s_chunk[tid] = cdf[min((tid + 1) * per, N) - 1];
        __syncthreads();
        // new = searchsorted(cdf, x, side="right"); first occurrence of each value, in draw order
        // Four draws per thread advance together, so the dependent probes of one search (global loads in the second
        // level) overlap with those of three others.
        for (int j0 = tid; j0 < need; j0 += 4 * kSelBlock) {
            double x[4];
            int lo[4], len[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kSelBlock;
                x[u] = (j < need) ? A.draws[j] : 0.0;
                lo[u] = 0; len[u] = (j < need) ? kSelBlock : 0;     // first run whose last cdf value exceeds x
            }
            // (probes are issued unconditionally at a clamped index - a finished search re-reads a valid element and
            //  ignores it - so that the four loads of a step are in flight together)
            while ((len[0] | len[1] | len[2] | len[3]) > 0) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = s_chunk[min(lo[u] + (len[u] >> 1), kSelBlock - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int half = len[u] >> 1;
                    const bool go = len[u] > 0, right = v[u] <= x[u];
                    lo[u] = (go && right) ? lo[u] + half + 1 : lo[u];
                    len[u] = go ? (right ? len[u] - half - 1 : half) : 0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kSelBlock;
                const int b = min(lo[u] * per, N);
                lo[u] = b; len[u] = (j < need) ? (min(b + per, N) - b) : 0;
            }
            while ((len[0] | len[1] | len[2] | len[3]) > 0) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = cdf[min(lo[u] + (len[u] >> 1), N - 1)];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int half = len[u] >> 1;
                    const bool go = len[u] > 0, right = v[u] <= x[u];
                    lo[u] = (go && right) ? lo[u] + half + 1 : lo[u];
                    len[u] = go ? (right ? len[u] - half - 1 : half) : 0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kSelBlock;
                if (j < need) { A.cand[j] = lo[u]; A.first[lo[u]] = 0x7fffffff; }
            }
        }
        __syncthreads();
        LFD_SEL_STAMP();  // search
        for (int j = tid; j < need; j += kSelBlock) atomicMin(&A.first[A.cand[j]], j);
        __syncthreads();
        // thread t owns the draws [t*c, (t+1)*c): one workgroup scan over the per-thread counts keeps the draw order.
        // (A.first was updated by L2 atomics: it is read back with agent-scope loads, past this CU's L1.)
        int appended = 0;
        {
            const int c = (need + kSelBlock - 1) / kSelBlock;
            const int j_lo = min(tid * c, need), j_hi = min(j_lo + c, need);
            unsigned long long keep_bits = 0ull;                       // c <= 64 draws per thread, else re-checked below
            int cnt = 0;
            for (int j = j_lo; j < j_hi; ++j) {
                const int k = __hip_atomic_load(&A.first[A.cand[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == j;
                cnt += k;
                if (j - j_lo < 64) keep_bits |= (unsigned long long)k << (j - j_lo);
            }
            int total;
            int pos = block_excl_scan_i32(cnt, s_i, tid, total);
            for (int j = j_lo; j < j_hi; ++j) {
                const int k = (j - j_lo < 64) ? (int)((keep_bits >> (j - j_lo)) & 1ull)
                                              : (__hip_atomic_load(&A.first[A.cand[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == j);
                if (k) A.found[n_uniq + pos++] = A.cand[j];
            }
            appended = total;
            __syncthreads();
        }
        if (tid == 0) s_n_uniq = n_uniq + appended;
        __syncthreads();
        LFD_SEL_STAMP();  // first-occurrence compaction
    }

    // ---- tile coverage: best cell of every tile bin, bins by descending weight (ties: lower index) ----------
    const int tile = max(1, W / A.tiles);
    const int nbx = (W - 1) / tile + 1, nby = (H - 1) / tile + 1;
    const int nbins = nbx * nby;
    if (nbins > LFD_SELECT_MAX_BINS) { if (tid == 0) *A.status = LFD_SELECT_TOO_MANY_BINS; return; }
    for (int b = tid; b < nbins; b += kSelBlock) s_bin[b] = 0ull;
    __syncthreads();
    if (vec4) {
#pragma unroll 4
        for (int g = tid; g < (N >> 2); g += kSelBlock) {
            const int i = g << 2;
            const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
            const float wvs[4] = {w4.x, w4.y, w4.z, w4.w};
            const int y = i / W, x = i - y * W;
            // the four cells share a row and usually a bin: their maximum is formed in registers and sent as one atomic
            const int by = y / tile;
            int bx = x / tile, rx = x - bx * tile;
            int cur_bin = -1;
            unsigned long long cur_key = 0ull;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (wvs[e] > 0.0f) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(wvs[e]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)(i + e));
                    const int b = bx * nby + by;
                    if (b == cur_bin) { cur_key = key > cur_key ? key : cur_key; }     // positive floats order like their bit patterns
                    else { if (cur_bin >= 0) atomicMax(&s_bin[cur_bin], cur_key); cur_bin = b; cur_key = key; }
                }
                if (++rx >= tile) { rx = 0; ++bx; }
            }
            if (cur_bin >= 0) atomicMax(&s_bin[cur_bin], cur_key);
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < N; i += kSelBlock) {
            const float wv = wbuf[i];
            if (wv > 0.0f) {
                const int y = i / W, x = i - y * W;
                const unsigned long long key = ((unsigned long long)__float_as_uint(wv) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
                atomicMax(&s_bin[(x / tile) * nby + (y / tile)], key);     // positive floats order like their bit patterns
            }
        }
    }
    __syncthreads();
    LFD_SEL_STAMP();      // coverage bins
    const int budget = max(A.M - size, 1);
    // mark array: the random part (most of it flagged already), then the `budget` heaviest bins
    for (int j = tid; j < size; j += kSelBlock) mark[A.found[j]] = 1;
    for (int b = tid; b < nbins; b += kSelBlock) {
        const unsigned long long mine = s_bin[b];
        if (mine == 0ull) continue;
        int rank = 0;
#pragma unroll 8
        for (int o = 0; o < nbins; ++o) rank += (s_bin[o] > mine);  // keys are distinct (they embed the cell index); unrolled: the LDS reads overlap
        if (rank < budget) mark[0xffffffffu - (unsigned)(mine & 0xffffffffull)] = 1;
    }
    __syncthreads();
    LFD_SEL_STAMP();      // marks
    // ---- np.unique(concat): marked cells in ascending order --------------------------------------------------------
    {
        const int lo = min(tid * per, N), hi = min(lo + per, N);
        const bool v16 = (per & 15) == 0 && hi - lo == per;       // whole 16-byte groups (the scratch array is 256-byte aligned)
        int cnt = 0;
        if (v16) {
#pragma unroll 4
            for (int i = lo; i < hi; i += 16) {
                const uint4 m = *reinterpret_cast<const uint4*>(mark + i);
                cnt += __popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w);      // marks are 0 or 1
            }
        } else {
            for (int i = lo; i < hi; ++i) cnt += mark[i];
        }
        int total;
        int pos = block_excl_scan_i32(cnt, s_i, tid, total);
        if ((long long)total > A.capacity) { if (tid == 0) { *A.status = LFD_SELECT_CAPACITY; *A.n_out = total; } return; }
        if (v16) {
            if (cnt) {
                for (int i = lo; i < hi; i += 16) {
                    const uint4 m = *reinterpret_cast<const uint4*>(mark + i);
                    const unsigned ws[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned bits = ws[q];
                        while (bits) {
                            const int b = __ffs((int)bits) - 1;          // bit 8*j set <=> byte j == 1
                            A.sel_out[pos++] = (long long)(i + 4 * q + (b >> 3));
                            bits &= bits - 1u;
                        }
                    }
                }
            }
        } else {
            for (int i = lo; i < hi; ++i) if (mark[i]) A.sel_out[pos++] = (long long)i;
        }
        if (tid == 0) { *A.n_out = total; if (A.sel_offsets_out) A.sel_offsets_out[1] = A.sel_offsets_out[0] + total; }
    }
    LFD_SEL_STAMP();      // unique
#undef LFD_SEL_STAMP
}

// =================================================================================================
// The same selection on several workgroups (one per CU): the streaming passes and the searches are split over n_wg
// workgroups that meet at grid barriers; one extra workgroup produces the MT19937 stream (sequential by nature) ahead of
// what is asked for and commits it - at the position the launch's last reference stopped at - only when that reference
// says so: a reference whose input upstream would refuse (its argument checks) draws nothing.  Results are bit-identical
// to the single-workgroup kernel: every sum involved is exact (see the header), so neither the partition nor the order of
// the atomic updates changes it; the cells found are a SET (upstream returns np.unique of them).
// =================================================================================================
namespace {

// all workgroups of the grid; bounded spin (a workgroup that never arrives is reported, not waited for forever)
__device__ bool grid_barrier(unsigned* bar, unsigned n_wg_total) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        __threadfence();
        const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicAdd(bar, 1u) == n_wg_total - 1u) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            atomicAdd(bar + 1, 1u);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                if (++spins > (1u << 22)) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __threadfence();
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(LFD_SELECT_BLOCK) lfd_select_filter_mw_kernel(LfdSelectArgs A_launch, LfdSelectNorms norms) {
    const LfdSelectArgs A = lfd_select_args_of(A_launch);
    const float s_handed_in = norms.use ? norms.s[blockIdx.y] : A.s_override;
    __shared__ double s_d[kSelBlock / 64];
    __shared__ int s_i[kSelBlock / 64];
    __shared__ unsigned long long s_bin[LFD_SELECT_MAX_BINS];
    __shared__ double s_tab[16 * LFD_SELECT_MAX_WG];       // span sums, fetched once per round
    __shared__ double s_pref[16 * LFD_SELECT_MAX_WG];      // ... their inclusive prefix (scan rounds)
    __shared__ unsigned long long s_msg;

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = (int)blockIdx.x, G = A.n_wg;                               // grid = G compute workgroups + 1
    const bool rng_wg = wg == G;
    const int H = A.H, W = A.W, N = H * W;
    const float* cert = A.best_cert;
    float* wbuf = A.weights;
    double* cdf = A.cdf;
    unsigned char* mark = A.mark;          // bit 0: drawn (p = 0 from now on), bit 1: coverage pick
    unsigned* bar = reinterpret_cast<unsigned*>(A.coop + LFD_COOP_BAR);
    int* flags = reinterpret_cast<int*>(A.coop + LFD_COOP_FLAGS);
    double* g_part = reinterpret_cast<double*>(A.coop + LFD_COOP_PART);
    double* g_span = reinterpret_cast<double*>(A.coop + LFD_COOP_SPAN);
    int* g_cnt = reinterpret_cast<int*>(A.coop + LFD_COOP_WGCNT);
    unsigned long long* g_bins = reinterpret_cast<unsigned long long*>(A.coop + LFD_COOP_BINS);
    const int size = min((int)((double)A.M * 0.85), N);           // int(M * 0.85), f64 product like Python

    // ================= the workgroup on the MT19937 stream =================
    // The stream of a launch is an ARRAY (mt_stream_producer): the extra workgroup of the launch's first reference writes its doubles into a ring,
    // ahead of what is asked for, and never meets the others at a barrier; a reference draws from where the one before it stopped (the chain block),
    // and where the last one stops is where the producer commits the stream.  A launch of one reference is a chain of one.
    if (rng_wg) {
        if (A.chain_refs == 1 || blockIdx.y == 0) mt_stream_producer(A, A.chain_refs, size, tid);
        return;
    }

    // ================= the compute workgroups =================
#define LFD_GRID_SYNC() do { if (!grid_barrier(bar, (unsigned)G)) { if (tid == 0) { *A.status = LFD_SELECT_NO_PROGRESS; chain_hand_on(LFD_CHAIN_BROKEN); } return; } } while (0)
    // where this reference's draws begin in the stream (absolute index of the doubles), learnt from the reference before it
    // (LFD_CHAIN_OFF: 1 + index; the launch's first reference starts at 0), and handed on - by EVERY way out of this kernel, refusals included: a
    // refused reference consumes nothing (upstream raises before it draws), a failed one breaks the chain for those behind it
    const int yref = A.chain_refs == 1 ? 0 : (int)blockIdx.y;            // this reference's place in its chain
    unsigned long long* ch_off = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_OFF);
    unsigned long long* ch_want = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_WANT);
    unsigned long long* ch_released = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_RELEASED);
    unsigned long long* ch_produced = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_PRODUCED);
    const unsigned long long ring_mask = (unsigned long long)A.ring_cap - 1ull;
    auto chain_begin = [&]() -> unsigned long long {            // every thread of the workgroup; LFD_CHAIN_BROKEN: the predecessor failed or never spoke
        if (yref == 0) return 0ull;
        if (tid == 0) {
            unsigned long long v = 0ull;
            unsigned spins = 0;
            while ((v = __hip_atomic_load(ch_off + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
                if (++spins > (1u << 23)) { v = LFD_CHAIN_BROKEN; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            s_msg = v;
        }
        __syncthreads();
        const unsigned long long v = s_msg;
        __syncthreads();
        return v == LFD_CHAIN_BROKEN ? v : v - 1ull;
    };
    auto chain_hand_on = [&](unsigned long long next_begin) {    // one thread
        __threadfence();                                           // (behind a tentative number, if one was said: see where the successor reads them)
        __hip_atomic_store(ch_off + yref + 1, next_begin == LFD_CHAIN_BROKEN ? next_begin : next_begin + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // a reference that leaves before it has drawn: the stream passes through it untouched
#define LFD_CHAIN_PASS() do { if (wg == 0) { const unsigned long long b_ = chain_begin(); if (tid == 0) chain_hand_on(b_); } } while (0)
#define LFD_CHAIN_BREAK() do { if (tid == 0) chain_hand_on(LFD_CHAIN_BROKEN); } while (0)
    if (wg == 0 && tid == 0) { *A.n_out = 0; *A.status = LFD_SELECT_OK; }
    int t_slot = 0;
#define LFD_MW_STAMP() do { if (A.timing && wg == 0 && tid == 0 && t_slot < 30) A.timing[t_slot] = wall_clock64(); ++t_slot; } while (0)
    LFD_MW_STAMP();
    // wave spans: span index s = wg * 16 + wave covers [s * span, (s + 1) * span), a multiple of 256 cells
    constexpr int nwaves = kSelBlock / 64;
    const int n_spans = G * nwaves;
    const int span = ((N + n_spans - 1) / n_spans + 255) & ~255;
    const int sidx = wg * nwaves + wave;
    const int w_lo = min(sidx * span, N), w_hi = min(w_lo + span, N);
    const bool cert16 = (reinterpret_cast<uintptr_t>(cert) & 15u) == 0;
    const int tile = max(1, W / A.tiles);
    const int nbx = (W - 1) / tile + 1, nby = (H - 1) / tile + 1;
    const int nbins = nbx * nby;                                  // <= LFD_SELECT_MAX_BINS (checked by the host)
    const int T = G * kSelBlock;                                  // compute threads
    const int gt = wg * kSelBlock + tid;

    // ---- weights, exact sum, NaN check ------------------------------------------------------------------------------
    // A normaliser that was handed in (upstream's own torch sum: the pipeline's default) needs no sum of the weights: this pass and its barrier are
    // skipped, the next one takes the certainties themselves (and looks for NaN)
    const bool direct = s_handed_in > 0.0f;
    for (int b = tid; b < nbins; b += kSelBlock) s_bin[b] = 0ull;
    if (direct) __syncthreads();
    else {
        double acc = 0.0;
        int bad = 0;
#pragma unroll 4
        for (int i = w_lo + 4 * lane; i < w_hi; i += 256) {
            float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const bool full = i + 3 < w_hi;
            if (full && cert16) { const float4 c4 = *reinterpret_cast<const float4*>(cert + i); cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w; }
            else { for (int e = 0; e < 4 && i + e < w_hi; ++e) cs[e] = cert[i + e]; }
            float ws[4];
            // (row and column of the first of the four cells by ONE division, the others by stepping: an integer division per cell - this pass
            // has nothing else to do - was most of its time)
            int y = i / W, x = i - y * W;
#pragma unroll
            for (int e = 0; e < 4; ++e, ++x) {
                const int ii = i + e;
                if (x == W) { x = 0; ++y; }
                float c = cs[e];
                c = (c > A.cap) ? A.cap : c;                       // torch.clamp(max=cap); NaN stays
                const bool inside = x >= A.border && x <= W - 1 - A.border && y >= A.border && y <= H - 1 - A.border;
                const float w = (ii < w_hi) ? c * (inside ? 1.0f : 0.0f) : 0.0f;
                ws[e] = w;
                if (w != w) bad = 1;
                acc += (double)w;
            }
            if (full) *reinterpret_cast<float4*>(wbuf + i) = make_float4(ws[0], ws[1], ws[2], ws[3]);
            else { for (int e = 0; e < 4 && i + e < w_hi; ++e) wbuf[i + e] = ws[e]; }
        }
        const double wsum = block_sum_f64(acc, s_d, tid);
        const int any_bad = block_sum_i32(bad, s_i, tid);
        if (tid == 0) { g_part[wg] = wsum; if (any_bad) atomicOr(&flags[3], 1); }
        LFD_MW_STAMP();
        LFD_GRID_SYNC();
        LFD_MW_STAMP();
    }
    // (the coverage bins in memory were zeroed with the barrier words, before the launch)
    double s64 = 0.0;
    if (!direct) for (int g = 0; g < G; ++g) s64 += g_part[g];     // same order in every workgroup
    const float s32 = direct ? s_handed_in : (float)s64;
    if (!direct && __hip_atomic_load(&flags[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (wg == 0 && tid == 0) *A.status = LFD_SELECT_NAN;
        LFD_CHAIN_PASS();
        return;
    }
    if (!(s32 > 0.0f)) {                                           // upstream: `if s <= 0: return empty` (the stream is not touched)
        LFD_CHAIN_PASS();
        return;
    }

    // ---- p = (weights / s) as f32; exactness precondition; non-zero count; span sums; marks initialised; coverage bins (best cell of every tile bin, by weight, ties: lower index) -----------------------------
    {
        int nz = 0, inexact = 0, neg = 0, nan_seen = 0;
        double part = 0.0;
#pragma unroll 2
        for (int i = w_lo + 4 * lane; i < w_hi; i += 256) {
            float pf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (direct) {
                // weight = clamp(certainty, max = cap) * inside, as the skipped pass computes it
                float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const bool full = i + 3 < w_hi;
                if (full && cert16) { const float4 c4 = *reinterpret_cast<const float4*>(cert + i); cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w; }
                else { for (int e = 0; e < 4 && i + e < w_hi; ++e) cs[e] = cert[i + e]; }
                int yy = i / W, xx = i - yy * W;
#pragma unroll
                for (int e = 0; e < 4; ++e, ++xx) {
                    if (xx == W) { xx = 0; ++yy; }
                    float c = cs[e];
                    c = (c > A.cap) ? A.cap : c;
                    const bool inside = xx >= A.border && xx <= W - 1 - A.border && yy >= A.border && yy <= H - 1 - A.border;
                    const float w = (i + e < w_hi) ? c * (inside ? 1.0f : 0.0f) : 0.0f;
                    if (w != w) nan_seen = 1;
                    pf[e] = w / s32;
                }
                if (full) { *reinterpret_cast<float4*>(wbuf + i) = make_float4(pf[0], pf[1], pf[2], pf[3]); *reinterpret_cast<unsigned*>(mark + i) = 0u; }
                else { for (int e = 0; e < 4 && i + e < w_hi; ++e) { wbuf[i + e] = pf[e]; mark[i + e] = 0; } }
            } else if (i + 3 < w_hi) {
                const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
                pf[0] = w4.x / s32; pf[1] = w4.y / s32; pf[2] = w4.z / s32; pf[3] = w4.w / s32;
                *reinterpret_cast<float4*>(wbuf + i) = make_float4(pf[0], pf[1], pf[2], pf[3]);
                *reinterpret_cast<unsigned*>(mark + i) = 0u;
            } else {
                for (int e = 0; e < 4 && i + e < w_hi; ++e) { pf[e] = wbuf[i + e] / s32; wbuf[i + e] = pf[e]; mark[i + e] = 0; }
            }
            int cur_bin = -1;
            unsigned long long cur_key = 0ull;
            // row, column and coverage tile of the first of the four cells by three divisions, the others by stepping (six divisions per cell were
            // most of this pass)
            int y = i / W, x = i - y * W, bx = x / tile, rx = x - bx * tile, by = y / tile;
#pragma unroll
            for (int e = 0; e < 4; ++e, ++x, ++rx) {
                if (x == W) { x = 0; ++y; bx = 0; rx = 0; by = y / tile; }
                if (rx == tile) { rx = 0; ++bx; }
                if (pf[e] > 0.0f) {
                    ++nz;
                    if (pf[e] < 1.862645149230957e-09f) inexact = 1;   // 2^-29
                    const int ii = i + e;
                    const unsigned long long key = ((unsigned long long)__float_as_uint(pf[e]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)ii);
                    const int b = bx * nby + by;
                    if (b == cur_bin) { cur_key = key > cur_key ? key : cur_key; }     // positive floats order like their bit patterns
                    else { if (cur_bin >= 0) atomicMax(&s_bin[cur_bin], cur_key); cur_bin = b; cur_key = key; }
                }
                if (pf[e] < 0.0f) neg = 1;
            }
            if (cur_bin >= 0) atomicMax(&s_bin[cur_bin], cur_key);       // (handing a run of lanes' keys to one lane first was measured: no gain)
            part += ((double)pf[0] + (double)pf[1]) + ((double)pf[2] + (double)pf[3]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) g_span[sidx] = part;
        nz = block_sum_i32(nz, s_i, tid);
        inexact = block_sum_i32(inexact, s_i, tid);
        neg = block_sum_i32(neg, s_i, tid);                       // (its barriers also complete the LDS bins)
        if (direct) nan_seen = block_sum_i32(nan_seen, s_i, tid);
        if (tid == 0) { atomicAdd(&flags[0], nz); if (inexact) atomicOr(&flags[1], 1); if (neg) atomicOr(&flags[2], 1); if (nan_seen) atomicOr(&flags[3], 1); }
        for (int b = tid; b < nbins; b += kSelBlock) if (s_bin[b]) atomicMax(&g_bins[b], s_bin[b]);
    }
    LFD_MW_STAMP();
    LFD_GRID_SYNC();
    LFD_MW_STAMP();
    {
        const int nz = __hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int inexact = __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int neg = __hip_atomic_load(&flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int st = LFD_SELECT_OK;
        if (direct && __hip_atomic_load(&flags[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) st = LFD_SELECT_NAN;
        else if (neg) st = LFD_SELECT_NEGATIVE; else if (nz < size) st = LFD_SELECT_FEWER_NONZERO; else if (inexact) st = LFD_SELECT_INEXACT;
        if (st != LFD_SELECT_OK) {
            if (wg == 0 && tid == 0) *A.status = st;
            LFD_CHAIN_PASS();
            return;
        }
    }

    // ---- legacy choice(replace=False, p) -----------------------------------------------------------------------
    // Guide table of the searches: guide[k] = searchsorted(cdf, k / K, side="right") for K a power of two of about a quarter of the cells.  The
    // cumulative-sum pass writes it as it goes (the cell whose interval a k / K falls into knows so), a draw x then starts from
    // [guide[k], guide[k + 1]], k = floor(x K): two dependent reads and - with weights within an order of magnitude of each other - about four
    // cells, instead of a bisection's eight dependent reads of a cumulative sum that other XCDs wrote a moment ago.  (The searches of a round are
    // bound by the cache misses a CU can have in flight: eight per draw were 13 of the phase's 19 us.)
    int guide_K = 1024;
    while (guide_K * 2 <= N / 4) guide_K *= 2;
    const double guide_Kd = (double)guide_K;
    int* guide = A.first;                                         // [N] scratch, K + 1 <= N entries used
    // searchsorted(cdf, x, side="right") = how many entries are <= x (the cumulative sum never decreases): from the guide table's bracket
    auto guided_search = [&](double x) -> int {
        const int k = (int)(x * guide_Kd);                        // x < 1: k <= K - 1
        int lo = guide[k];
        int hi = (k + 1 < guide_K) ? guide[k + 1] : N;
        while (hi - lo > 8) { const int half = (hi - lo) >> 1, mid = lo + half; if (cdf[mid] <= x) lo = mid + 1; else hi = mid; }
        int below = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (lo + e < hi) below += (cdf[lo + e] <= x) ? 1 : 0;
        return lo + below;
    };
    // A reference that is not the first of its chain looks its draws up BEFORE it knows which they are: they start somewhere behind its predecessor's
    // first round - at that round's end if the predecessor needs no more, a few (the duplicates of that round) further otherwise - so it looks up a
    // window of M doubles from there while the predecessor is still drawing; a lookup changes nothing.  What is left for the chain is to MARK the
    // cells of the `size` draws the reference turns out to own.
#ifdef LFD_CHAIN_STAMPS
    unsigned long long st_[8] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0, 0, 0};
#define LFD_CS(k) do { st_[k] = (unsigned long long)wall_clock64(); } while (0)
#else
#define LFD_CS(k) do { } while (0)
#endif
    unsigned long long pre_base = 0ull;          // absolute index of the first double looked up in advance
    int pre_win = 0, pre_cell = -1;              // doubles looked up in advance (thread gt: the double pre_base + gt), this thread's cell
    unsigned long long* ch_beg = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_BEG);
    unsigned long long* ch_tent = reinterpret_cast<unsigned long long*>(A.chain + LFD_CHAIN_TENT);
    int n_uniq = 0, guard = 0;
    unsigned long long my_begin = 0ull, consumed = 0ull;           // this reference's first double in the stream, doubles it has used
    while (n_uniq < size) {
        if (++guard > 60) {
            if (wg == 0 && tid == 0) *A.status = LFD_SELECT_NO_PROGRESS;
            if (wg == 0) LFD_CHAIN_BREAK();
            return;
        }
        const int need = size - n_uniq;
        // A LATER round with few draws left (the duplicates of the round before: ~140 of 8 500) does not rebuild the cumulative sum - a streaming
        // pass and a grid barrier for a hundred searches: a wave finds every draw's cell from the live span sums and ONE span's weights (below).
        const bool scan_round = guard > 1 && need <= kScanRoundDraws;
        // the live span sums (the p pass wrote them, the draws of the rounds so far took their cells out), their total, the division by it
        for (int i = tid; i < n_spans; i += kSelBlock) s_tab[i] = g_span[i];
        __syncthreads();
        double carry = 0.0, total = 0.0;
        for (int i = lane; i < n_spans; i += 64) { const double v = s_tab[i]; total += v; if (i < sidx) carry += v; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { carry += __shfl_xor(carry, off, 64); total += __shfl_xor(total, off, 64); }
        const double rtot = 1.0 / total;
        const bool quick_div = (__double_as_longlong(total) & 0xfffffffffffffll) != 0xfffffffffffffll;
        auto div_total = [&](double a) {       // correctly rounded a / total (see the single-workgroup kernel)
            if (quick_div) { const double q = a * rtot; return fma(fma(-total, q, a), rtot, q); }
            return a / total;
        };
        if (guard == 1) {
            // the coverage picks (the `budget` heaviest bins) are flagged here - before the first cumulative sum, which does not look at them: with several
            // references on one stream this is the part of a selection that runs side by side with the others, not the part they wait for - - bit 1 of the mark, which the cumulative sum ignores
            unsigned* mark32 = reinterpret_cast<unsigned*>(mark);
            const int budget = max(A.M - size, 1);
            for (int b = tid; b < nbins; b += kSelBlock) s_bin[b] = g_bins[b];
            __syncthreads();
            // rank of a bin = bins with a larger key (keys are distinct: they embed the cell index); 32 threads per bin share the
            // comparisons (one thread per bin walked the whole table: ~30 us of dependent LDS reads)
            const int bpw = (nbins + G - 1) / G, b0 = wg * bpw;
            for (int bl0 = 0; bl0 < bpw; bl0 += kSelBlock / 32) {
                const int bl = bl0 + (tid >> 5), b = b0 + bl, sub = tid & 31;
                const bool valid = bl < bpw && b < nbins;
                const unsigned long long mine = valid ? s_bin[b] : 0ull;
                int rank = 0;
                if (mine != 0ull) for (int o = sub; o < nbins; o += 32) rank += (s_bin[o] > mine);
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) rank += __shfl_xor(rank, off, 32);
                if (mine != 0ull && sub == 0 && rank < budget) {
                    const unsigned cell = 0xffffffffu - (unsigned)(mine & 0xffffffffull);
                    atomicOr(mark32 + (cell >> 2), 2u << (8 * (cell & 3u)));
                }
            }
        }
        // cdf = cumsum(p) / cdf[-1] over this wave's span; the carry comes from the table of span sums
        if (!scan_round) {
            for (int base = w_lo; base < w_hi; base += 256) {
                const int i = base + 4 * lane;
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                const bool full = i + 3 < w_hi;
                if (full) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
                    const unsigned m4 = *reinterpret_cast<const unsigned*>(mark + i);
                    v0 = (m4 & 0x1u) ? 0.0 : (double)w4.x; v1 = (m4 & 0x100u) ? 0.0 : (double)w4.y;
                    v2 = (m4 & 0x10000u) ? 0.0 : (double)w4.z; v3 = (m4 & 0x1000000u) ? 0.0 : (double)w4.w;
                } else {
                    if (i < w_hi) v0 = (mark[i] & 1) ? 0.0 : (double)wbuf[i];
                    if (i + 1 < w_hi) v1 = (mark[i + 1] & 1) ? 0.0 : (double)wbuf[i + 1];
                    if (i + 2 < w_hi) v2 = (mark[i + 2] & 1) ? 0.0 : (double)wbuf[i + 2];
                }
                v1 += v0; v2 += v1; v3 += v2;
                double v = v3;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double n = __shfl_up(v, off, 64);
                    if (lane >= off) v += n;
                }
                const double before = carry + (v - v3);
                const double c0 = div_total(before + v0), c1 = div_total(before + v1), c2 = div_total(before + v2), c3 = div_total(before + v3);
                if (full) {
                    *reinterpret_cast<double2*>(cdf + i) = make_double2(c0, c1);
                    *reinterpret_cast<double2*>(cdf + i + 2) = make_double2(c2, c3);
                } else {
                    if (i < w_hi) cdf[i] = c0;
                    if (i + 1 < w_hi) cdf[i + 1] = c1;
                    if (i + 2 < w_hi) cdf[i + 2] = c2;
                }
                // guide table: cell j is the first one whose cumulative sum exceeds k / K for every k in [ceil(cdf[j - 1] K), ceil(cdf[j] K)) (all of it
                // exact: K is a power of two); usually none or one k per cell.  A cell that owns many (a weight far above the others) shares them
                // out over the wave.
                {
                    int kl = (int)ceil(div_total(before) * guide_Kd);
                    const double cs[4] = {c0, c1, c2, c3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool valid = i + e < w_hi;
                        const int kh = valid ? min((int)ceil(cs[e] * guide_Kd), guide_K) : kl;
                        if (kl < kh) guide[kl] = i + e;
                        if (kl + 1 < kh) guide[kl + 1] = i + e;
                        unsigned long long many = __ballot(kh - kl > 2);
                        while (many) {
                            const int src = __ffsll((long long)many) - 1;
                            many &= many - 1ull;
                            const int from = __shfl(kl, src, 64) + 2, to = __shfl(kh, src, 64), cell = __shfl(i, src, 64) + e;
                            for (int k = from + lane; k < to; k += 64) guide[k] = cell;
                        }
                        kl = kh;
                    }
                }
                carry += __shfl(v, 63, 64);
            }
            LFD_MW_STAMP();
            LFD_GRID_SYNC();
            LFD_MW_STAMP();
        } else {
            // inclusive prefix of the live span sums (exact, like every sum here): wave 0, 64 spans at a time
            if (wave == 0) {
                double run = 0.0;
                for (int base = 0; base < n_spans; base += 64) {
                    const int q = base + lane;
                    double v = q < n_spans ? s_tab[q] : 0.0;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const double n = __shfl_up(v, off, 64); if (lane >= off) v += n; }
                    if (q < n_spans) s_pref[q] = run + v;
                    run += __shfl(v, 63, 64);
                }
            }
            __syncthreads();
        }
    LFD_MW_STAMP();
        if (guard == 1 && yref > 0) {
            // the predecessor's first double, or this reference's own (the predecessor refused its input or is through already): whichever is known first
            if (tid == 0) {
                unsigned long long v = 0ull;
                unsigned spins = 0;
                while (true) {
                    if (__hip_atomic_load(ch_off + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { v = 0ull; break; }
                    v = __hip_atomic_load(ch_beg + yref - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v != 0ull) break;
                    if (++spins > (1u << 23)) { v = 0ull; break; }          // (the wait for this reference's own first double below reports it)
                    __builtin_amdgcn_s_sleep(4);
                }
                s_msg = v;
            }
            __syncthreads();
            const unsigned long long pb = s_msg;
            __syncthreads();
            LFD_CS(1);
            if (pb != 0ull) {
                pre_base = (pb - 1ull) + (unsigned long long)size;
                pre_win = min(max(A.M, size), T);                            // (one double per thread)
                // the producer is on its way there: it stays M doubles ahead of the predecessor's first round while another reference follows
                if (tid == 0) {
                    unsigned spins = 0;
                    int ok = 1;
                    // (this workgroup's share of the window only: the producer is busy with little else than this window while the predecessor
                    // draws, the lookups follow it piece by piece)
                    const int my_end = min(pre_win, (wg + 1) * kSelBlock);
                    while (__hip_atomic_load(ch_produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pre_base + (unsigned long long)my_end) {
                        if (__hip_atomic_load(ch_off + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == LFD_CHAIN_BROKEN || ++spins > (1u << 22)) { ok = 0; break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    __threadfence();
                    s_i[0] = ok;
                }
                __syncthreads();
                const int ok = s_i[0];
                __syncthreads();
                if (!ok) pre_win = 0;                                       // (no lookup in advance: the ordinary path below, which also reports a broken chain)
                else if (gt < pre_win) pre_cell = guided_search(A.ring[(pre_base + (unsigned long long)gt) & ring_mask]);
            }
            LFD_CS(2);
        }
        // A reference that is not the first of its chain marks the cells of its first round on a TENTATIVE first double: its predecessor says, as
        // soon as ITS first round is counted, where the successor starts if the predecessor's next round is its last (it nearly always is: that round
        // looks for the first round's ~140 duplicates and would have to draw a duplicate itself) - the successor's marking and counting then run
        // beside the predecessor's second round.  When the predecessor is through, the number is confirmed - or the successor takes its marks back
        // (exactly: every thread the cells it itself turned, the weights back into their spans), meets at a barrier and marks again.
        bool speculative = false;
        unsigned long long newbits = 0ull;           // which of this thread's draws of the round turned a cell (bit 0: the one looked up in advance)
        int appended = 0;
        auto publish_begin = [&]() {
            if (wg == 0 && tid == 0) {
                __hip_atomic_store(reinterpret_cast<unsigned*>(A.chain + LFD_CHAIN_CURRENT), (unsigned)yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ch_released, my_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ch_want, my_begin + (unsigned long long)need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ch_beg + yref, my_begin + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (after `want`: the successor waits for the producer)
            }
        };
        for (int attempt = 0; ; ++attempt) {
        // the draws of this round must be in place
        {
            if (guard == 1 && attempt == 0) {
                // only now does anything here depend on the references before this one: where they left the stream
                if (yref > 0 && (need + T - 1) / T <= 60) {
                    if (tid == 0) {
                        unsigned long long v = 0ull;
                        int fin = 1;
                        unsigned spins = 0;
                        // EVERY workgroup of the reference has to take the same way: tentative if the predecessor has said a tentative number at
                        // all - even when the final one is known by now - and final only if it never will (it was through after one round, or
                        // refused).  The predecessor writes the tentative number first and the final one behind a fence; read in the opposite
                        // order, a final number without a tentative one means there is none.
                        while (true) {
                            const unsigned long long f = __hip_atomic_load(ch_off + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            unsigned long long t = __hip_atomic_load(ch_tent + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (f == LFD_CHAIN_BROKEN) { v = f; break; }
                            if (t == 0ull && f != 0ull) {           // (the fence only here, not in every turn of the wait)
                                __threadfence();
                                t = __hip_atomic_load(ch_tent + yref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            if (t != 0ull) { v = t; fin = 0; break; }
                            if (f != 0ull) { v = f; break; }
                            if (++spins > (1u << 23)) { v = LFD_CHAIN_BROKEN; break; }
                            __builtin_amdgcn_s_sleep(4);
                        }
                        s_msg = v;
                        s_i[1] = fin;
                    }
                    __syncthreads();
                    const unsigned long long v = s_msg;
                    speculative = s_i[1] == 0;
                    __syncthreads();
                    my_begin = v == LFD_CHAIN_BROKEN ? v : v - 1ull;
                } else my_begin = chain_begin();
                LFD_CS(3);
                if (my_begin == LFD_CHAIN_BROKEN) { if (wg == 0 && tid == 0) { *A.status = LFD_SELECT_NO_PROGRESS; chain_hand_on(LFD_CHAIN_BROKEN); } return; }
                if (!speculative) publish_begin();
            }
            if (tid == 0) {
                const unsigned long long end = my_begin + consumed + (unsigned long long)need;
                unsigned spins = 0;
                int ok = 1;
                while (__hip_atomic_load(ch_produced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < end) {
                    if (++spins > (1u << 22)) { ok = 0; break; }
                    __builtin_amdgcn_s_sleep(4);
                }
                __threadfence();
                s_i[0] = ok;
            }
            __syncthreads();
            const int ok = s_i[0];
            __syncthreads();
            if (!ok) { if (tid == 0) { *A.status = LFD_SELECT_NO_PROGRESS; chain_hand_on(LFD_CHAIN_BROKEN); } return; }
        }
        // new = searchsorted(cdf, x, side="right"), and what numpy does with it - keep the first occurrence of every distinct value, append, p[found] = 0 -
        // in the SAME phase: the selection's result is np.unique(...) of the cells found, so the ORDER they are appended in is never observed; a cell is
        // new exactly when its "drawn" bit was clear (a cell found in an earlier round has p = 0 and a zero-width interval of the cumulative sum: the
        // search cannot return it again), which one atomic OR on the mark's word tells - no first-occurrence words, no ordered compaction, and two grid
        // barriers per round instead of four.  (The marks live in bytes; the coverage picks below set bit 1 with the same atomic, so that neither
        // update can lose the other's bit.)
        unsigned* mark32 = reinterpret_cast<unsigned*>(mark);
        int cnt = 0;
        const unsigned long long round_begin = my_begin + consumed;
        // a cell found in round r gets bit 0 AND r in bits 2-7 of its mark (r <= 60): a scan round has no snapshot of the cumulative sum to search
        // in - it reads the weights themselves, while other waves mark this round's finds - and must count exactly the cells of EARLIER rounds as gone
        const unsigned found_byte = 1u | ((unsigned)guard << 2);
        auto found_now = [&](int cell) {             // one lane per found cell; true: nobody had drawn it before
            const unsigned sh = 8u * ((unsigned)cell & 3u);
            const unsigned old = atomicOr(mark32 + (cell >> 2), found_byte << sh);
            if (old & (1u << sh)) return false;
            atomicAdd(&g_span[cell / span], -(double)wbuf[cell]);                                // p[found] = 0 from now on: exact, hence order-independent
            return true;
        };
        if (scan_round) {
            // searchsorted(cumsum(p) / total, x, "right") = #{i : fl(S_i / total) <= x} = #{i : S_i <= s*}, s* the largest multiple of 2^-52 whose
            // correctly rounded quotient by the total is <= x (the quotient is monotone, every S_i is a multiple of 2^-52): no cell is divided, the
            // draw's threshold is.  s* = floor(x total) up to the rounding of that product: settled with the division itself.
            const double U = 2.220446049250313e-16, t_units = total * 4503599627370496.0;       // 2^-52, total in units (an integer below 2^53)
            const int wave_global = wg * nwaves + wave, n_waves = G * nwaves;
            for (int j = wave_global; j < need; j += n_waves) {
                const double x = A.ring[(round_begin + (unsigned long long)j) & ring_mask];
                double m = floor(x * t_units);
                while (div_total((m + 1.0) * U) <= x) m += 1.0;
                while (div_total(m * U) > x) m -= 1.0;                 // (fl(0 / total) = 0 <= x: m >= 0)
                const double sstar = m * U;
                int sp = 0;
                for (int q = lane; q < n_spans; q += 64) sp += (s_pref[q] <= sstar) ? 1 : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sp += __shfl_xor(sp, off, 64);
                // (sp < n_spans: the last prefix is the total, whose quotient 1.0 is above every draw)
                const double rem = sstar - (sp ? s_pref[sp - 1] : 0.0);
                const int c_lo = min(sp * span, N), c_hi = min(c_lo + span, N);
                int below = 0;
                double run = 0.0;
                // (256 cells at a time, stopping at the piece the threshold falls into: reading the whole span ahead was measured and is slower)
                for (int base = c_lo; base < c_hi; base += 256) {
                    const int i = base + 4 * lane;
                    double v[4] = {0.0, 0.0, 0.0, 0.0};
                    if (i + 3 < c_hi) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wbuf + i);
                        const unsigned m4 = *reinterpret_cast<const unsigned*>(mark + i);
                        const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const unsigned b = (m4 >> (8 * e)) & 0xffu; v[e] = ((b & 1u) && (b >> 2) < (unsigned)guard) ? 0.0 : (double)ws[e]; }
                    } else {
                        for (int e = 0; e < 4 && i + e < c_hi; ++e) { const unsigned b = mark[i + e]; v[e] = ((b & 1u) && (b >> 2) < (unsigned)guard) ? 0.0 : (double)wbuf[i + e]; }
                    }
                    v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
                    double incl = v[3];
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const double n = __shfl_up(incl, off, 64); if (lane >= off) incl += n; }
                    const double before = run + (incl - v[3]);
                    int c = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) c += (i + e < c_hi && before + v[e] <= rem) ? 1 : 0;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
                    below += c;
                    run += __shfl(incl, 63, 64);
                    if (run > rem) break;
                }
                if (lane == 0 && found_now(c_lo + below)) ++cnt;
            }
        } else if (guard == 1 && pre_win > 0) {
            // the cells were looked up in advance: mark those of the draws this reference owns, [round_begin, round_begin + need); what the window
            // did not reach (the predecessor's later rounds took more than M - size doubles: weights piled on a few cells) is looked up now
            const unsigned long long mine = pre_base + (unsigned long long)gt;
            if (gt < pre_win && mine >= round_begin && mine < round_begin + (unsigned long long)need && found_now(pre_cell)) { ++cnt; newbits |= 1ull; }
            const unsigned long long reached = pre_base + (unsigned long long)pre_win, end = round_begin + (unsigned long long)need;
            if (end > reached) {
                const unsigned long long from = round_begin > reached ? round_begin : reached;
                int it = 1;
                for (unsigned long long a = from + (unsigned long long)gt; a < end; a += (unsigned long long)T, ++it)
                    if (found_now(guided_search(A.ring[a & ring_mask]))) { ++cnt; newbits |= 1ull << it; }
            }
        } else {
            int it = 1;
            for (int j = gt; j < need; j += T, ++it)
                if (found_now(guided_search(A.ring[(round_begin + (unsigned long long)j) & ring_mask]))) { ++cnt; if (it < 64) newbits |= 1ull << it; }
        }
        // per-workgroup counts of this round, in a slot of their own per round parity (the `found` scratch, which nothing else uses any more): a
        // workgroup that is through the barrier below and already counting for the NEXT barrier - the next round's, or the final unique pass's, which
        // uses g_cnt - must not overwrite what a slower one is still summing
        int* round_cnt = A.found + (guard & 1) * G;
        {
            const int total = block_sum_i32(cnt, s_i, tid);
            if (tid == 0) round_cnt[wg] = total;
        }
        LFD_MW_STAMP();
        LFD_GRID_SYNC();
        LFD_MW_STAMP();
        appended = 0;
        for (int g = 0; g < G; ++g) appended += round_cnt[g];
        if (!speculative) break;
        {
            // marked and counted on a tentative first double: confirmed ...
            const unsigned long long final_begin = chain_begin();
            if (final_begin == LFD_CHAIN_BROKEN) { if (wg == 0 && tid == 0) { *A.status = LFD_SELECT_NO_PROGRESS; chain_hand_on(LFD_CHAIN_BROKEN); } return; }
            speculative = false;
#ifdef LFD_CHAIN_STAMPS
            st_[6] = my_begin; st_[7] = final_begin;
#endif
            if (final_begin == my_begin) { publish_begin(); break; }
            // ... or taken back: every thread the cells its own draws turned (found again by the same walk), their weights back into the span sums
            auto take_back = [&](int cell) {
                const unsigned sh = 8u * ((unsigned)cell & 3u);
                atomicAnd(mark32 + (cell >> 2), ~(0xfdu << sh));                      // (bit 1, a coverage pick, stays)
                atomicAdd(&g_span[cell / span], (double)wbuf[cell]);
            };
            if (pre_win > 0) {
                if (newbits & 1ull) take_back(pre_cell);
                const unsigned long long reached = pre_base + (unsigned long long)pre_win, end = round_begin + (unsigned long long)need;
                if (end > reached) {
                    const unsigned long long from = round_begin > reached ? round_begin : reached;
                    int it = 1;
                    for (unsigned long long a = from + (unsigned long long)gt; a < end; a += (unsigned long long)T, ++it)
                        if ((newbits >> it) & 1ull) take_back(guided_search(A.ring[a & ring_mask]));
                }
            } else {
                int it = 1;
                for (int j = gt; j < need; j += T, ++it)
                    if ((newbits >> it) & 1ull) take_back(guided_search(A.ring[(round_begin + (unsigned long long)j) & ring_mask]));
            }
            newbits = 0ull;
            my_begin = final_begin;
            publish_begin();
            LFD_GRID_SYNC();
        }
        }      // (attempt)
        {
            if (guard == 1) LFD_CS(4);
            consumed += (unsigned long long)need;
            if (n_uniq + appended < size) {
                if (wg == 0 && tid == 0) {
                    // a successor may start marking now: this is where it starts if the round to come is this reference's last
                    if (guard == 1) __hip_atomic_store(ch_tent + yref + 1, my_begin + consumed + (unsigned long long)(size - n_uniq - appended) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (everybody is through with this round's draws: they are behind the barrier above)
                    __hip_atomic_store(ch_released, my_begin + consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ch_want, my_begin + consumed + (unsigned long long)(size - n_uniq - appended), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            n_uniq += appended;
        }
    }
    if (wg == 0 && tid == 0) {                                     // the next reference starts where this one stopped (the producer commits behind the last)
        __hip_atomic_store(ch_released, my_begin + consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chain_hand_on(my_begin + consumed);
        LFD_CS(5);
#ifdef LFD_CHAIN_STAMPS
        printf("ref %d wg0: pred-begin seen +%.1f, lookup done +%.1f, own begin +%.1f, round 1 counted +%.1f, handed on +%.1f us (rounds %d; tentative first double %llu, final %llu)\n", yref,
               (double)(long long)(st_[1] - st_[0]) * 0.01, (double)(long long)(st_[2] - st_[0]) * 0.01, (double)(long long)(st_[3] - st_[0]) * 0.01,
               (double)(long long)(st_[4] - st_[0]) * 0.01, (double)(long long)(st_[5] - st_[0]) * 0.01, guard, st_[6], st_[7]);
#endif
    }

    // ---- np.unique(concat): marked cells in ascending order; thread (wg, wave, lane) owns span/64 consecutive cells ------------
    {
        const int per_lane = span >> 6;                             // multiple of 4
        const int lo = min(w_lo + lane * per_lane, w_hi), hi = min(lo + per_lane, w_hi);
        int cnt = 0;
        for (int i = lo; i < hi; i += 4) {
            if (i + 3 < hi) { const unsigned m = *reinterpret_cast<const unsigned*>(mark + i); cnt += __popc((m | (m >> 1)) & 0x01010101u); }
            else for (int e = 0; i + e < hi; ++e) cnt += (mark[i + e] & 3) != 0;
        }
        int total;
        int pos = block_excl_scan_i32(cnt, s_i, tid, total);
        if (tid == 0) g_cnt[wg] = total;
        LFD_MW_STAMP();
        LFD_GRID_SYNC();
        LFD_MW_STAMP();
    LFD_MW_STAMP();
        int base = 0, all = 0;
        for (int g = 0; g < G; ++g) { const int v = g_cnt[g]; if (g < wg) base += v; all += v; }
        if ((long long)all > A.capacity) { if (wg == 0 && tid == 0) { *A.status = LFD_SELECT_CAPACITY; *A.n_out = all; } return; }
        pos += base;
        if (cnt) for (int i = lo; i < hi; ++i) if (mark[i] & 3) A.sel_out[pos++] = (long long)i;
        if (wg == 0 && tid == 0) { *A.n_out = all; if (A.sel_offsets_out) A.sel_offsets_out[1] = A.sel_offsets_out[0] + all; }
    }
    LFD_MW_STAMP();
#undef LFD_MW_STAMP
#undef LFD_GRID_SYNC
#undef LFD_CHAIN_PASS
#undef LFD_CHAIN_BREAK
}

// =================================================================================================
// no_filter: the M largest capped certainties, in descending order (upstream: argsort(-flat)[:M],
// core/sampling.py:15-21).  NumPy's introsort leaves the order of EQUAL values unspecified; here ties
// are broken by ascending cell index.  NaN sorts last, as in NumPy.
// Radix-select the M-th largest 64-bit key {value bits, ~index} (8 rounds of 8-bit histograms), gather
// the M winners, bitonic-sort them in LDS.
// =================================================================================================
namespace {

// order-preserving unsigned image of a capped certainty: larger = earlier in the output; NaN sorts last (NumPy)
__device__ __forceinline__ unsigned topm_ord(float c, float cap) {
    c = (c > cap) ? cap : c;
    const unsigned u = __float_as_uint(c);
    if (c != c) return 0u;                                               // NaN: last
    unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // total order of finite floats
    if (ord == 0u) ord = 1u;
    return ord;
}

// workgroup-wide: the digit d (scanning bins from the top) at which the running count reaches `need`; returns d and
// leaves in need_out what is still wanted inside that bin.  hist has n_bins (<= 2 * kSelBlock) entries.
__device__ int topm_pick_bin(const unsigned* hist, int n_bins, unsigned need, int* s_i, int* s_pick, int tid, unsigned& need_out) {
    const int b_hi = n_bins - 1 - 2 * tid, b_lo = b_hi - 1;               // this thread's two bins, higher first
    const unsigned h_hi = b_hi >= 0 ? hist[b_hi] : 0u, h_lo = b_lo >= 0 ? hist[b_lo] : 0u;
    int total;
    const int excl = block_excl_scan_i32((int)(h_hi + h_lo), s_i, tid, total);
    if ((unsigned)excl < need && need <= (unsigned)excl + h_hi + h_lo) {
        if (need <= (unsigned)excl + h_hi) { s_pick[0] = b_hi; s_pick[1] = (int)(need - (unsigned)excl); }
        else { s_pick[0] = b_lo; s_pick[1] = (int)(need - (unsigned)excl - h_hi); }
    }
    __syncthreads();
    need_out = (unsigned)s_pick[1];
    return s_pick[0];
}

}  // namespace

// Three histogram passes over the certainty map itself (11 + 11 + 10 bits of the ordered value, 16-byte loads) find the
// value of the M-th largest cell; ties at that value - the norm when many cells sit at the cap - are taken in index
// order by an ordered count; the winners are then sorted in LDS.  (The first version materialised 64-bit keys and made
// eight 8-bit passes over them: 19 MB through one CU instead of 4-5 MB.)
extern "C" __global__ void __launch_bounds__(LFD_SELECT_BLOCK) lfd_select_topm_kernel(LfdSelectArgs A_launch) {
    const LfdSelectArgs A = lfd_select_args_of(A_launch);
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];   // [LFD_SELECT_TOPM_MAX]
    __shared__ unsigned s_hist[2048];
    __shared__ unsigned s_count;
    __shared__ int s_i[kSelBlock / 64];
    __shared__ int s_pick[2];
    __shared__ int s_wave_ties[kSelBlock / 64];

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = A.H * A.W;
    const int M = min(A.M, N);
    if (tid == 0) { *A.n_out = 0; *A.status = LFD_SELECT_OK; }
    if (M <= 0) return;
    if (M > LFD_SELECT_TOPM_MAX || (long long)M > A.capacity) { if (tid == 0) *A.status = LFD_SELECT_CAPACITY; return; }
    const float* cert = A.best_cert;
    const float cap = A.cap;
    const bool vec4 = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(cert) & 15u) == 0);

    // visit every cell: f(i, ord)
    auto for_each_cell = [&](auto&& f) {
        if (vec4) {
#pragma unroll 4
            for (int g = tid; g < (N >> 2); g += kSelBlock) {
                const float4 c4 = *reinterpret_cast<const float4*>(cert + 4 * g);
                f(4 * g + 0, topm_ord(c4.x, cap)); f(4 * g + 1, topm_ord(c4.y, cap));
                f(4 * g + 2, topm_ord(c4.z, cap)); f(4 * g + 3, topm_ord(c4.w, cap));
            }
        } else {
#pragma unroll 4
            for (int i = tid; i < N; i += kSelBlock) f(i, topm_ord(cert[i], cap));
        }
    };

    // ---- the M-th largest value, 11 + 11 + 10 bits at a time -------------------------------------------------------
    unsigned need = (unsigned)M;
    for (int b = tid; b < 2048; b += kSelBlock) s_hist[b] = 0;
    __syncthreads();
    for_each_cell([&](int, unsigned ord) { atomicAdd(&s_hist[ord >> 21], 1u); });
    __syncthreads();
    const unsigned d2 = (unsigned)topm_pick_bin(s_hist, 2048, need, s_i, s_pick, tid, need);
    __syncthreads();
    for (int b = tid; b < 2048; b += kSelBlock) s_hist[b] = 0;
    __syncthreads();
    for_each_cell([&](int, unsigned ord) { if ((ord >> 21) == d2) atomicAdd(&s_hist[(ord >> 10) & 0x7ffu], 1u); });
    __syncthreads();
    const unsigned d1 = (unsigned)topm_pick_bin(s_hist, 2048, need, s_i, s_pick, tid, need);
    __syncthreads();
    for (int b = tid; b < 2048; b += kSelBlock) s_hist[b] = 0;
    __syncthreads();
    const unsigned hi21 = (d2 << 11) | d1;
    for_each_cell([&](int, unsigned ord) { if ((ord >> 10) == hi21) atomicAdd(&s_hist[ord & 0x3ffu], 1u); });
    __syncthreads();
    const unsigned d0 = (unsigned)topm_pick_bin(s_hist, 1024, need, s_i, s_pick, tid, need);
    const unsigned ord_k = (hi21 << 10) | d0;            // value of the M-th largest cell
    const unsigned ties = s_hist[d0];                    // cells holding exactly that value; `need` of them are wanted
    __syncthreads();

    // ---- gather: every cell above ord_k, and the `need` lowest-indexed cells at ord_k ---------------------------------------
    if (tid == 0) s_count = 0;
    __syncthreads();
    if (need >= ties) {          // all of them: no order to respect
        for_each_cell([&](int i, unsigned ord) {
            if (ord >= ord_k) { const unsigned pos = atomicAdd(&s_count, 1u); if (pos < (unsigned)LFD_SELECT_TOPM_MAX) s_keys[pos] = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)i); }
        });
    } else {
        // each wave owns one contiguous span; ties are ranked by (wave, step, lane, cell) = index order
        constexpr int nwaves = kSelBlock / 64;
        const int span = ((N + nwaves - 1) / nwaves + 255) & ~255;
        const int w_lo = min(wave * span, N), w_hi = min(w_lo + span, N);
        int my_ties = 0;
#pragma unroll 4
        for (int i = w_lo + 4 * lane; i < w_hi; i += 256)
            for (int e = 0; e < 4 && i + e < w_hi; ++e) my_ties += topm_ord(cert[i + e], cap) == ord_k;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) my_ties += __shfl_xor(my_ties, off, 64);
        if (lane == 0) s_wave_ties[wave] = my_ties;
        __syncthreads();
        unsigned rank_base = 0;
        for (int w = 0; w < wave; ++w) rank_base += (unsigned)s_wave_ties[w];
        for (int base = w_lo; base < w_hi; base += 256) {
            const int i0 = base + 4 * lane;
            unsigned ords[4];
            int t = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ords[e] = (i0 + e < w_hi) ? topm_ord(cert[i0 + e], cap) : 0u; t += (i0 + e < w_hi) && ords[e] == ord_k; }
            int incl = t;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int n = __shfl_up(incl, off, 64); if (lane >= off) incl += n; }
            unsigned r = rank_base + (unsigned)(incl - t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (i0 + e >= w_hi) continue;
                bool take = ords[e] > ord_k;
                if (ords[e] == ord_k) { take = r < need; ++r; }
                if (take) { const unsigned pos = atomicAdd(&s_count, 1u); if (pos < (unsigned)LFD_SELECT_TOPM_MAX) s_keys[pos] = ((unsigned long long)ords[e] << 32) | (unsigned long long)(0xffffffffu - (unsigned)(i0 + e)); }
            }
            rank_base += (unsigned)__shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    // pad to a power of two with 0 (sorts last), bitonic sort descending
    int P = 1;
    while (P < M) P <<= 1;
    for (int i = M + tid; i < P; i += kSelBlock) s_keys[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += kSelBlock) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[l];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) { s_keys[i] = b; s_keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < M; i += kSelBlock) A.sel_out[i] = (long long)(0xffffffffu - (unsigned)(s_keys[i] & 0xffffffffull));
    if (tid == 0) { *A.n_out = M; if (A.sel_offsets_out) A.sel_offsets_out[1] = A.sel_offsets_out[0] + M; }
}

// seed exactly like np.random.seed(uint32): init_genrand
// {begin, end} pairs of the references of a fused sampled call: reference r's cells start at r * stride, none selected yet
extern "C" __global__ void lfd_select_begins_kernel(long long* pairs, long long stride, int n) {
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (r < n) { pairs[2 * r] = (long long)r * stride; pairs[2 * r + 1] = (long long)r * stride; }
}

extern "C" __global__ void lfd_mt_seed_kernel(unsigned* mt, unsigned seed) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
        mt[624] = 624;      // position: a twist is due before the first output
    }
}

// the same for a batch: workgroup b seeds the state of reference b (states LFD_MT_STATE_STRIDE words apart)
extern "C" __global__ void lfd_mt_seed_batch_kernel(unsigned* mt_base, LfdSeedBatch seeds) {
    if (threadIdx.x == 0) {
        unsigned* mt = mt_base + (size_t)blockIdx.x * LFD_MT_STATE_STRIDE;
        mt[0] = seeds.seed[blockIdx.x];
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
        mt[624] = 624;
    }
}
