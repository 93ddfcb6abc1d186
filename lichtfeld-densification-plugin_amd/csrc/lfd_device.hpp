// Structures shared between the host API (lfd_api.hip) and the kernels (lfd_kernels.hip).
#pragma once

#include <stdint.h>

#include "lfd_geometry.hpp"

#ifndef LFD_MAX_SLOTS
#define LFD_MAX_SLOTS 16
#endif

#ifndef LFD_DENSE_BLOCK
#define LFD_DENSE_BLOCK 256     // threads per workgroup of the fused dense kernel (4 waves)
#endif
#ifndef LFD_COPY_UNROLL
#define LFD_COPY_UNROLL 2          // survivor records a thread copies out per step (dense kernel)
#endif
#ifndef LFD_TICKET_LANES
#define LFD_TICKET_LANES 8        // interleaved ticket sequences of the ticketed dense kernel (one per XCD)
#endif
#define LFD_DENSE_CPT 4         // consecutive grid cells per thread (one 16-byte certainty load per slot)
#ifndef LFD_DENSE_WAVES_PER_SIMD
#define LFD_DENSE_WAVES_PER_SIMD 8       // register budget of the fused kernel: 512/8 = 64 VGPRs.  Round 3: the geometry loop fits (63-64, no scratch) since
                                         // nothing constant is kept in vector registers across it any more, and the tile's LDS block is 20.4 KB: eight workgroups per CU (7: 0.2819, 8: see profiles/history.md (r3/ablation.txt))
#endif
#ifndef LFD_DENSE_ALL_WARPS
#define LFD_DENSE_ALL_WARPS 0   // dense kernel, references with at most this many neighbours (two-channel warps, no masks): the warps of ALL slots ride
                                // along with the certainty planes - one dependent memory round trip less for 8 (k-1) B per cell more traffic.
                                // Measured in round 2 (profiles/history.md (r2/ablation.txt)): k = 1 0.118 -> 0.116 ms, k = 2 0.307 -> 0.290, k = 3 0.317 -> 0.307 (but
                                // 1.44 x the algorithmic bytes instead of 1.10 x), k = 4 slower: it was on for up to two neighbours until round 6.  With the winner's
                                // warp fetched by consecutive lanes (LFD_WARP_BY_LANE) the dependent round trip is cheap enough: 0 (never) now measures -2.0 % at
                                // k = 2 and -0.5 % at k = 1 against 2 (profiles/r6/ab_warp_by_lane.txt).
#endif
#ifndef LFD_FRONT_PRIO
#define LFD_FRONT_PRIO 1        // s_setprio of a dense-kernel wave until its geometry loop starts (0 = off): the handful of instructions between the
                                // front end's memory requests then go ahead of older waves' f64 streams instead of waiting for a free slot
                                // (profiles/history.md (r2/ablation.txt): 0.331 -> 0.318 ms)
#endif
#define LFD_INDEXED_BLOCK 1024  // one workgroup (16 waves) per reference in the indexed kernel
#define LFD_INDEXED_EVAL_BLOCK 256   // cells per workgroup of the indexed-mode evaluation kernel

#define LFD_EPOCH_BITS 22
#define LFD_EPOCH_MASK ((1u << LFD_EPOCH_BITS) - 1u)
#define LFD_VALUE_BITS 40     // survivors-so-far fits 40 bits (1e12 points)
#ifndef LFD_LOOKBACK_PER_LANE
#define LFD_LOOKBACK_PER_LANE 1    // tile-state words a lane of the look-back wave reads per round trip (window = 64 x this)
#endif
#ifndef LFD_STAGGER_UNITS
#define LFD_STAGGER_UNITS 48    // dense kernel: start-up stagger per resident-workgroup slot of a CU (x64 cycles; 0 = off)
#endif
#ifndef LFD_NT_LOADS
#define LFD_NT_LOADS 0          // certainty / warp planes with the non-temporal hint: measured slower (0.312 against 0.304)
#endif
#ifndef LFD_NT_STORES
#define LFD_NT_STORES 1         // dense kernel's records (written once, read by nobody on the device) with the non-temporal hint: 0.304 -> 0.300
#endif
#ifndef LFD_LOOKBACK_LANES
#define LFD_LOOKBACK_LANES 16      // lanes of the look-back wave that read a tile-state word per round trip (the window).  The words are read past the caches: 16 per round measured 0.306 ms, 32 0.308, 64 0.311, 8 0.316; 128 ... 1024 (several words per lane) 0.32 ... 0.56
#endif
#ifndef LFD_WARP_BY_LANE
#define LFD_WARP_BY_LANE 1         // dense kernel: the winner's warp fetched by consecutive lanes for consecutive cells (slots exchanged through LDS inside the wave)
#endif
#ifndef LFD_LOOKBACK_WATCH_ONE
#define LFD_LOOKBACK_WATCH_ONE 1     // look-back: one lane watches the nearest predecessor's word until it is published, then the window is read
#endif
#ifndef LFD_WATCH_SLEEP
#define LFD_WATCH_SLEEP 127          // s_sleep argument (x64 cycles, the instruction's maximum: ~4 us) between two looks at the watched word
#endif
#ifndef LFD_POLL_SLEEP
#define LFD_POLL_SLEEP 8       // s_sleep argument (x64 cycles) between two polls of a look-back window
#endif
#define LFD_SPIN_LIMIT (1u << 24)   // ~ seconds of polling: a look-back that starves reports instead of hanging
#define LFD_LAUNCH_TIMEOUT 1u

struct LfdRefDesc {             // one per reference of a launch (device table)
    const uint8_t* image;       // u8 [h_match][w_match][3]
    const uint8_t* mask_a;      // u8 {0,1} [h_match][w_match] or null
    int32_t cam;                // row of the camera table
    int32_t n_slots;            // valid neighbour slots
    int32_t any_mask;           // 1 if mask_a or any mask_b of the reference is set (one word decides the kernels' plain path)
    int32_t pad;
};

struct LfdSlotDesc {            // one per (reference, slot)
    const float* cert;          // f32 [H*W]
    const float* warp;          // f32 [H*W*C]
    const uint8_t* mask_b;      // u8 {0,1} [h_match][w_match] or null
    int32_t cam;
    int32_t pad;
};

#define LFD_COPY_MAX_SEGS 96
#define LFD_COPY_CHUNK 32768       // bytes one workgroup of lfd_copy_segments_kernel moves
struct LfdCopyArgs {               // by value in the kernel arguments (2.7 KB): no upload, no allocation
    long long src[LFD_COPY_MAX_SEGS], dst[LFD_COPY_MAX_SEGS], n[LFD_COPY_MAX_SEGS];
    int chunk0[LFD_COPY_MAX_SEGS + 1];      // first workgroup of every segment (exclusive prefix of ceil(n / LFD_COPY_CHUNK))
    int n_segs;
};

struct LfdTileSeg { int32_t offset, count; };     // == lfd_tile_segment of the C-ABI

struct LfdLaunch {              // kernel argument, passed by value
    const LfdCam* cams;
    const LfdRefDesc* refs;
    const LfdSlotDesc* slots;
    const LfdRefConst* ref_const;    // [n_refs]    written by lfd_pair_setup_kernel
    const LfdPairConst* pair_const;  // [n_refs*k]
    const float* axis_x;        // [W]
    const float* axis_y;        // [H]
    const float* fund_override; // [n_refs*k*9] f32 fundamental matrices handed over by the caller (lfd_batch.fundamental), or null
    int32_t n_refs, k, H, W, w_match, h_match, warp_channels, tiles_per_ref;
    float mask_sx, mask_sy;     // (float)w_match/(float)W, (float)h_match/(float)H  (nearest resize)
    float inv_w;                // 1.0f / W (cell -> row estimate)
    int32_t axis_identity;      // 1: axis_x/axis_y hold lfd_identity_axis() values, which the kernels may compute (ax, ay) instead of loading
    LfdAxis ax, ay;             // the analytic A-grid axes (valid when axis_identity)
    int32_t w_log2;             // log2(W) when W is a power of two (cell -> row / column by shift and mask), else -1
    int32_t pad1;
    LfdKernelParams kp;
    // outputs
    float* xyz;
    float* rgb;
    float* err;
    int32_t* cell;
    uint8_t* slot;
    long long capacity;
    long long* ref_offsets;
    int32_t* seg_counts;
    // look-back workspace.  The ticket counter is never reset (ticket_base = its value at launch)
    // and tile-state words carry the launch epoch, so nothing is memset between launches.
    unsigned long long* tile_state;
    unsigned long long* ticket;
    unsigned long long* ticket_lanes;   // dense kernel: sequence s counts at ticket_lanes[16 * s] (one cache line each)
    unsigned long long ticket_base;                         // indexed kernel: one sequence
    unsigned long long ticket_base_lane[LFD_TICKET_LANES];   // dense kernel: value of each sequence's counter at launch
    unsigned int epoch;
    unsigned int pad0;
    unsigned int* status;         // 0 = ok, LFD_LAUNCH_TIMEOUT if a look-back spin gave up
    unsigned int* seg_ready;      // == epoch once the workgroup of tile 0 has zeroed seg_counts
    const LfdColourCol* colour_cols;    // dense mode, analytic A-grid, two-channel warps: [W] / [H] colour tables (lfd_geometry.hpp), else null
    const LfdColourRow* colour_rows;
    unsigned long long* phase_stamps;   // profiling builds (-DLFD_DENSE_TIMING) with LFD_DENSE_TIMING set in the environment: [n_tiles][16] clock stamps, else null
    // unordered retirement (lfd_triangulate_dense_segments): a tile claims room in ITS REFERENCE's region [r*H*W, (r+1)*H*W) of the output with
    // one atomic on the reference's cursor and records where it went; no look-back.  Both null for the ordered kernels.
    unsigned long long* ref_cursor;     // [n_refs] survivors claimed so far per reference (zeroed by the workgroup of tile 0, like seg_counts)
    LfdTileSeg* tile_table;             // [n_refs * tiles_per_ref] {offset inside the reference's region, survivors} of every tile
    // file-payload output (lfd_triangulate_dense_ply): 15-byte PLY vertex records [capacity * 15] instead of xyz / rgb / err (those are null then)
    unsigned char* ply;
};

// ---- S: on-device coverage sampling (lfd_select.hip) ---------------------------------------------
#define LFD_SELECT_BLOCK 1024
#define LFD_SELECT_MAX_BINS 2304     // coverage tiles a map may have: every SQUARE grid fits (the most: 47 x 47 at one cell per tile; RoMa's grids have 576 ... 625)
#define LFD_SELECT_TOPM_MAX 16384    // no_filter: winners sorted in LDS (128 KiB)
#ifndef LFD_SELECT_DEFAULT_WG
#define LFD_SELECT_DEFAULT_WG 16      // compute workgroups the selection uses by default (0: single-workgroup kernel)
#endif
#define LFD_SELECT_MAX_WG 64         // compute workgroups of the multi-workgroup selection kernel
// shared scratch of the multi-workgroup selection kernel (byte offsets)
#define LFD_COOP_BAR 0               // u32 arrivals, u32 generation
#define LFD_COOP_FLAGS 16            // i32 nz, inexact, negative, nan
#define LFD_COOP_MSG 32              // u32 draws_ready (round whose draws are in place), u32 go (1 commit / 2 abort), u64 request {round << 32 | need}, round 0xffffffff = done
#define LFD_COOP_PART 64             // f64 [MAX_WG] partial sums of the weights
#define LFD_COOP_SPAN (LFD_COOP_PART + 8 * LFD_SELECT_MAX_WG)                       // f64 [MAX_WG * 16] span sums of the live p
#define LFD_COOP_WGCNT (LFD_COOP_SPAN + 8 * 16 * LFD_SELECT_MAX_WG)                 // i32 [MAX_WG + 1] per-workgroup counts of an ordered compaction
#define LFD_COOP_BINS ((LFD_COOP_WGCNT + 4 * (LFD_SELECT_MAX_WG + 1) + 7) & ~7)     // u64 [MAX_BINS] coverage bins
#define LFD_COOP_MT (LFD_COOP_BINS + 8 * LFD_SELECT_MAX_BINS)                       // u32 [625] speculative copy of the MT19937 state
#define LFD_SELECT_COOP_BYTES (LFD_COOP_MT + 4 * 640)
#define LFD_SELECT_OK 0
#define LFD_SELECT_NAN 1             // a weight is NaN           (upstream: ValueError from np.random.choice)
#define LFD_SELECT_NEGATIVE 2        // a weight is negative      (upstream: ValueError)
#define LFD_SELECT_FEWER_NONZERO 3   // fewer non-zero weights than draws requested (upstream: ValueError)
#define LFD_SELECT_INEXACT 4         // a weight < 2^-29: the parallel cumsum would not be exact
#define LFD_SELECT_NO_PROGRESS 5
#define LFD_SELECT_TOO_MANY_BINS 6
#define LFD_SELECT_CAPACITY 7

#define LFD_SELECT_BATCH_MAX 32       // references per selection launch (17 workgroups each: co-resident with room to spare)
struct LfdSelectArgs {
    const float* best_cert;   // [H*W] aggregated certainty of ONE reference (output of lfd_aggregate)
    float* weights;           // [H*W] scratch: capped, border-masked, normalised f32 weights
    double* p;                // [H*W] scratch
    double* cdf;              // [H*W] scratch
    int* first;               // [H*W] scratch
    unsigned char* mark;      // [H*W] scratch
    double* draws;            // [M]   scratch
    int* cand;                // [M]   scratch
    int* found;               // [M]   scratch
    unsigned* mt;             // [625] MT19937 key + position (the context's legacy stream)
    long long* sel_out;       // [capacity] selected cells, ascending
    int* n_out;               // [1]
    int* status;              // [1]
    long long capacity;
    int H, W, M, border, tiles;
    float cap;
    float s_override;         // > 0: use this normaliser instead of the exact device sum (parity tests)
    unsigned long long* timing;   // profiling: wall_clock64() at the phase boundaries of the filter kernel, or null
    unsigned char* coop;          // multi-workgroup kernel: shared scratch (LFD_SELECT_COOP_BYTES, header zeroed before the launch)
    int n_wg;                     // multi-workgroup kernel: compute workgroups (the grid has one more, which runs the MT19937 stream)
    long long* sel_offsets_out;   // optional device i64 [2]: {begin, begin + n_out} for the indexed kernels that follow in the same stream ([0] = [1] = begin set by the host)
    // Several references in one launch (blockIdx.y = reference, every pointer above is reference 0's): the kernels move on to
    // their reference's block first thing (lfd_select_args_of).  All zero / null for a launch of one reference.
    long long batch_scratch_stride;   // bytes between the scratch blocks (weights ... coop, n_out / status) of consecutive references
    long long batch_cert_stride;      // elements between their aggregated-certainty maps
    long long batch_out_stride;       // elements between their sel_out areas
    long long batch_mt_stride;        // words between their MT19937 states (one stream per reference)
    int* batch_info;                  // device i32 [2 * n] or null: reference y reports {n_out, status} at [2y], [2y + 1] instead of n_out / status
    // The multi-workgroup kernel's view of the MT19937 stream - an ARRAY: one producer workgroup (the extra workgroup of reference 0) writes its
    // doubles, by absolute index, into `ring`.  With several references in one launch on ONE stream (lfd_triangulate_sampled_chain, batch_mt_stride
    // = 0) reference y uses the doubles from where reference y - 1 stopped, an offset it learns from the chain block when y - 1 is through with
    // its last round; everything that does not depend on the stream (weights, p, the first cumulative sum) runs side by side.  With one stream PER
    // reference (lfd_triangulate_sampled_multi) every reference is a launch of its own as far as this goes: chain / ring / snaps at a stride.
    unsigned char* chain;             // the launch's chain block (LFD_CHAIN_* byte offsets, zeroed before the launch)
    double* ring;                     // [ring_cap] double i of the stream at i & (ring_cap - 1)
    unsigned* snaps;                  // [snap_slots * 624] the key after the t-th twist at slot t % snap_slots (what the producer commits from)
    long long ring_cap;               // a power of two >= 4 * int(M * 0.85)
    int snap_slots;
    int chain_refs;                   // references that share this chain: the whole launch (one stream), or 1 (a stream per reference)
    long long batch_chain_stride;     // bytes between the chain / ring / snaps sets of consecutive references (0: one set for the launch)
};
// per-reference normalisers of a batched launch, a kernel argument of its own: read with a run-time index straight from the argument segment (inside
// LfdSelectArgs - which every kernel copies and edits for its reference - the array would drag the whole structure into scratch memory)
struct LfdSelectNorms {
    int use;                              // the normaliser of reference y is s[y] (> 0: overrides the exact device sum) instead of s_override
    float s[LFD_SELECT_BATCH_MAX];
};
#define LFD_MT_STATE_STRIDE 640       // words per MT19937 state of a batch (624 key + position, padded)
// chain block (byte offsets; u64 words unless noted)
#define LFD_CHAIN_PRODUCED 0          // doubles of the stream written so far
#define LFD_CHAIN_WANT 8              // end (absolute index) of the draws the reference at work has asked for
#define LFD_CHAIN_RELEASED 16         // doubles below this index are not read any more
#define LFD_CHAIN_STATE 24            // u32: 0 producer at work, 1 stream committed, 2 producer gave up / chain broken (stream left where it was)
#define LFD_CHAIN_CURRENT 28          // u32: the reference of the launch that is drawing (the producer looks a whole first round ahead only while another follows)
#define LFD_CHAIN_OFF 64              // [LFD_SELECT_BATCH_MAX + 1]: 1 + the absolute index of reference y's first draw; 0 = not known yet
#define LFD_CHAIN_BROKEN (~0ull)      //   ... or this: a predecessor failed, nobody knows where the stream stands
#define LFD_CHAIN_BEG 384             // [LFD_SELECT_BATCH_MAX + 1]: the same number, published by reference y ITSELF when it starts drawing (its follower looks a window behind that first round up in advance)
#define LFD_CHAIN_TENT 656            // [LFD_SELECT_BATCH_MAX + 1]: 1 + where reference y starts IF its predecessor's second round is its last (published when the predecessor's first round is counted)
#define LFD_CHAIN_BYTES 1024
struct LfdSeedBatch { unsigned seed[LFD_SELECT_BATCH_MAX]; };
