/*
 * lfd_densify.h -- C ABI of the MI355X-native dense-initialisation hot path.
 *
 * The upstream plugin (shadygm/Lichtfeld-Densification-Plugin) has no FFI: its boundary is a set of
 * Python call signatures inside one interpreter.  This header is the C-ABI a maintainer would bind
 * (ctypes / cffi / pybind) to replace, per reference view, the CPU stage
 *
 *     core/pipeline.py:405-442   _collect_reference_matches epilogue (certainty floor, masks, D2H)
 *     core/pipeline.py:602-780   _triangulate_ref
 *     core/geometry.py:53-141    skew / DLT / reprojection / cheirality / parallax / F / Sampson
 *     core/sampling.py:8-53      select_samples_with_coverage        (lfd_select_*: see below)
 *     core/writers.py:15-46      write_ply / write_points3D_bin      (lfd_pack_*)
 *
 * with hand-written HIP kernels for gfx950.  All tensor arguments are raw DEVICE pointers unless the
 * comment says "host"; nothing here depends on torch.  Every function returns LFD_OK (0) or an
 * error code and records a message retrievable with lfd_last_error(); nothing aborts.  A context is
 * bound to one HIP device and one stream; use one context per thread (no hidden globals).
 *
 * Every entry point that computes needs a GPU and fails with LFD_ERR_HIP when none is present; nothing falls back to
 * the host.  The CPU twin at the end of this header (lfd_create_host + the *_host entry points: the host build of
 * the same per-cell source over host arrays) is a separate, explicitly chosen interface for upstream's CPU-only
 * configuration, for timing a CPU baseline and for parity checks.
 */
#ifndef LFD_DENSIFY_H
#define LFD_DENSIFY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFD_ABI_VERSION 9
#define LFD_MAX_SLOTS 16 /* neighbours per reference handled by one launch */

enum lfd_status {
    LFD_OK = 0,
    LFD_ERR_INVALID = 1,  /* bad argument */
    LFD_ERR_HIP = 2,      /* HIP runtime error / no device */
    LFD_ERR_CAPACITY = 3, /* output buffers too small (counts are still valid) */
    LFD_ERR_STATE = 4     /* call order (e.g. cameras not uploaded) */
};

typedef struct lfd_context lfd_context;

/* Thresholds of DensePipelineConfig that reach the kernels (core/config.py:7-26).  Comparison
 * dtypes follow upstream: Sampson f64 `<`, reprojection f32 `<=`, depth f32 `> 0`, parallax f32 `>=`. */
typedef struct lfd_params {
    double sampson_thresh;  /* px^2; <= 0 disables the Sampson gate (core/pipeline.py:708)          */
    float certainty_thresh; /* FLOOR applied to certainty, not a reject (core/pipeline.py:407)       */
    float sample_cap;       /* RomaMatcher.sample_thresh = 0.9 (core/matcher.py:92)                  */
    float reproj_thresh;    /* px (core/pipeline.py:745)                                             */
    float min_parallax_deg; /* degrees; <= 0 disables (core/pipeline.py:748)                         */
    int32_t no_filter;      /* keep every finite point (core/pipeline.py:739-743)                    */
    int32_t flags;          /* LFD_FLAG_* below, 0 = defaults                                        */
} lfd_params;

/* lfd_triangulate_dense blends the four colour taps in f32 by default (same taps, weights and order as upstream
 * core/pipeline.py:661-679, within 2.5e-7 of upstream's f64 blend).  With this flag it runs upstream's f64
 * arithmetic: rgb is then bit-identical to upstream's, at ~6 % more kernel time.  The upstream-equivalent entry
 * points (lfd_triangulate_indexed / _sampled) always use the f64 form. */
#define LFD_FLAG_EXACT_COLOUR 1

/* One launch = n_refs reference views, each with up to k neighbour slots (slots [0, n_slots[r]) are
 * valid, in the order upstream's `nn_ids` lists the loaded neighbours).  Arrays marked "host" are
 * host arrays whose ELEMENTS are device pointers. */
typedef struct lfd_batch {
    int32_t n_refs;
    int32_t k;                    /* slot stride of the per-slot arrays, 1..LFD_MAX_SLOTS            */
    int32_t H, W;                 /* RoMa output grid                                                */
    int32_t w_match, h_match;     /* matcher.w_resized / h_resized (pixel conversion, image, masks)  */
    int32_t warp_channels;        /* 4: [xA,yA,xB,yB] as upstream's matcher emits; 2: [xB,yB] only   */
    int32_t reserved;
    const int32_t* ref_cam;       /* host [n_refs]      index into the uploaded camera table         */
    const int32_t* n_slots;       /* host [n_refs]      valid slots of each reference (<= k)         */
    const int32_t* nbr_cam;       /* host [n_refs*k]    camera index of every slot                   */
    const float* const* cert;     /* host [n_refs*k] -> device f32 [H*W]   raw certainty (pre-floor) */
    const float* const* warp;     /* host [n_refs*k] -> device f32 [H*W*warp_channels], normalised   */
    const uint8_t* const* image;  /* host [n_refs]   -> device u8 [h_match*w_match*3] (RGB, resized) */
    const uint8_t* const* mask_a; /* NULL, or host [n_refs]   -> device u8 {0,1} [h_match*w_match] or NULL */
    const uint8_t* const* mask_b; /* NULL, or host [n_refs*k] -> device u8 {0,1} [h_match*w_match] or NULL */
    const float* axis_x;          /* device f32 [W]: A-grid x of column j (torch.linspace(-1+1/W,1-1/W,W)); */
    const float* axis_y;          /* device f32 [H]; both NULL -> lfd_identity_axis() values. Used when warp_channels==2 */
    const float* fundamental;     /* NULL, or host f32 [n_refs*k*9]: F of every (reference, slot) pair as upstream's
                                   * fundamental_from_world2cam returns it (core/geometry.py:122-130, row-major), used instead of
                                   * the F the library derives from the camera table.  The library's own F is built with the
                                   * closed-form inverse of K (exactly 1/fx, -cx/fx ...) where upstream calls np.linalg.inv
                                   * (LAPACK sgetrf/sgetri); the two agree to ~2e-6 relative, not bit for bit, which moves a
                                   * Sampson value next to the threshold by ~1e-5.  A caller that already holds upstream's F
                                   * (integration path B in INTEGRATION.md) passes it here and gets upstream's Sampson decisions. */
} lfd_batch;

/* Survivors.  Capacity is in points.  cell / slot are optional (NULL to skip). */
typedef struct lfd_points {
    float* xyz;     /* [capacity*3]                                                                  */
    float* rgb;     /* [capacity*3]  f32 in [0,1] (quantised only by the writers, as upstream)       */
    float* err;     /* [capacity]    max reprojection error of the two views, px                     */
    int32_t* cell;  /* [capacity]    flat grid index y*W+x                                            */
    uint8_t* slot;  /* [capacity]    neighbour slot that won the arg-max                              */
    int64_t capacity;
} lfd_points;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int lfd_abi_version(void);
int lfd_create(int device_index, void* hip_stream, lfd_context** out);
void lfd_destroy(lfd_context* ctx);
int lfd_set_stream(lfd_context* ctx, void* hip_stream);
/* The profiling / A-B switches of the environment (LFD_DENSE_EXTRA_LDS, LFD_INDEXED_SPLIT, LFD_SELECT_WORKGROUPS, LFD_SELECT_TIMING,
 * LFD_DENSE_TIMING) are read when a context is created, never on a launch path; a test or profiling script that changes them for a
 * live context calls this to have them read again. */
int lfd_reload_env(lfd_context* ctx);
/* Measurement: the device-side duration of the dense kernel's launches.  lfd_kernel_timing(ctx, n) makes the next n launches of
 * lfd_triangulate_dense carry a start and a stop event of their own (recorded by the command processor where the kernel begins and
 * ends: the figure a kernel trace reports, without the dispatch gaps that events recorded around the call include); n = 0 switches it
 * off.  lfd_kernel_timing_read waits for the last timed launch and writes the durations in launch order (milliseconds; at most
 * `capacity`, *n_out = how many), then starts a new series.  Asynchronous launches stay asynchronous. */
int lfd_kernel_timing(lfd_context* ctx, int32_t n_launches);
int lfd_kernel_timing_read(lfd_context* ctx, float* ms, int32_t capacity, int32_t* n_out);
const char* lfd_last_error(const lfd_context* ctx); /* ctx may be NULL: last creation error */
/* How THIS build of the library lays out the structures of this header, so that a binding that mirrors them by hand (ctypes, cffi ABI mode,
 * JNA ...) can check itself at load time instead of trusting a transcription: for lfd_params, lfd_batch, lfd_points, lfd_tile_segment and
 * lfd_copy_segment, in that order, {sizeof, number of fields, then offsetof(field), sizeof(field) for every field in declaration order}.  Writes at most `capacity`
 * values to `out` (host int32; may be NULL with capacity 0) and returns how many values the table has.  The Python mirror compares its
 * ctypes `_fields_` with it when it loads the library (core/hip_backend.py::check_struct_layout). */
int lfd_struct_layout(int32_t* out, int32_t capacity);
/* ... and the names behind those numbers: "lfd_params:sampson_thresh,certainty_thresh,...;lfd_batch:n_refs,...;..." - the same structures and
 * fields in the same order (two neighbouring fields of one type swapped in a mirror keep every offset; their names tell). */
const char* lfd_struct_fields(void);

/* Camera table, all host f32 row-major as upstream's CameraRecord holds them
 * (core/camera_models.py:10-28): K[n][9] R[n][9] t[n][3] P[n][12] C[n][3], wh[n][2] = width,height. */
int lfd_upload_cameras(lfd_context* ctx, int32_t n, const float* K, const float* R, const float* t,
                       const float* P, const float* C, const int32_t* wh);

/* ---- the hot path ------------------------------------------------------------------------------ */
/* Optional: the batch-dependent preparation of a launch on its own - validation, upload of the descriptor tables (skipped when the
 * device already holds the same tables) and the per-(reference, neighbour) constants (F5: fundamental matrices, projection blocks,
 * core/geometry.py:53-55,122-130) - stream-ordered, asynchronous.  Every compute entry point below does the same itself when the
 * batch differs from the last one it saw; calling this first only moves that work (a driver can issue it for reference i+1 while
 * reference i computes, and a benchmark can time the kernels apart from it). */
int lfd_prepare_batch(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params);

/* P1+F1: certainty floor, masks, per-cell arg-max over the neighbours (first maximum wins).
 * best_cert: device f32 [n_refs*H*W]; best_slot: device u8 [n_refs*H*W] or NULL. */
int lfd_aggregate(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params,
                  float* best_cert, uint8_t* best_slot);

/* Fused dense kernel: every grid cell -> floor/masks/arg-max -> Sampson -> DLT -> reprojection,
 * cheirality, parallax -> colour -> ordered compaction.  Survivors are emitted per reference in
 * raster order.  Candidates are the cells upstream's sampler could draw (core/sampling.py:24-27, 41-43:
 * a cell whose best certainty after floor and masks is <= 0 - masked out - has weight 0 there) plus the
 * two-cell border upstream keeps out of its draw; a masked-out cell is never triangulated.  ref_offsets: device i64 [n_refs+1] (exclusive prefix of survivors per reference;
 * last = total); seg_counts: device i32 [n_refs*k] survivors per (reference, slot) or NULL. */
int lfd_triangulate_dense(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params,
                          const lfd_points* out, int64_t* ref_offsets, int32_t* seg_counts);

/* The same kernel writing the FILE PAYLOAD itself: survivors leave as 15-byte PLY vertex records (x y z f32 LE, r g b u8 quantised like
 * upstream's to_uint8_rgb - what lfd_pack_ply makes of lfd_triangulate_dense's arrays, byte for byte) in raster order per reference, so a run
 * whose consumer is the PLY writer (streamed output, the exchange of a sharded run) needs no packing pass and writes 15 instead of 28 bytes
 * per survivor.  records: device u8 [capacity * 15]; the reprojection error is not part of a PLY vertex and is not produced; cell / slot:
 * optional as in the lfd_points structure - NULL to skip.  Replaces core/pipeline.py:753-780 + core/writers.py:29-46 for that consumer. */
int lfd_triangulate_dense_ply(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, uint8_t* records, int64_t capacity,
                              int64_t* ref_offsets, int32_t* seg_counts, int32_t* cell, uint8_t* slot);

/* The same kernel with UNORDERED RETIREMENT (opt-in; the default entry point above stays ordered).  The ordered kernel makes a tile wait
 * for the survivor counts of every tile before it (a decoupled look-back: a fifth of a tile's life on the benchmark shape); here a tile
 * claims room with ONE atomic on its reference's cursor and records where it went:
 *   - reference r owns the region [r*H*W, (r+1)*H*W) of out (out->capacity >= n_refs*H*W, else LFD_ERR_CAPACITY); its survivors fill
 *     [r*H*W, r*H*W + ref_counts[r]) tile after tile in the order the tiles RETIRED, raster order inside a tile;
 *   - table[r * lfd_dense_tiles_per_ref(H, W) + t] = {offset inside the reference's region, survivors} of the reference's t-th tile
 *     (tile t covers grid cells [1024 t, 1024 t + 1024));
 *   - ref_counts: device i64 [n_refs] survivors per reference (it is the cursor array: zeroed by the launch itself).
 * Raster order is tile order, so the consumers below restore upstream's sequence from the table: lfd_order_segments writes the ordered
 * structure-of-arrays result (bit-identical to lfd_triangulate_dense's), lfd_pack_ply_segments / lfd_pack_points3d_segments write the
 * file payload in raster order straight from the unordered buffers.  The reference interface both forms replace is the per-group
 * concatenation of core/pipeline.py:753-780,917-919. */
typedef struct lfd_tile_segment { int32_t offset; int32_t count; } lfd_tile_segment;
#define LFD_FLAG_TILE_SEGMENTS 2 /* informational: set in lfd_params.flags by callers that take the unordered route (the entry point decides) */
int lfd_dense_tiles_per_ref(int32_t H, int32_t W); /* host helper: rows of the tile table per reference */
int lfd_triangulate_dense_segments(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, const lfd_points* out,
                                   int64_t* ref_counts, int32_t* seg_counts, lfd_tile_segment* table);
/* Both at once: the 15-byte PLY vertex records of lfd_triangulate_dense_ply with the unordered retirement of lfd_triangulate_dense_segments, for a
 * consumer that wants every reference's point SET as file payload and does not care about its raster order (a point cloud has none; the streamed
 * output of a run that does not have to reproduce upstream's byte sequence).  records: device u8 [capacity * 15], capacity >= n_refs*H*W;
 * reference r's records fill [15*r*H*W, 15*(r*H*W + ref_counts[r])) tile after tile in retirement order, raster order inside a tile; table and
 * ref_counts as above.  The same points, byte for byte per record, as lfd_triangulate_dense_ply emits (tests/test_gpu_segments.py). */
int lfd_triangulate_dense_ply_segments(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, uint8_t* records, int64_t capacity,
                                       int64_t* ref_counts, int32_t* seg_counts, lfd_tile_segment* table);
/* src: the unordered buffers of lfd_triangulate_dense_segments (capacity ignored); dst: ordered result, at most dst->capacity records
 * (cell / slot copied when both sides give them); ref_offsets: device i64 [n_refs + 1] or NULL.  Asynchronous on the context's stream. */
int lfd_order_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const lfd_points* src,
                       const lfd_points* dst, int64_t* ref_offsets);
/* lfd_pack_ply / lfd_pack_points3d through the table: out receives min(total, capacity) records in raster order (ids count from
 * id_base + 1 in that order); ref_offsets as above (the caller reads the total there). */
int lfd_pack_ply_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const float* xyz,
                          const float* rgb, int64_t capacity, uint8_t* out, int64_t* ref_offsets);
int lfd_pack_points3d_segments(lfd_context* ctx, int32_t n_refs, int32_t H, int32_t W, const lfd_tile_segment* table, const float* xyz,
                               const float* rgb, const float* err, int64_t capacity, uint64_t id_base, uint8_t* out, int64_t* ref_offsets);

/* Upstream-equivalent mode: only the selected cells (sel_idx: device i64, concatenated per
 * reference; sel_offsets: host i64 [n_refs+1]) are triangulated, and survivors are emitted in
 * upstream's order: per reference, neighbour groups in order of first appearance while scanning
 * sel_idx, members in sel_idx order (core/pipeline.py:685-780).  seg_order: device i32 [n_refs*k]
 * or NULL: the slot of the g-th group (or -1). */
int lfd_triangulate_indexed(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params,
                            const int64_t* sel_idx, const int64_t* sel_offsets, const lfd_points* out,
                            int64_t* ref_offsets, int32_t* seg_counts, int32_t* seg_order);

/* One reference view (batch->n_refs == 1) through upstream's whole per-reference stage (core/pipeline.py:602-780,
 * `_triangulate_ref`) in ONE stream-ordered call with nothing read back in between: lfd_aggregate -> selection (coverage
 * sampling on the context's MT19937 stream, or the top-M of the no_filter mode when params->no_filter) ->
 * lfd_triangulate_indexed on the cells selected.  Equivalent to the three calls, bit for bit.
 * out->capacity >= M + tiles*tiles + 64.  sel_info: device i32 [3] = {cells selected, selection status (0 = ok, else
 * the LFD_SELECT_* code upstream would have raised for), launch status (what lfd_launch_status would report: 0 = ok)};
 * with a non-zero selection status or an empty selection no point is emitted (ref_offsets = {0, 0}).  sel_cells: device i64 [M + tiles*tiles + 64] receiving the selected cells, or NULL. */
int lfd_triangulate_sampled(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                            int32_t border, int32_t tiles, float s_override, const lfd_points* out, int64_t* ref_offsets,
                            int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info, int64_t* sel_cells);

/* The same for SEVERAL reference views per call, each drawing from its own MT19937 stream seeded with seeds[r] (host u32
 * [n_refs]) like np.random.seed(seeds[r]) - the per-reference streams of the multi-GPU / per_reference_rng mode, where no
 * reference depends on another one's draws.  One aggregate launch and one pair of indexed launches serve the whole batch; the
 * selections run one after the other.  out->capacity >= n_refs * (M + tiles*tiles + 64).  sel_info: device i32 [2*n_refs + 1] =
 * {cells selected, selection status} per reference, then the launch status.  sel_cells: device i64 [n_refs * (M + tiles*tiles +
 * 64)] (reference r's cells start at r * (M + tiles*tiles + 64)) or NULL.  The context's own stream is left seeded with the last
 * reference's seed. */
int lfd_triangulate_sampled_multi(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                                  int32_t border, int32_t tiles, const uint32_t* seeds, const lfd_points* out,
                                  int64_t* ref_offsets, int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info,
                                  int64_t* sel_cells);

/* SEVERAL reference views per call on the CONTEXT's one MT19937 stream - upstream's single global stream (np.random.seed(config.seed) once,
 * core/pipeline.py:793; every reference's np.random.choice continues where the one before it stopped, core/sampling.py:32) - consumed in
 * batch order: the points, the cells selected and the position the stream is left at are those of n_refs successive
 * lfd_triangulate_sampled calls, bit for bit.  What does not depend on the stream (weights, probabilities, the first cumulative sum of every
 * reference) runs side by side; a reference starts drawing where the one before it stopped.  A reference whose selection refuses its input
 * (selection status 1-3: upstream raises before it draws) consumes nothing and emits nothing, the others are not affected.  A status of 4-7 on
 * any reference is THIS implementation's refusal, not upstream's: the call is VOID as a whole (a chain whose bounded waits expire - 5 - commits
 * nothing to the stream although its first references have drawn; a reference refused for inexactness - 4 - draws nothing although upstream would
 * have).  Callers that read the status behind further launches take lfd_rng_checkpoint before the call and, on such a status, roll back and redo
 * the references with lfd_triangulate_sampled one at a time (core/strategies.py::SampledLoop._recover does).
 * s_overrides: host f32 [n_refs] or NULL - the normaliser of reference r (> 0: upstream's own torch sum, handed in; else the exact device sum).
 * out->capacity, sel_info, sel_cells: as lfd_triangulate_sampled_multi.  lfd_rng_seed / lfd_rng_set_state first. */
int lfd_triangulate_sampled_chain(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, int32_t M, float cap,
                                  int32_t border, int32_t tiles, const float* s_overrides, const lfd_points* out,
                                  int64_t* ref_offsets, int32_t* seg_counts, int32_t* seg_order, int32_t* sel_info,
                                  int64_t* sel_cells);

/* S: coverage sampling on the device (core/sampling.py:8-53, filter mode).  The context owns a legacy
 * MT19937 stream seeded like np.random.seed(seed); every call consumes it exactly as upstream's
 * np.random.choice does, so successive references see the same stream upstream would.
 * best_cert: device f32 [H*W] (one reference, from lfd_aggregate).  sel_out: device i64 [capacity],
 * ascending cell indices (capacity >= M + tiles*tiles is always enough).  Synchronises; the count and
 * a status (0 ok; 1 NaN, 2 negative, 3 "Fewer non-zero entries in p than size" - the conditions
 * under which upstream's np.random.choice raises ValueError; 4.. internal) are returned on the host.
 * s_override > 0 replaces the normaliser sum(weights) (upstream's is a torch f32 reduction whose
 * rounding depends on the host's thread count; the device uses the correctly rounded exact sum).
 * Limit: the coverage pass holds at most 2304 tiles (tile = max(1, W / tiles) cells per side, ceil(W / tile) * ceil(H / tile) of
 * them): every square grid fits (47 x 47 has the most, 2209; RoMa's grids of 320 ... 1280 cells per side have 576 ... 625); a grid
 * beyond the limit (much taller than wide: 24 x 200) is refused with LFD_ERR_INVALID and a message that says so - the host
 * selection stage has no such limit. */
int lfd_rng_seed(lfd_context* ctx, uint32_t seed);
int lfd_rng_get_state(lfd_context* ctx, uint32_t* key624_host, int32_t* pos_host);
int lfd_rng_set_state(lfd_context* ctx, const uint32_t* key624_host, int32_t pos);
/* The stream put aside and taken back ON THE DEVICE, in stream order, without a host wait: lfd_rng_checkpoint copies the context's MT19937
 * state (key + position) into one of LFD_RNG_CHECKPOINTS places, lfd_rng_rollback copies it back.  A caller that launches fused calls AHEAD
 * of reading their status (lfd_triangulate_sampled_chain: a chain whose bounded spins expire commits nothing, a reference refused for
 * inexactness draws nothing although upstream would have) takes a checkpoint before every call and, on such a status, rolls back to the
 * first affected call's checkpoint and redoes the references one at a time - the stream then continues exactly where upstream's would. */
#define LFD_RNG_CHECKPOINTS 4
int lfd_rng_checkpoint(lfd_context* ctx, int32_t place);
int lfd_rng_rollback(lfd_context* ctx, int32_t place);
int lfd_select_samples(lfd_context* ctx, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                       int32_t border, int32_t tiles, float s_override, int64_t* sel_out, int64_t capacity,
                       int32_t* n_sel_host, int32_t* status_host);

/* no_filter branch of the selection (core/sampling.py:15-21): the min(M, H*W) largest capped
 * certainties in descending order (ties: ascending cell index; NumPy leaves tie order unspecified;
 * NaN last).  M <= 16384.  Does not touch the RNG stream.  Synchronises like lfd_select_samples. */
int lfd_select_top_m(lfd_context* ctx, const float* best_cert, int32_t H, int32_t W, int32_t M, float cap,
                     int64_t* sel_out, int64_t capacity, int32_t* n_sel_host, int32_t* status_host);

/* N1: file payloads on the device (core/writers.py:15-46, core/image_utils.py:24-26).  Colours are
 * quantised like upstream's to_uint8_rgb: clip(round_half_even(c * 255), 0, 255).
 * lfd_pack_ply:      out[n*15] = per point x y z (f32 LE) r g b (u8): the PLY body after upstream's header.
 * lfd_pack_points3d: out[n*43] = per point u64 id (id_base + i + 1), xyz as f64, rgb u8, error f64
 *                    (err may be NULL -> 0.0): upstream's points3D.bin body after the u64 count.
 * out must be 4-byte aligned.  Asynchronous on the context's stream. */
int lfd_pack_ply(lfd_context* ctx, const float* xyz, const float* rgb, int64_t n, uint8_t* out);
int lfd_pack_points3d(lfd_context* ctx, const float* xyz, const float* rgb, const float* err, int64_t n,
                      uint64_t id_base, uint8_t* out);
int lfd_quantise_rgb(lfd_context* ctx, const float* rgb, int64_t n, uint8_t* out);

/* (e) multi-GPU exchange, placement step (no upstream counterpart - upstream has no multi-GPU code): n copies
 * dst[dst_offset .. +nbytes) = src[src_offset .. +nbytes) in ONE launch on `hip_stream` of device `device_index` (offsets and lengths in
 * bytes, no alignment required: 15-byte PLY records).  The overlapped exchange receives every rank's records of a round as one padded block
 * per rank; this puts each reference's records at its place in the ordered cloud (core/distributed.py::OverlappedExchange).  Needs no
 * context; segments must not overlap each other's destination.  Asynchronous. */
typedef struct lfd_copy_segment {
    int64_t src_offset, dst_offset, nbytes;
} lfd_copy_segment;
int lfd_copy_segments(void* hip_stream, int32_t device_index, const void* src, void* dst, const lfd_copy_segment* segs, int32_t n);

/* Debug / test read-back: the f64-widened fundamental matrices the kernels of the LAST prepared batch used, one
 * row-major 3x3 per (reference, slot) pair (n_pairs = n_refs * k of that batch; rows of unused slots are unspecified).
 * Synchronises the stream. */
int lfd_get_pair_fundamental(lfd_context* ctx, int32_t n_pairs, double* F_out_host);

/* ---- N3: image preparation on the device (core/image_utils.py:40-91, core/pipeline.py:163-171) ------------------- */
/* Decoding stays on the host (PIL); the decoded u8 arrays are uploaded and prepared here, bit for bit like Pillow 12:
 * lfd_prepare_image:  dst = Image.resize((w_out, h_out), BILINEAR) of the (h_in, w_in, 3) u8 image src (8-bit two-pass
 *                     fixed-point convolution, horizontal first - vertical first where Pillow's Image.resize does that: an
 *                     image more than 100 times taller than wide that shrinks vertically), then masked pixels black (mask01: device u8 {0,1}
 *                     [h_out*w_out] or NULL) as apply_mask_to_rgb does.  dst: device u8 [h_out*w_out*3].
 * lfd_prepare_mask:   dst01 = load_mask_resized_np after the "L" conversion: Image.resize(NEAREST) of the (h_in, w_in) u8
 *                     mask, then (v / 255 as f32) > threshold, optionally inverted; dst01: device u8 {0,1} [h_out*w_out].
 * Asynchronous on the context's stream (the tables of a new size pair are built on the host and uploaded first). */
int lfd_prepare_image(lfd_context* ctx, const uint8_t* src_rgb, int32_t w_in, int32_t h_in, int32_t w_out, int32_t h_out,
                      const uint8_t* mask01, uint8_t* dst_rgb);
int lfd_prepare_mask(lfd_context* ctx, const uint8_t* src_l, int32_t w_in, int32_t h_in, int32_t w_out, int32_t h_out,
                     float threshold, int32_t invert, uint8_t* dst01);
/* host helpers (CPU tests): the resampling tables exactly as the kernels use them.  bounds: [out_size*2] = {first, count};
 * kk: [out_size * *ksize_out] 22-bit fixed-point coefficients (LFD_ERR_CAPACITY if kk_capacity is too small, *ksize_out is
 * still set); idx: [out_size] NEAREST source indices. */
int lfd_host_resize_tables(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk, int32_t kk_capacity, int32_t* ksize_out);
int lfd_host_nearest_indices(int32_t in_size, int32_t out_size, int32_t* idx);

/* Synchronise the context's stream and report whether the last launches completed normally.
 * *status_out = 0, or 1 when a bounded look-back spin gave up (results invalid; returns LFD_ERR_HIP). */
int lfd_launch_status(lfd_context* ctx, int32_t* status_out);

/* ---- host-side helpers (no GPU needed; used by the CPU test-suite) -------------------------------- */
/* A-grid axis used when axis_x/axis_y are NULL: start + step*j below the midpoint,
 * end - step*(n-1-j) from it on, f32 (the per-element form of torch.linspace, core/matcher.py:132-133). */
int lfd_identity_axis(int32_t n, float* out_host);
/* Largest f32 d with degrees(acosf(d)) >= min_deg: the kernels test `dot <= d` instead of calling
 * acos per cell (core/geometry.py:113-119). */
float lfd_parallax_dot_threshold(float min_deg);
/* F (f32, row-major 9) for one camera pair, the same routine the kernels run in their prologue
 * (core/geometry.py:122-130). */
int lfd_host_fundamental(const float* K1, const float* R1, const float* t1, const float* K2,
                         const float* R2, const float* t2, float* F_out);
/* Smallest right singular vector of a row-major 4x4 f32 matrix, the routine the kernels triangulate with
 * (f64 inverse iteration on A^T A), on the HOST build of the same source; out4 is un-normalised.  Returns the
 * number of solves made (>= 3) or a negative lfd_status.  CPU unit tests compare it with an f64 SVD. */
int lfd_host_null_vector(const float* A16, double* out4);
/* One correspondence through the per-cell routine on the HOST build of the same source (debug /
 * CPU unit tests of the arithmetic; not a fallback: no batch entry point uses it).
 * cam1/cam2: K[9] R[9] t[3] P[12] C[3] w h (as floats, 38 values).  out: x y z r g b err keep. */
int lfd_host_eval_correspondence(const float* cam1, const float* cam2, float xa_norm, float ya_norm,
                                 float xb_norm, float yb_norm, int32_t w_match, int32_t h_match,
                                 const lfd_params* params, float* out8);

/* ---- CPU twin ------------------------------------------------------------------ */
/* The same three steps on the host: EVERY pointer of lfd_batch / lfd_points / the arguments below is a HOST pointer.
 * The per-cell arithmetic is the host build of the very source the kernels compile (csrc/lfd_geometry.hpp; IEEE
 * division / square root where the device uses the 1-ulp v_rcp / v_sqrt), spread over n_threads std::threads
 * (<= 0: all hardware threads).  A host context accepts lfd_upload_cameras, lfd_last_error, lfd_destroy and the
 * three *_host calls; every device entry point refuses it with LFD_ERR_STATE, and the *_host calls refuse a device
 * context: neither side ever stands in for the other.  Semantics (orders, counts, optional outputs, LFD_ERR_CAPACITY
 * with valid counts) are those of lfd_aggregate / lfd_triangulate_dense / lfd_triangulate_indexed. */
int lfd_create_host(int32_t n_threads, lfd_context** out);
int lfd_host_threads(const lfd_context* ctx); /* threads a host context uses (0 for a device context) */
int lfd_aggregate_host(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params, float* best_cert,
                       uint8_t* best_slot);
int lfd_triangulate_dense_host(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params,
                               const lfd_points* out, int64_t* ref_offsets, int32_t* seg_counts);
int lfd_triangulate_indexed_host(lfd_context* ctx, const lfd_batch* batch, const lfd_params* params,
                                 const int64_t* sel_idx, const int64_t* sel_offsets, const lfd_points* out,
                                 int64_t* ref_offsets, int32_t* seg_counts, int32_t* seg_order);

#ifdef __cplusplus
}
#endif
#endif /* LFD_DENSIFY_H */
