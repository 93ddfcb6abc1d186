// The context behind the opaque `lfd_context` of include/lfd_densify.h, shared by the device entry points (lfd_api.hip)
// and the CPU twin (lfd_host.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/lfd_densify.h"
#include "lfd_device.hpp"

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

struct LfdHostPool;                              // lfd_host.hip: the host context's parked worker threads
void lfd_host_pool_destroy(LfdHostPool* p);

// One of the two places a batch's tables live on the device: descriptor tables (refs | slots | selection offsets | fundamental matrices)
// and the per-pair constants derived from them.  Two slots let lfd_prepare_batch stage batch i+1 - on the context's own preparation
// stream - while the kernels of batch i still read theirs.
struct LfdBatchSlot {
    DeviceBuffer desc, consts;
    std::vector<unsigned char> cache;      // what `desc` currently holds (or will hold once the stream has caught up)
    bool consts_valid = false;
    int wm = 0, hm = 0, refs = 0, k = 0;   // what the constants were derived for
    void* pinned = nullptr;                // host staging of the table upload
    size_t pinned_bytes = 0;
    hipEvent_t pinned_free = nullptr;      // behind the upload that last read `pinned`
    bool pinned_in_flight = false;
    void* pinned_x = nullptr;              // host staging of the selection offsets (lfd_triangulate_indexed), uploaded apart from the tables
    size_t pinned_x_bytes = 0;
    hipEvent_t x_free = nullptr;           // behind the upload that last read `pinned_x`
    bool x_in_flight = false;
    hipEvent_t ready = nullptr;            // behind upload + setup issued on the preparation stream
    bool ready_pending = false;            // ... which the launch stream has not been told to wait for yet
    hipEvent_t idle = nullptr;             // on the launch stream, where the launch AFTER this slot's last user begins ...
    hipEvent_t idle_ext = nullptr;         // ... or, when that user was a dense launch, the stop event the kernel itself carried (no packet of
    bool idle_attached = false;            //     its own in the stream): `idle`, or an event of lfd_kernel_timing's ring (not owned)
    bool used = false;
};

struct LfdEnvSwitches {          // profiling / A-B switches of the environment, read once by lfd_create
    size_t dense_extra_lds = 0;           // LFD_DENSE_EXTRA_LDS: dynamic LDS per dense workgroup (lowers the resident workgroups per CU)
    std::string dense_timing_path;        // LFD_DENSE_TIMING (profiling builds): file the per-tile phase stamps are dumped to
    bool indexed_split = true;            // LFD_INDEXED_SPLIT=0: indexed mode in one kernel
    int select_timing = 0;                // LFD_SELECT_TIMING: 1 phase stamps of the selection kernel, 2 of the multi-workgroup one
    int select_workgroups = -1;           // LFD_SELECT_WORKGROUPS: compute workgroups of the selection (-1: default)
};

struct lfd_context {
    LfdEnvSwitches env;
    // a context made by lfd_create_host() never touches HIP: it serves the *_host entry points only
    bool is_host = false;
    int host_threads = 1;
    LfdHostPool* host_pool = nullptr;            // created with the context, joined by lfd_destroy
    std::vector<unsigned char> host_stage;       // survivors of the chunks in flight (lfd_triangulate_dense_host), kept across calls
    std::vector<LfdCam> host_cams;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // camera table
    DeviceBuffer cams;
    int32_t n_cams = 0;
    // descriptor tables + per-pair constants of the batches in flight (see LfdBatchSlot)
    LfdBatchSlot slot[2];
    int cur = 0;                   // slot of the last launch
    int last_used = -1;            // ... -1 before the first one
    hipStream_t prep_stream = nullptr;   // lfd_prepare_batch's uploads and setup kernels (created on first use)
    int* pinned_words = nullptr;   // 16 pinned ints: landing place of the small synchronous read-backs (status, selection count)
    // look-back workspace: [0] u64 ticket counter, [1..] tile states
    DeviceBuffer ws;
    unsigned long long tickets_issued = 0;   // host mirror of the device ticket counter (indexed kernel)
    unsigned long long lane_issued[LFD_TICKET_LANES] = {};   // host mirrors of the dense kernel's ticket sequences
    unsigned epoch = 0;
    int n_cus = 0;                 // compute units of the device
    // default A-grid axes
    DeviceBuffer axes;
    int axes_w = 0, axes_h = 0;
    // dense mode's colour tables of the analytic A-grid (one entry per grid column / row), for the last grid + match size
    DeviceBuffer colour_tab;
    int colour_key[4] = {0, 0, 0, 0};
    // indexed-mode scratch
    DeviceBuffer scratch, codes, idx_tab, agg, sel_buf;
    // selection stage: legacy MT19937 stream (625 words) + scratch
    DeviceBuffer mt, sel_scratch, mt_batch;      // mt_batch: per-reference MT19937 states of lfd_triangulate_sampled_multi
    DeviceBuffer mt_ckpt;                        // lfd_rng_checkpoint: LFD_RNG_CHECKPOINTS copies of `mt` (640 words apart)
    unsigned mt_ckpt_taken = 0;                  // ... which of them hold a state
    DeviceBuffer sel_chain;                      // lfd_triangulate_sampled_chain: chain block, the stream's ring of doubles, the keys of its twists
    DeviceBuffer stamps;           // profiling builds: phase stamps of the dense kernel
    DeviceBuffer seg_scan;         // tile segments: exclusive prefix of the last table handed to lfd_order_segments / lfd_pack_*_segments
    // N3 image preparation: coefficient / index tables of the last size pair
    DeviceBuffer img_tab, msk_tab;
    int img_key[4] = {0, 0, 0, 0}, img_ks[2] = {0, 0};
    int msk_key[4] = {0, 0, 0, 0}, msk_inv = 0;
    float msk_thr = -1.0f;
    // lfd_kernel_timing: start / stop events handed to the dense kernel's launches (hipExtLaunchKernelGGL), a ring of `kt_start.size()`
    std::vector<hipEvent_t> kt_start, kt_stop;
    size_t kt_used = 0;            // launches timed since the last read
    bool mt_seeded = false;
    bool topm_lds_attr_set = false;   // hipFuncSetAttribute is per device: remembered per context, not per process
};

// records `msg` on the context (or as the creation error when ctx is null) and returns `code`
int lfd_fail(lfd_context* ctx, int code, const std::string& msg);
