/*
 * caller.c -- the C-ABI of include/lfd_densify.h called from plain C (C99, no torch, no Python).
 *
 *     gcc -std=c99 -Wall -Werror -I include tests/abi/caller.c -L <pkg> -llfd_densify -lamdhip64 -lm -o caller
 *     caller <case.bin> host|device
 *
 * "the drop-in boundary is a C-ABI shared library (extern "C", plain pointers and sizes)".  This program is the proof that the
 * header alone is enough to drive it: it reads one upstream-made case (tests/test_abi_from_c.py dumps golden g3's `_triangulate_ref` case -
 * upstream core/pipeline.py:602-780 on captured selections - into a flat file), runs
 *     lfd_create_host / lfd_create  ->  lfd_upload_cameras  ->  lfd_triangulate_indexed_host / lfd_triangulate_indexed
 * and compares counts, segment sizes and positions with upstream's result; then hands the library a bad argument and expects a status and a
 * message instead of a crash.  Exit code 0 and a line "OK ..." on success.
 *
 * File layout (little endian): int32 header[16] = {magic 0x4C464443, H, W, w_match, h_match, k, warp_channels, n_cams, ref_cam, n_sel, n_expected,
 * no_filter, 0...}; double sampson; float certainty, cap, reproj, parallax; int32 nbr_cam[k]; float K[n][9] R[n][9] t[n][3] P[n][12] C[n][3];
 * int32 wh[n][2]; float cert[k][H*W]; float warp[k][H*W*ch]; uint8 image[h_match*w_match*3]; int64 sel[n_sel]; int32 seg_count[k];
 * float xyz[n_expected][3].
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lfd_densify.h"

/* the four HIP runtime calls the device leg needs, declared by hand: this file stays C99 and includes nothing of ROCm */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

static void* xread(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
        fprintf(stderr, "short read (%zu bytes)\n", bytes);
        exit(3);
    }
    return p;
}

static void* to_device(const void* host, size_t bytes, int on_device) {
    void* d = NULL;
    if (!on_device) return (void*)host;
    if (hipMalloc(&d, bytes ? bytes : 4) != 0 || (bytes && hipMemcpy(d, host, bytes, 1) != 0)) {
        fprintf(stderr, "hipMalloc / hipMemcpy failed\n");
        exit(4);
    }
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s case.bin host|device\n", argv[0]);
        return 2;
    }
    const int on_device = strcmp(argv[2], "device") == 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    int32_t* hd = (int32_t*)xread(f, 16 * sizeof(int32_t));
    if (hd[0] != 0x4C464443) {
        fprintf(stderr, "bad magic\n");
        return 2;
    }
    const int32_t H = hd[1], W = hd[2], wm = hd[3], hm = hd[4], k = hd[5], ch = hd[6], n_cams = hd[7], ref_cam = hd[8], n_sel = hd[9], n_exp = hd[10];
    lfd_params params;
    memset(&params, 0, sizeof params);
    double* sampson = (double*)xread(f, sizeof(double));
    float* thr = (float*)xread(f, 4 * sizeof(float));
    params.sampson_thresh = *sampson;
    params.certainty_thresh = thr[0];
    params.sample_cap = thr[1];
    params.reproj_thresh = thr[2];
    params.min_parallax_deg = thr[3];
    params.no_filter = hd[11];
    int32_t* nbr_cam = (int32_t*)xread(f, (size_t)k * sizeof(int32_t));
    float* K = (float*)xread(f, (size_t)n_cams * 9 * sizeof(float));
    float* R = (float*)xread(f, (size_t)n_cams * 9 * sizeof(float));
    float* t = (float*)xread(f, (size_t)n_cams * 3 * sizeof(float));
    float* P = (float*)xread(f, (size_t)n_cams * 12 * sizeof(float));
    float* C = (float*)xread(f, (size_t)n_cams * 3 * sizeof(float));
    int32_t* wh = (int32_t*)xread(f, (size_t)n_cams * 2 * sizeof(int32_t));
    const size_t cells = (size_t)H * W;
    float* cert = (float*)xread(f, (size_t)k * cells * sizeof(float));
    float* warp = (float*)xread(f, (size_t)k * cells * ch * sizeof(float));
    uint8_t* image = (uint8_t*)xread(f, (size_t)hm * wm * 3);
    int64_t* sel = (int64_t*)xread(f, (size_t)n_sel * sizeof(int64_t));
    int32_t* seg_expected = (int32_t*)xread(f, (size_t)k * sizeof(int32_t));
    float* xyz_expected = (float*)xread(f, (size_t)n_exp * 3 * sizeof(float));
    fclose(f);

    lfd_context* ctx = NULL;
    int rc = on_device ? lfd_create(0, NULL, &ctx) : lfd_create_host(1, &ctx);
    if (rc != LFD_OK) {
        fprintf(stderr, "create failed (%d): %s\n", rc, lfd_last_error(NULL));
        return 5;
    }
    if (lfd_abi_version() != LFD_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header ABI %d\n", lfd_abi_version(), LFD_ABI_VERSION);
        return 5;
    }
    rc = lfd_upload_cameras(ctx, n_cams, K, R, t, P, C, wh);
    if (rc != LFD_OK) {
        fprintf(stderr, "lfd_upload_cameras failed (%d): %s\n", rc, lfd_last_error(ctx));
        return 5;
    }

    /* one reference view, k neighbour slots: the per-slot arrays are HOST arrays of (device) pointers */
    const float** cert_ptrs = (const float**)malloc((size_t)k * sizeof(float*));
    const float** warp_ptrs = (const float**)malloc((size_t)k * sizeof(float*));
    for (int j = 0; j < k; ++j) {
        cert_ptrs[j] = (const float*)to_device(cert + (size_t)j * cells, cells * sizeof(float), on_device);
        warp_ptrs[j] = (const float*)to_device(warp + (size_t)j * cells * ch, cells * ch * sizeof(float), on_device);
    }
    const uint8_t* image_ptr = (const uint8_t*)to_device(image, (size_t)hm * wm * 3, on_device);
    const int64_t* sel_dev = (const int64_t*)to_device(sel, (size_t)n_sel * sizeof(int64_t), on_device);
    lfd_batch batch;
    memset(&batch, 0, sizeof batch);
    batch.n_refs = 1;
    batch.k = k;
    batch.H = H;
    batch.W = W;
    batch.w_match = wm;
    batch.h_match = hm;
    batch.warp_channels = ch;
    batch.ref_cam = &ref_cam;
    batch.n_slots = &k;
    batch.nbr_cam = nbr_cam;
    batch.cert = cert_ptrs;
    batch.warp = warp_ptrs;
    batch.image = &image_ptr;

    const int64_t cap = n_sel;
    lfd_points out;
    memset(&out, 0, sizeof out);
    float* xyz_host = (float*)calloc((size_t)cap * 3 + 1, sizeof(float));
    float* rgb_host = (float*)calloc((size_t)cap * 3 + 1, sizeof(float));
    float* err_host = (float*)calloc((size_t)cap + 1, sizeof(float));
    out.xyz = (float*)to_device(xyz_host, (size_t)cap * 3 * sizeof(float), on_device);
    out.rgb = (float*)to_device(rgb_host, (size_t)cap * 3 * sizeof(float), on_device);
    out.err = (float*)to_device(err_host, (size_t)cap * sizeof(float), on_device);
    out.capacity = cap;
    int64_t offs_host[2] = {0, 0};
    int32_t* seg_host = (int32_t*)calloc((size_t)k, sizeof(int32_t));
    int32_t* order_host = (int32_t*)calloc((size_t)k, sizeof(int32_t));
    int64_t* offs = (int64_t*)to_device(offs_host, sizeof offs_host, on_device);
    int32_t* seg = (int32_t*)to_device(seg_host, (size_t)k * sizeof(int32_t), on_device);
    int32_t* order = (int32_t*)to_device(order_host, (size_t)k * sizeof(int32_t), on_device);
    const int64_t sel_offsets[2] = {0, n_sel};
    rc = on_device ? lfd_triangulate_indexed(ctx, &batch, &params, sel_dev, sel_offsets, &out, offs, seg, order)
                   : lfd_triangulate_indexed_host(ctx, &batch, &params, sel_dev, sel_offsets, &out, offs, seg, order);
    if (rc != LFD_OK) {
        fprintf(stderr, "lfd_triangulate_indexed%s failed (%d): %s\n", on_device ? "" : "_host", rc, lfd_last_error(ctx));
        return 6;
    }
    if (on_device) {
        int32_t st = 0;
        if (lfd_launch_status(ctx, &st) != LFD_OK || st != 0) {
            fprintf(stderr, "launch status %d: %s\n", st, lfd_last_error(ctx));
            return 6;
        }
        hipDeviceSynchronize();
        hipMemcpy(offs_host, offs, sizeof offs_host, 2);
        hipMemcpy(seg_host, seg, (size_t)k * sizeof(int32_t), 2);
        hipMemcpy(order_host, order, (size_t)k * sizeof(int32_t), 2);
        hipMemcpy(xyz_host, out.xyz, (size_t)cap * 3 * sizeof(float), 2);
    } else {
        memcpy(xyz_host, out.xyz, (size_t)cap * 3 * sizeof(float));
    }

    /* upstream's result: the same number of survivors, the same group sizes (in upstream's group order), positions within the stated tolerance
     * (xyz rel 1e-5: the library's f64 null vector against LAPACK's f32 SVD) */
    if (offs_host[1] != n_exp) {
        fprintf(stderr, "survivors: %lld, upstream %d\n", (long long)offs_host[1], n_exp);
        return 7;
    }
    for (int g = 0; g < k; ++g) {
        const int slot = order_host[g];
        const int32_t have = slot >= 0 ? seg_host[slot] : 0;
        if (have != seg_expected[g]) {
            fprintf(stderr, "group %d (slot %d): %d survivors, upstream %d\n", g, slot, have, seg_expected[g]);
            return 7;
        }
    }
    double worst = 0.0;
    for (int64_t i = 0; i < (int64_t)n_exp * 3; ++i) {
        const double scale = fabs((double)xyz_expected[i]) > 1.0 ? fabs((double)xyz_expected[i]) : 1.0;
        const double d = fabs((double)xyz_host[i] - (double)xyz_expected[i]) / scale;
        if (!(d <= worst)) worst = d;
    }
    if (!(worst <= 1e-5)) {
        fprintf(stderr, "positions differ from upstream's by %.3g (relative)\n", worst);
        return 7;
    }

    /* a bad argument is a status and a message, never a crash */
    lfd_batch bad = batch;
    bad.k = LFD_MAX_SLOTS + 1;
    rc = on_device ? lfd_triangulate_indexed(ctx, &bad, &params, sel_dev, sel_offsets, &out, offs, seg, order)
                   : lfd_triangulate_indexed_host(ctx, &bad, &params, sel_dev, sel_offsets, &out, offs, seg, order);
    char msg[256];
    strncpy(msg, lfd_last_error(ctx) ? lfd_last_error(ctx) : "", sizeof msg - 1);
    msg[sizeof msg - 1] = '\0';
    if (rc == LFD_OK || msg[0] == '\0') {
        fprintf(stderr, "a batch with %d slots was accepted (rc %d)\n", bad.k, rc);
        return 8;
    }
    /* the other side's entry point refuses this kind of context */
    rc = on_device ? lfd_triangulate_indexed_host(ctx, &batch, &params, sel_dev, sel_offsets, &out, offs, seg, order)
                   : lfd_triangulate_indexed(ctx, &batch, &params, sel_dev, sel_offsets, &out, offs, seg, order);
    if (rc == LFD_OK) {
        fprintf(stderr, "a %s context was accepted by the other side's entry point\n", on_device ? "device" : "host");
        return 8;
    }
    /* the device leg also drives upstream's whole per-reference stage (sampling on the context's MT19937 stream included) from C: one call per
     * reference twice in a row, then the same two references in ONE call (lfd_triangulate_sampled_chain) on a re-seeded stream - the same counts,
     * reference by reference, and the same first position */
    long long chain_counts[2] = {-1, -1};
    if (on_device) {
        const int32_t M = 200, tiles = 24;
        const int64_t cap2 = 2 * ((int64_t)M + tiles * tiles + 64);
        lfd_points o2;
        memset(&o2, 0, sizeof o2);
        o2.xyz = (float*)to_device(calloc((size_t)cap2 * 3, sizeof(float)), (size_t)cap2 * 3 * sizeof(float), 1);
        o2.rgb = (float*)to_device(calloc((size_t)cap2 * 3, sizeof(float)), (size_t)cap2 * 3 * sizeof(float), 1);
        o2.err = (float*)to_device(calloc((size_t)cap2, sizeof(float)), (size_t)cap2 * sizeof(float), 1);
        o2.capacity = cap2;
        int64_t offs2_host[3] = {0, 0, 0};
        int32_t info_host[5] = {0, 0, 0, 0, 0};
        int64_t* offs2 = (int64_t*)to_device(offs2_host, sizeof offs2_host, 1);
        int32_t* seg2 = (int32_t*)to_device(calloc((size_t)2 * k, sizeof(int32_t)), (size_t)2 * k * sizeof(int32_t), 1);
        int32_t* order2 = (int32_t*)to_device(calloc((size_t)2 * k, sizeof(int32_t)), (size_t)2 * k * sizeof(int32_t), 1);
        int32_t* info = (int32_t*)to_device(info_host, sizeof info_host, 1);
        long long single_counts[2];
        float first_single[3], first_chain[3];
        if (lfd_rng_seed(ctx, 5u) != LFD_OK) return 9;
        for (int r = 0; r < 2; ++r) {
            rc = lfd_triangulate_sampled(ctx, &batch, &params, M, 0.9f, 2, tiles, 0.0f, &o2, offs2, seg2, order2, info, NULL);
            if (rc != LFD_OK) { fprintf(stderr, "lfd_triangulate_sampled failed (%d): %s\n", rc, lfd_last_error(ctx)); return 9; }
            hipDeviceSynchronize();
            hipMemcpy(offs2_host, offs2, 2 * sizeof(int64_t), 2);
            hipMemcpy(info_host, info, 3 * sizeof(int32_t), 2);
            if (info_host[1] != 0 || info_host[2] != 0) { fprintf(stderr, "sampled call %d: selection status %d, launch status %d\n", r, info_host[1], info_host[2]); return 9; }
            single_counts[r] = (long long)offs2_host[1];
            if (r == 0) hipMemcpy(first_single, o2.xyz, sizeof first_single, 2);
        }
        /* the same reference twice in one batch */
        const int32_t ref2[2] = {ref_cam, ref_cam}, slots2[2] = {k, k};
        int32_t* nbr2 = (int32_t*)malloc((size_t)2 * k * sizeof(int32_t));
        const float** cert2 = (const float**)malloc((size_t)2 * k * sizeof(float*));
        const float** warp2 = (const float**)malloc((size_t)2 * k * sizeof(float*));
        const uint8_t* image2[2] = {image_ptr, image_ptr};
        for (int j = 0; j < 2 * k; ++j) { nbr2[j] = nbr_cam[j % k]; cert2[j] = cert_ptrs[j % k]; warp2[j] = warp_ptrs[j % k]; }
        lfd_batch b2 = batch;
        b2.n_refs = 2; b2.ref_cam = ref2; b2.n_slots = slots2; b2.nbr_cam = nbr2; b2.cert = cert2; b2.warp = warp2; b2.image = image2;
        if (lfd_rng_seed(ctx, 5u) != LFD_OK) return 9;
        rc = lfd_triangulate_sampled_chain(ctx, &b2, &params, M, 0.9f, 2, tiles, NULL, &o2, offs2, seg2, order2, info, NULL);
        if (rc != LFD_OK) { fprintf(stderr, "lfd_triangulate_sampled_chain failed (%d): %s\n", rc, lfd_last_error(ctx)); return 9; }
        hipDeviceSynchronize();
        hipMemcpy(offs2_host, offs2, 3 * sizeof(int64_t), 2);
        hipMemcpy(info_host, info, 5 * sizeof(int32_t), 2);
        hipMemcpy(first_chain, o2.xyz, sizeof first_chain, 2);
        chain_counts[0] = (long long)(offs2_host[1] - offs2_host[0]);
        chain_counts[1] = (long long)(offs2_host[2] - offs2_host[1]);
        if (info_host[1] != 0 || info_host[3] != 0 || info_host[4] != 0 || chain_counts[0] != single_counts[0] || chain_counts[1] != single_counts[1] ||
            chain_counts[0] <= 0 || memcmp(first_single, first_chain, sizeof first_single) != 0) {
            fprintf(stderr, "chained call: %lld + %lld survivors (status %d %d, launch %d), one call per reference: %lld + %lld\n", chain_counts[0], chain_counts[1],
                    info_host[1], info_host[3], info_host[4], single_counts[0], single_counts[1]);
            return 9;
        }
        /* the stream put aside before a call and taken back after it: the second reference alone, from the checkpoint between the two, is the chain's second */
        if (lfd_rng_seed(ctx, 5u) != LFD_OK) return 9;
        rc = lfd_triangulate_sampled(ctx, &batch, &params, M, 0.9f, 2, tiles, 0.0f, &o2, offs2, seg2, order2, info, NULL);
        if (rc != LFD_OK || lfd_rng_checkpoint(ctx, 2) != LFD_OK) { fprintf(stderr, "checkpoint: %s\n", lfd_last_error(ctx)); return 9; }
        for (int again = 0; again < 2; ++again) {
            rc = lfd_triangulate_sampled(ctx, &batch, &params, M, 0.9f, 2, tiles, 0.0f, &o2, offs2, seg2, order2, info, NULL);
            if (rc != LFD_OK) return 9;
            hipDeviceSynchronize();
            hipMemcpy(offs2_host, offs2, 2 * sizeof(int64_t), 2);
            if ((long long)offs2_host[1] != single_counts[1]) { fprintf(stderr, "after a rollback: %lld survivors, expected %lld\n", (long long)offs2_host[1], single_counts[1]); return 9; }
            if (lfd_rng_rollback(ctx, 2) != LFD_OK) { fprintf(stderr, "rollback: %s\n", lfd_last_error(ctx)); return 9; }
        }
        if (lfd_rng_rollback(ctx, 1) == LFD_OK || lfd_rng_checkpoint(ctx, LFD_RNG_CHECKPOINTS) == LFD_OK) { fprintf(stderr, "a place without a checkpoint / out of range was accepted\n"); return 9; }
    }
    printf("OK %s: %lld survivors in upstream's groups, positions within %.2g of upstream's; bad argument -> status %d \"%s\"\n", argv[2],
           (long long)offs_host[1], worst, LFD_ERR_INVALID, msg);
    if (on_device) printf("OK device: two references chained on one MT19937 stream from C: %lld + %lld survivors, as one call per reference\n", chain_counts[0], chain_counts[1]);
    lfd_destroy(ctx);
    return 0;
}
