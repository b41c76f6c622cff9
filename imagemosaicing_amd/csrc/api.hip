// csrc/api.hip -- the extern "C" surface of libmi355mosaic.so (include/mi355_mosaic.h): argument checks,
// locking, host<->device staging around the kernels of warp.hip / ransac.hip / match.hip / sift.hip.
// There is no CPU compute path in this library: without a usable gfx950 device mi355_create fails.
#include "common.h"
#include <new>

static thread_local std::string g_create_error;   // mi355_last_error(NULL): the calling thread's last creation error

extern "C" void mi355_default_params(mi355_params* p) {
    if (!p) return;
    p->nfeatures = 2000; p->n_octave_layers = 3; p->contrast_threshold = 0.01f; p->edge_threshold = 20.0f; p->sigma = 1.6f;
    p->max_selected = 400; p->select_fraction = 0.3; p->grid_x = 3; p->grid_y = 3; p->min_inliers = 30;
    p->ransac_dist = 2.5f; p->sample_times = 1000; p->pair_window = 182; p->ratio = 0.0f;
}

void mi355_ctx::prof_begin(const char* cls, double alg_bytes, hipStream_t st) {
    ProfClass& pc = prof[cls];
    if (pc.used == pc.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        pc.ev.emplace_back(a, b);
    }
    pc.bytes += alg_bytes;
    (void)hipEventRecord(pc.ev[pc.used].first, st);
}
void mi355_ctx::prof_end(const char* cls, hipStream_t st) {
    ProfClass& pc = prof[cls];
    if (pc.used < pc.ev.size()) { (void)hipEventRecord(pc.ev[pc.used].second, st); pc.used++; }
}

extern "C" int mi355_create(mi355_ctx** out, const mi355_params* params, int device_ordinal) {
    if (!out) return MI355_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { g_create_error = std::string("no HIP device: ") + hipGetErrorString(e); return MI355_ERR_DEVICE; }
    if (device_ordinal < 0 || device_ordinal >= ndev) { g_create_error = "device ordinal out of range"; return MI355_ERR_ARG; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return MI355_ERR_DEVICE; }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return MI355_ERR_DEVICE;
    }
    // the caller's current device is left as it was (every entry point selects ctx->device itself)
    struct DeviceRestore { int prev = -1; ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); } } restore;
    if (hipGetDevice(&restore.prev) != hipSuccess) restore.prev = -1;
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return MI355_ERR_DEVICE; }
    mi355_ctx* c = new (std::nothrow) mi355_ctx();
    if (!c) return MI355_ERR_NOMEM;
    c->device = device_ordinal;
    c->num_cu = prop.multiProcessorCount;
    if (params) c->p = *params; else mi355_default_params(&c->p);
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); delete c; return MI355_ERR_DEVICE; }
    c->stream = c->own_stream;
    if (const char* e = getenv("MI355_BLUR_STREAM")) c->blur_stream = atoi(e) ? 1 : 0;
    if (const char* e = getenv("MI355_CASCADE")) { const int v = atoi(e); c->cascade = v < 0 ? 0 : (v > 3 ? 3 : v); }
    if (getenv("MI355_SERIAL_HEAVY")) c->serial_heavy = 1;
    {
        struct { const char* n; int* p; } knobs[] = {{"MI355_RANSAC_SPLIT", &c->ransac_split}};
        for (auto& k : knobs) if (const char* e = getenv(k.n)) *k.p = atoi(e);
    }
    if (const char* e = getenv("MI355_SIFT_SLOTS")) { const int v = atoi(e); c->sift_nslots = v < 1 ? 1 : (v > 4 ? 4 : v); }
    if (const char* e = getenv("MI355_XSTREAM_MIN_W")) { const int v = atoi(e); c->xstream_min_w = v < 256 ? 256 : v; }
    if (const char* e = getenv("MI355_SIFT_BATCH")) { const int v = atoi(e); c->sift_batch = v < 1 ? 1 : (v > 32 ? 32 : v); }
    *out = c;
    return MI355_OK;
}

extern "C" void mi355_destroy(mi355_ctx* ctx) {
    if (!ctx) return;
    struct DeviceRestore { int prev = -1; ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); } } restore;
    if (hipGetDevice(&restore.prev) != hipSuccess) restore.prev = -1;
    { std::lock_guard<std::mutex> lk(ctx->mu); }       // a call still inside the ctx finishes first (the caller must not start new ones)
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)mi_resolve_features(ctx);
    mi_sift_release(ctx);
    mi_comm_release(ctx);
    mi_surf_release(ctx);
    for (auto& kv : ctx->feats) kv.second.release();
    for (auto& kv : ctx->ws) kv.second.release();
    for (auto& kv : ctx->hws) kv.second.release();
    for (auto& b : ctx->host_frames) b.release();
    for (hipEvent_t e : ctx->host_frame_ev) if (e) (void)hipEventDestroy(e);
    for (auto& t : ctx->draw_tables) { t.buf.release(); t.raw.release(); if (t.ready) (void)hipEventDestroy(t.ready); }
    if (ctx->aux_ev) (void)hipEventDestroy(ctx->aux_ev);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->draw_flags) (void)hipHostFree(ctx->draw_flags);
    for (auto& kv : ctx->prof) for (auto& ev : kv.second.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->heavy_ev) (void)hipEventDestroy(ctx->heavy_ev);
    for (hipEvent_t e : ctx->batch_events) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* mi355_last_error(mi355_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int mi355_set_stream(mi355_ctx* ctx, void* hip_stream) {
    LOCKED_PROLOGUE
    { int rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; }      // parked frames were produced on the old stream
    (void)hipStreamSynchronize(ctx->stream);
    if (hip_stream) {
        // the context's own stream would only occupy a hardware queue from here on
        if (ctx->own_stream) { (void)hipStreamDestroy(ctx->own_stream); ctx->own_stream = nullptr; }
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        if (!ctx->own_stream) MI_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
        ctx->stream = ctx->own_stream;
    }
    return MI355_OK;
}

extern "C" int mi355_synchronize(mi355_ctx* ctx) {
    LOCKED_PROLOGUE
    int rc = mi_resolve_features(ctx);
    if (rc != MI355_OK) return rc;
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

extern "C" void mi355_free(void* p) { free(p); }

// ---- features --------------------------------------------------------------------------------------------------
extern "C" int mi355_sift_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int width_step, int* n_kp) {
    LOCKED_PROLOGUE
    if (!d_bgr || w < 16 || h < 16 || width_step < 3 * w) { ctx->set_error("sift_extract: bad image geometry"); return MI355_ERR_ARG; }
    return mi_sift_extract_dev(ctx, img_id, d_bgr, w, h, width_step, n_kp);
}

extern "C" int mi355_sift_extract(mi355_ctx* ctx, int img_id, const uint8_t* bgr, int w, int h, int width_step,
                                  mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp) {
    int n = 0;
    const bool deferred = !kp && !desc128 && !n_kp;           // nothing asked back: the frame may join a batch
    {
        LOCKED_PROLOGUE
        if (!bgr || w < 16 || h < 16 || width_step < 3 * w) { ctx->set_error("sift_extract: bad image geometry"); return MI355_ERR_ARG; }
        const size_t bytes = (size_t)width_step * h;
        if (deferred) {
            // staging ring in HBM: a slot is reused only after the batch that read it has finished (event of its work area)
            const size_t ring = (size_t)(ctx->sift_nslots + 1) * (size_t)(ctx->sift_batch < 1 ? 1 : ctx->sift_batch);
            if (ctx->host_frames.size() != ring) {
                int rc = mi_resolve_features(ctx);
                if (rc != MI355_OK) return rc;
                for (auto& b : ctx->host_frames) b.release();
                for (hipEvent_t e : ctx->host_frame_ev) if (e) (void)hipEventDestroy(e);
                ctx->host_frames.assign(ring, DevBuf());
                ctx->host_frame_ev.assign(ring, nullptr);
                ctx->host_frame_used.assign(ring, 0);
                ctx->host_frame_next = 0;
            }
            const size_t slot = ctx->host_frame_next;
            ctx->host_frame_next = (ctx->host_frame_next + 1) % ring;
            DevBuf& dimg = ctx->host_frames[slot];
            if (!ctx->host_frame_ev[slot]) MI_HIP(hipEventCreateWithFlags(&ctx->host_frame_ev[slot], hipEventDisableTiming));
            if (ctx->host_frame_used[slot]) {
                // the batch that read this slot was launched at least `batch` frames ago; its event orders the overwrite
                int rc = mi_sift_flush_if_parked(ctx, ctx->host_frame_ev[slot]);
                if (rc != MI355_OK) return rc;
                MI_HIP(hipStreamWaitEvent(ctx->stream, ctx->host_frame_ev[slot], 0));
            }
            ctx->host_frame_used[slot] = 1;
            MI_HIP(dimg.reserve(bytes + 16));
            MI_HIP(hipMemcpyAsync(dimg.p, bgr, bytes, hipMemcpyHostToDevice, ctx->stream));
            ctx->pend_event = ctx->host_frame_ev[slot];      // recorded by the batch this frame ends up in
            const int rc = mi_sift_extract_dev(ctx, img_id, dimg.as<uint8_t>(), w, h, width_step, nullptr);
            ctx->pend_event = nullptr;
            return rc;
        }
        DevBuf& dimg = ctx->buf("sift_host_img");
        MI_HIP(dimg.reserve(bytes + 16));
        MI_HIP(hipMemcpyAsync(dimg.p, bgr, bytes, hipMemcpyHostToDevice, ctx->stream));
        int rc = mi_sift_extract_dev(ctx, img_id, dimg.as<uint8_t>(), w, h, width_step, &n);
        if (rc != MI355_OK) return rc;
    }
    if (n_kp) *n_kp = n;
    if (kp || desc128) return mi355_get_features(ctx, img_id, kp, desc128, max_kp, nullptr);
    return MI355_OK;
}

extern "C" int mi355_set_features(mi355_ctx* ctx, int img_id, const mi355_keypoint* kp, const float* desc128, int n_kp, int w, int h) {
    LOCKED_PROLOGUE
    return mi_set_features(ctx, img_id, kp, desc128, n_kp, w, h);
}

extern "C" int mi355_drop_features(mi355_ctx* ctx, int img_id) {
    LOCKED_PROLOGUE
    (void)mi_resolve_features(ctx);
    MI_HIP(hipStreamSynchronize(ctx->stream));
    if (img_id < 0) { for (auto& kv : ctx->feats) kv.second.release(); ctx->feats.clear(); }
    else { auto it = ctx->feats.find(img_id); if (it != ctx->feats.end()) { it->second.release(); ctx->feats.erase(it); } }
    return MI355_OK;
}

// ---- match ------------------------------------------------------------------------------------------------------
extern "C" int mi355_match_pairs_dev(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed, mi355_pair_result* d_out) {
    LOCKED_PROLOGUE
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_ij || !d_out))) return MI355_ERR_ARG;
    return mi_match_pairs_dev(ctx, pairs_ij, n_pairs, ransac_dist, seed, d_out);
}

extern "C" int mi355_match_pairs(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed, mi355_pair_result* out) {
    LOCKED_PROLOGUE
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_ij || !out))) return MI355_ERR_ARG;
    if (n_pairs == 0) return MI355_OK;
    DevBuf& dres = ctx->buf("pair_results");
    MI_HIP(dres.reserve(sizeof(mi355_pair_result) * (size_t)n_pairs));
    int rc = mi_match_pairs_dev(ctx, pairs_ij, n_pairs, ransac_dist, seed, dres.as<mi355_pair_result>());
    if (rc != MI355_OK) return rc;
    MI_HIP(hipMemcpyAsync(out, dres.p, sizeof(mi355_pair_result) * (size_t)n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

extern "C" int mi355_bf_match(mi355_ctx* ctx, int img_i, int img_j, int sorted, mi355_dmatch* matches, int32_t* d2, int32_t* second_d2,
                              int max_matches, int* n_matches) {
    LOCKED_PROLOGUE
    return mi_bf_match(ctx, img_i, img_j, sorted, matches, d2, second_d2, max_matches, n_matches);
}

extern "C" int mi355_select_grid(mi355_ctx* ctx, const mi355_dmatch* sorted, int n, const float* kp1_xy, int n_kp1,
                                 const float* kp2_xy, int n_kp2, int nMatch, int width, int height, int gridX, int gridY,
                                 mi355_sfpoint* v1, mi355_sfpoint* v2, int* n_out) {
    LOCKED_PROLOGUE
    if (nMatch < 0 || nMatch > MI355_MAX_SELECTED) { ctx->set_error("select_grid: nMatch must be in [0,400]"); return MI355_ERR_ARG; }
    return mi_select_grid(ctx, sorted, n, kp1_xy, n_kp1, kp2_xy, n_kp2, nMatch, width, height, gridX, gridY, v1, v2, n_out);
}

extern "C" int mi355_ransac2d(mi355_ctx* ctx, const mi355_sfpoint* p1, const mi355_sfpoint* p2, int n, float dist, int sample_times,
                              uint32_t seed, mi355_sfpoint* in1, mi355_sfpoint* in2, int* n_in, float H[9]) {
    LOCKED_PROLOGUE
    if (!n_in || !H) return MI355_ERR_ARG;
    *n_in = 0;
    for (int i = 0; i < 9; i++) H[i] = 0.0f;
    if (n <= 0 || !p1 || !p2) return 0;                                // Ransac2D: empty input -> false (mosaicimage.h:1739-1744)
    if (n > MI355_RANSAC_BIG_MAX) { ctx->set_error("ransac2d: at most 65535 correspondences"); return MI355_ERR_ARG; }
    if (n > MI355_MAX_SELECTED) {                                      // beyond the live path's maxNum (MosaicWithoutPos.cpp:5146): the large-n kernel
        int ok = 0;
        const int rc = mi_ransac_big(ctx, p1, p2, n, dist, sample_times, seed, in1, in2, n_in, H, &ok);
        return rc != MI355_OK ? rc : ok;
    }
    DevBuf& d1 = ctx->buf("r1_p1"); DevBuf& d2 = ctx->buf("r1_p2"); DevBuf& dn = ctx->buf("r1_n"); DevBuf& dres = ctx->buf("pair_results");
    MI_HIP(d1.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED)); MI_HIP(d2.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED));
    MI_HIP(dn.reserve(sizeof(int))); MI_HIP(dres.reserve(sizeof(mi355_pair_result)));
    MI_HIP(hipMemcpyAsync(d1.p, p1, sizeof(mi355_sfpoint) * n, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(d2.p, p2, sizeof(mi355_sfpoint) * n, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dn.p, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    int rc = mi_ransac_batch(ctx, d1.as<mi355_sfpoint>(), d2.as<mi355_sfpoint>(), dn.as<int>(), &n, 1, MI355_MAX_SELECTED, dist, sample_times, seed,
                             dres.as<mi355_pair_result>());
    if (rc != MI355_OK) return rc;
    mi355_pair_result* r = (mi355_pair_result*)malloc(sizeof(mi355_pair_result));
    if (!r) return MI355_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(r, dres.p, sizeof(mi355_pair_result), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(r); ctx->set_error(hipGetErrorString(e)); return MI355_ERR_DEVICE; }
    *n_in = r->n_in;
    if (in1) memcpy(in1, r->a, sizeof(mi355_sfpoint) * r->n_in);
    if (in2) memcpy(in2, r->b, sizeof(mi355_sfpoint) * r->n_in);
    memcpy(H, r->H, sizeof(float) * 9);
    const int ok = r->ok;
    free(r);
    return ok;
}

// ---- warps ------------------------------------------------------------------------------------------------------
extern "C" int mi355_warp_image(mi355_ctx* ctx, const uint8_t* src, int w, int h, int ws, int ch, const float h9[9],
                                uint8_t** dst, int* dw, int* dh, int* dws) {
    LOCKED_PROLOGUE
    if (!dst || !dw || !dh || !dws) return MI355_ERR_ARG;
    return mi_warp_image(ctx, src, w, h, ws, ch, h9, dst, dw, dh, dws);
}

extern "C" int mi355_mosaic_refined_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n,
                                        const float* h9s, uint8_t* d_canvas, int cw, int ch, int cws, int row0, int rows) {
    LOCKED_PROLOGUE
    if (!d_imgs || !w || !h || !ws || !h9s || !d_canvas || n <= 0) return MI355_ERR_ARG;
    return mi_mosaic_refined_dev(ctx, d_imgs, w, h, ws, n, h9s, d_canvas, cw, ch, cws, row0, rows);
}

extern "C" int mi355_mosaic_refined(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                                    const float* h9s, uint8_t** canvas, int* cw, int* ch, int* cws) {
    LOCKED_PROLOGUE
    if (!imgs || !w || !h || !ws || !h9s || !canvas || !cw || !ch || !cws) return MI355_ERR_ARG;
    if (n <= 1) { ctx->set_error("mosaic_refined: needs more than one image"); return MI355_ERR_FAILED; }   // MergeImagesRefined convention (:2164-2167)
    int lw, lh, lws;
    int rc = mi355_mosaic_layout(w, h, n, h9s, &lw, &lh, &lws, nullptr);
    if (rc != MI355_OK) { ctx->set_error("mosaic_refined: empty canvas"); return rc; }
    // stage every contributing image in HBM (frames stay resident: 288 GB), then composite in index order
    size_t total = 0;
    std::vector<size_t> off(n, 0);
    for (int k = 0; k < n; k++) { if (h9s[9 * k + 8] == 0.0f) continue; if (!imgs[k] || w[k] < 2 || h[k] < 2 || ws[k] < 3 * w[k]) return MI355_ERR_ARG; off[k] = total; total += ((size_t)ws[k] * h[k] + 255) & ~(size_t)255; }
    DevBuf& dall = ctx->buf("mosaic_srcs");
    DevBuf& dcan = ctx->buf("mosaic_canvas");
    MI_HIP(dall.reserve(total + 16));
    MI_HIP(dcan.reserve((size_t)lws * lh));
    std::vector<const uint8_t*> dptr(n, nullptr);
    for (int k = 0; k < n; k++) {
        if (h9s[9 * k + 8] == 0.0f) continue;
        dptr[k] = dall.as<uint8_t>() + off[k];
        MI_HIP(hipMemcpyAsync((void*)dptr[k], imgs[k], (size_t)ws[k] * h[k], hipMemcpyHostToDevice, ctx->stream));
    }
    rc = mi_mosaic_refined_dev(ctx, dptr.data(), w, h, ws, n, h9s, dcan.as<uint8_t>(), lw, lh, lws, 0, lh);
    if (rc != MI355_OK) return rc;
    uint8_t* out = (uint8_t*)malloc((size_t)lws * lh);
    if (!out) return MI355_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(out, dcan.p, (size_t)lws * lh, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(out); ctx->set_error(hipGetErrorString(e)); return MI355_ERR_DEVICE; }
    *canvas = out; *cw = lw; *ch = lh; *cws = lws;
    return MI355_OK;
}

extern "C" int mi355_chips_and_masks(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                                     const float* h9s, const uint8_t* keep, int find_masks,
                                     int* n_chips, mi355_chip_info** chips, uint8_t*** chip_imgs, uint8_t*** masks, int* canvas_w, int* canvas_h) {
    LOCKED_PROLOGUE
    return mi_chips_and_masks(ctx, imgs, w, h, ws, n, h9s, keep, find_masks, n_chips, chips, chip_imgs, masks, canvas_w, canvas_h);
}

extern "C" int mi355_multiband_blend(mi355_ctx* ctx, const uint8_t* const* chips, const uint8_t* const* masks, const mi355_chip_info* info, int n,
                                     int canvas_w, int canvas_h, int band, uint8_t** out, int* out_w, int* out_h, int* out_ws) {
    LOCKED_PROLOGUE
    return mi_multiband_blend(ctx, chips, masks, info, n, canvas_w, canvas_h, band, out, out_w, out_h, out_ws);
}

extern "C" int mi355_mosaic_blended(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                                    const uint8_t* keep, int band, uint8_t** out, int* out_w, int* out_h, int* out_ws) {
    LOCKED_PROLOGUE
    return mi_mosaic_blended(ctx, imgs, w, h, ws, n, h9s, keep, band, out, out_w, out_h, out_ws);
}

extern "C" int mi355_mosaic_blended_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                                        const uint8_t* keep, int band, uint8_t* d_canvas, int cw, int ch, int cws) {
    LOCKED_PROLOGUE
    return mi_mosaic_blended_dev(ctx, d_imgs, w, h, ws, n, h9s, keep, band, d_canvas, cw, ch, cws);
}

extern "C" int mi355_mosaic_blended_rows_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                                             const uint8_t* keep, int band, uint8_t* d_rows, int cw, int ch, int cws, int row0, int rows) {
    LOCKED_PROLOGUE
    if (rows < 1) { ctx->set_error("mosaic_blended_rows_dev: bad stripe"); return MI355_ERR_ARG; }
    return mi_mosaic_blended_dev(ctx, d_imgs, w, h, ws, n, h9s, keep, band, d_rows, cw, ch, cws, row0, rows);
}

// The frames a stripe call reads: the stripe calls themselves with the device work left out (cover_only), so the list cannot drift from them.
extern "C" int mi355_mosaic_stripe_cover(mi355_ctx* ctx, int mode, const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int band,
                                         int row0, int rows, uint8_t* need) {
    LOCKED_PROLOGUE
    if (!w || !h || !h9s || !need || n <= 0 || mode < 0 || mode > 2) { ctx->set_error("mosaic_stripe_cover: bad arguments"); return MI355_ERR_ARG; }
    memset(need, 0, (size_t)n);
    std::vector<const uint8_t*> none((size_t)n, nullptr);
    std::vector<int> ws((size_t)n);
    for (int k = 0; k < n; k++) ws[k] = (3 * w[k] + 3) & ~3;
    if (mode != MI355_COVER_BLENDED) {
        int cw = 0, ch = 0, cws = 0;
        const int rc = mi355_mosaic_layout(w, h, n, h9s, &cw, &ch, &cws, nullptr);
        if (rc != MI355_OK) { ctx->set_error("mosaic_stripe_cover: no image with h[8] != 0 / empty canvas"); return rc; }
        return mi_mosaic_refined_dev(ctx, none.data(), w, h, ws.data(), n, h9s, nullptr, cw, ch, cws, row0, rows, need, mode == MI355_COVER_REFINED_EXACT ? 1 : 0);
    }
    int cw = 0, ch = 0;
    { const int rc = mi_blend_layout(w, h, n, h9s, keep, &cw, &ch); if (rc != MI355_OK) return rc; }
    return mi_mosaic_blended_dev(ctx, none.data(), w, h, ws.data(), n, h9s, keep, band, nullptr, cw, ch, (cw * 3 + 3) & ~3, row0, rows, need);
}

extern "C" int mi355_blend_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int* cw, int* ch, int* cws) {
    if (!cw || !ch) return MI355_ERR_ARG;
    const int rc = mi_blend_layout(w, h, n, h9s, keep, cw, ch);
    if (rc == MI355_OK && cws) *cws = (*cw * 3 + 3) & ~3;
    return rc;
}

// ---- measurement hooks --------------------------------------------------------------------------------------------
extern "C" int mi355_profile_enable(mi355_ctx* ctx, int on) {
    LOCKED_PROLOGUE
    ctx->profiling = on != 0;
    return MI355_OK;
}
extern "C" int mi355_last_sift_counters(mi355_ctx* ctx, int32_t out8[8]) {
    LOCKED_PROLOGUE
    if (!out8) return MI355_ERR_ARG;
    for (int i = 0; i < 8; i++) out8[i] = ctx->last_counts[i];
    return MI355_OK;
}
extern "C" int mi355_set_option(mi355_ctx* ctx, const char* name, int value) {
    LOCKED_PROLOGUE
    if (!name) return MI355_ERR_ARG;
    if (std::string(name) == "sift_slots") {
        int rc = mi_resolve_features(ctx);
        if (rc != MI355_OK) return rc;
        ctx->sift_nslots = value < 1 ? 1 : (value > 4 ? 4 : value);
        return MI355_OK;
    }
    if (std::string(name) == "sift_batch") {
        int rc = mi_resolve_features(ctx);
        if (rc != MI355_OK) return rc;
        ctx->sift_batch = value < 1 ? 1 : (value > 32 ? 32 : value);
        return MI355_OK;
    }
    {
        struct { const char* n; int* p; } knobs[] = {{"ransac_split", &ctx->ransac_split}, {"strict_frames", &ctx->strict_frames}};
        for (auto& k : knobs) if (std::string(name) == k.n) { *k.p = value; return MI355_OK; }
    }
    if (std::string(name) == "sift_flush") return mi_sift_flush(ctx);      // close the batch that is collecting frames now (no wait): the caller shapes the batches of a short survey
    if (std::string(name) == "blur_stream") { ctx->blur_stream = value ? 1 : 0; return MI355_OK; }
    if (std::string(name) == "sift_cascade") { ctx->cascade = value < 0 ? 0 : (value > 3 ? 3 : value); return MI355_OK; }
    if (std::string(name) == "serial_heavy") {
        int rc = mi_resolve_features(ctx);
        if (rc != MI355_OK) return rc;
        ctx->serial_heavy = value ? 1 : 0; ctx->heavy_ev_valid = false;
        return MI355_OK;
    }
    if (std::string(name).rfind("profile_every:", 0) == 0) { ctx->prof[std::string(name).substr(14)].every = value < 1 ? 1 : value; return MI355_OK; }
    if (std::string(name) == "xstream_min_w") { ctx->xstream_min_w = value < 256 ? 256 : value; return MI355_OK; }
    if (std::string(name) == "xstream_min_frames") { ctx->xstream_min_frames = value < 1 ? 1 : value; return MI355_OK; }
    ctx->set_error(std::string("set_option: unknown option ") + name);
    return MI355_ERR_ARG;
}
extern "C" int mi355_profile_only(mi355_ctx* ctx, const char* kernel_class) {
    LOCKED_PROLOGUE
    ctx->prof_only = kernel_class ? kernel_class : "";
    return MI355_OK;
}
extern "C" int mi355_profile_reset(mi355_ctx* ctx) {
    LOCKED_PROLOGUE
    (void)mi_resolve_features(ctx);
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->prof) { kv.second.used = 0; kv.second.bytes = 0.0; kv.second.seen = 0; }
    return MI355_OK;
}
extern "C" int mi355_profile_get(mi355_ctx* ctx, const char* kernel_class, double* total_ms, int64_t* launches, double* alg_bytes) {
    LOCKED_PROLOGUE
    if (!kernel_class) return MI355_ERR_ARG;
    (void)mi_resolve_features(ctx);
    MI_HIP(hipStreamSynchronize(ctx->stream));
    double ms = 0.0; int64_t cnt = 0; double bytes = 0.0;
    auto it = ctx->prof.find(kernel_class);
    if (it != ctx->prof.end()) {
        for (size_t i = 0; i < it->second.used; i++) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, it->second.ev[i].first, it->second.ev[i].second) == hipSuccess) ms += t;
        }
        cnt = (int64_t)it->second.used; bytes = it->second.bytes;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    if (alg_bytes) *alg_bytes = bytes;
    return MI355_OK;
}

// ---- multi-GPU pair schedule ----------------------------------------------------------------------------------------
extern "C" int mi355_pair_schedule(int n_images, int window, int rank, int world, int32_t* pairs_ij, int max_pairs, int* n_pairs) {
    if (n_images < 0 || window < 2 || world < 1 || rank < 0 || rank >= world || !n_pairs) return MI355_ERR_ARG;
    int cnt = 0;
    for (int i = rank; i < n_images; i += world) {                       // MosaicWithoutPos.cpp:5066 (thread k takes i = k mod T)
        const int jEnd = n_images < i + window ? n_images : i + window;   // :5083
        for (int j = i + 1; j < jEnd; j++) {
            if (pairs_ij && cnt < max_pairs) { pairs_ij[2 * cnt] = i; pairs_ij[2 * cnt + 1] = j; }
            cnt++;
        }
    }
    *n_pairs = cnt;
    return (pairs_ij && cnt > max_pairs) ? MI355_ERR_ARG : MI355_OK;
}
