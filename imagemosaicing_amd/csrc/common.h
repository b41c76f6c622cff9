// csrc/common.h -- internal declarations shared by the HIP translation units of libmi355mosaic.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/mi355_mosaic.h"

#define MI_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->set_error(std::string(#call) + ": " + hipGetErrorString(e_));             \
            return MI355_ERR_DEVICE;                                                       \
        }                                                                                  \
    } while (0)

// every ctx entry point: argument check, the ctx lock, and the ctx's device made current for the calling thread
// (several contexts / devices in one process: allocations and launches must land on ctx->device)
#define LOCKED_PROLOGUE                                  \
    if (!ctx) return MI355_ERR_ARG;                      \
    std::lock_guard<std::mutex> lk(ctx->mu);             \
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->set_error("hipSetDevice failed"); return MI355_ERR_DEVICE; }

// grow-only device buffer (workspaces live as long as the ctx: no hipMalloc in steady state)
constexpr int MI355_SIFT_BATCH_MAX = 32;      // frames per SIFT batch (per-frame pointers travel in kernel arguments)
constexpr int MI355_SIFT_KEEPALL_MAX = 32768;  // keypoints per frame the feature record holds with nfeatures <= 0 (keep all); the matcher takes such frames in chunks of 2048 (match.hip, large-pair path)

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t slack = bytes / 4;                     // growth room for small workspaces; bounded: a 100 GB chip buffer must not take 125
        if (slack > ((size_t)256 << 20)) slack = (size_t)256 << 20;
        size_t want = bytes + slack + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// grow-only PINNED host buffer (hipHostMalloc): the landing area of the result exchange's device-to-host copies.  A fresh pageable
// allocation per call (what mi355_allgather_results did until round 5) pays a page fault per 4 KB and a staging copy: 3.7 GB/s measured on
// C5's 1.1 GB of records against the ~50 GB/s a pinned target takes from the same link.
struct HostBuf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// device-resident features of one image (what the reference keeps in d:/feature_temp files)
struct Features {
    int n = 0;          // keypoints
    int npad = 0;       // rows of the matcher's int8 matrix (multiple of 256, zero padded)
    int w = 0, h = 0;   // image size (grid selection needs it)
    DevBuf kp;          // n x mi355_keypoint
    DevBuf xy;          // n x float2
    DevBuf d8;          // n x 128 u8   (the integers OpenCV's SIFT stores in its float Mat)
    DevBuf s8;          // npad x 128 int8 (the same integers minus 128: the MFMA operands)
    DevBuf n8;          // npad x int32  squared norms of the shifted rows
    // asynchronous SIFT: the keypoint count is produced on the device; it lands in pinned host memory and is
    // adopted by mi_resolve_features() the first time the host needs it (match, get_features)
    bool pending = false;
    volatile int* h_cnt = nullptr;   // 8 ints: extrema, refined, keypoints, kept, overflow, ...
    hipEvent_t ready = nullptr;      // recorded after the frame's batch (owned by ctx->batch_events): lets a match wait for ITS frames only
    unsigned caps[3] = {0, 0, 0};
    void release() { kp.release(); xy.release(); d8.release(); s8.release(); n8.release(); }
};

struct ProfClass {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    double bytes = 0.0;
    size_t seen = 0;     // launches of the class since the last reset (bracketed or not)
    int every = 1;       // bracket every n-th launch only (set_option "profile_every:<class>")
};

struct SiftWork;   // sift.hip
struct mi355_comm; // comm.hip (RCCL communicator)
struct SurfState;  // surf.hip (device-resident SURF features)

struct mi355_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every kernel is launched on
    hipStream_t own_stream = nullptr;  // created by the ctx
    std::mutex mu;
    mi355_params p;
    std::string err;
    std::unordered_map<int, Features> feats;
    std::map<std::string, DevBuf> ws;                  // named grow-only workspaces
    std::map<std::string, HostBuf> hws;                // named grow-only pinned host buffers (results handed to the caller stay valid until the next call that fills the same buffer)
    // RANSAC draw tables of every n in [4, 400] for the last seeds used (ransac.hip mi_ransac_tables): four slots, reused in turn (15.9 MB each;
    // built on a side stream so that a survey's new seed costs its pair stage nothing: the build runs beside the matcher)
    struct DrawTables { uint32_t seed = 0; bool valid = false; DevBuf buf, raw; hipEvent_t ready = nullptr; unsigned long long used = 0; };
    DrawTables draw_tables[4];
    unsigned long long draw_clock = 0;
    hipStream_t aux_stream = nullptr;                  // side stream of the table builds
    hipEvent_t aux_ev = nullptr;
    int* draw_flags = nullptr;                         // pinned, one word per slot: the raw rand() stream was too short for some n (never seen; checked at the slot's next use)
    bool profiling = false;
    std::string prof_only;                             // non-empty: bracket only these kernel classes (comma separated)
    std::map<std::string, ProfClass> prof;
    std::vector<SiftWork*> sift_slots;                 // batch work areas, each with its own stream (sift.hip)
    int sift_next = 0;
    std::vector<hipEvent_t> batch_events; size_t batch_events_used = 0;   // one event per enqueued batch, recycled when nothing is pending
    hipEvent_t heavy_ev = nullptr; bool heavy_ev_valid = false;   // option "serial_heavy" (measurement): the chip-filling phases of consecutive batches do not overlap
    int serial_heavy = 0;
    std::vector<DevBuf> host_frames;                   // staging ring of mi355_sift_extract's deferred mode (host frames joining batches)
    std::vector<hipEvent_t> host_frame_ev;             // per ring slot: recorded after the batch that read the slot
    std::vector<char> host_frame_used;
    size_t host_frame_next = 0;
    hipEvent_t pend_event = nullptr;                   // handed to the frame being parked by mi_sift_extract_dev
    int cascade = 3;                                   // octaves >= 2000 px wide: 3 (default) = the first three levels ({gray | L0} -> L0/L1 L2) in one pass (pyr_chain), the rest per level; 2 = also L3..L5 in one pass; 1 = all six in one pass (pyr_cascade); 0 = every level on its own. Same bits; option "sift_cascade"
    int blur_stream = 1;                               // big pyramid levels through blur_stream (0: tile kernel only); option "blur_stream"
    int sift_nslots = 3;                               // batch work areas in flight, each on its own stream (option "sift_slots", env MI355_SIFT_SLOTS)
    int xstream_min_w = 1500, xstream_min_frames = 4;  // extrema_stream for octaves at least this wide (and 3/4 as high) in batches of at least so many frames
    int sift_batch = 16;                               // frames per batch (option "sift_batch", env MI355_SIFT_BATCH)
    int strict_frames = 0;                             // warp.hip: check that a stripe call reads none of the images it was given no pointer for (option "strict_frames": one extra cover pass per call)
    int ransac_split = -1;                             // ransac.hip: workgroups per pair when there are few pairs (-1: by the pair count, 0: never, k: k); option "ransac_split", env MI355_RANSAC_SPLIT
    hipEvent_t sift_in_ev = nullptr;                   // orders the SIFT streams after the caller's stream
    std::vector<int*> pinned_chunks;                   // pinned count slots, 8 ints per frame
    size_t pinned_used = 0;
    int num_cu = 256;
    mi355_comm* comm = nullptr;                        // RCCL communicator (mi355_comm_init), comm.hip
    SurfState* surf = nullptr;                         // SURF variant of the path (surf.hip)
    int last_counts[8] = {0};                          // SIFT counters of the last frame: candidates, refined, keypoints, selected, overflow

    void set_error(const std::string& s) { err = s; }
    DevBuf& buf(const std::string& name) { return ws[name]; }
    HostBuf& hbuf(const std::string& name) { return hws[name]; }
    std::vector<int> deferred_dims;                    // per prepared entry: launch extent (groups of 4 columns, rows)
    std::vector<unsigned char> deferred_warps;         // warp.hip: the chips' warp arguments when mi_chips_and_masks_dev was asked to leave the pixels to mi_chip_pixels_prepare / _launch
    // profiling brackets
    void prof_begin(const char* cls, double alg_bytes, hipStream_t st);
    void prof_end(const char* cls, hipStream_t st);
};

struct ProfScope {
    mi355_ctx* c; const char* cls; hipStream_t st;
    bool on;
    ProfScope(mi355_ctx* c_, const char* cls_, double bytes, hipStream_t st_ = nullptr) : c(c_), cls(cls_), st(st_ ? st_ : c_->stream) {
        on = c->profiling && (c->prof_only.empty() || ("," + c->prof_only + ",").find(std::string(",") + cls + ",") != std::string::npos);
        if (on) { ProfClass& pc = c->prof[cls]; if (pc.every > 1) on = (pc.seen++ % (size_t)pc.every) == 0; }
        if (on) c->prof_begin(cls, bytes, st);
    }
    ~ProfScope() { if (on) c->prof_end(cls, st); }
};

// ---- internal entry points implemented by the .hip files (ctx lock already held) -----------------------
int mi_warp_image(mi355_ctx*, const uint8_t* src, int w, int h, int ws, int ch, const float* h9,
                  uint8_t** dst, int* dw, int* dh, int* dws);
int mi_mosaic_refined_dev(mi355_ctx*, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n,
                          const float* h9s, uint8_t* d_canvas, int cw, int ch, int cws, int row0, int rows, uint8_t* cover_only = nullptr, int cover_exact = 0);
int mi_sift_flush(mi355_ctx*);                               // enqueues every partly filled batch (no wait)
int mi_sift_flush_if_parked(mi355_ctx*, hipEvent_t ev);   // launches the batch still holding a parked frame with this event
int mi_chips_and_masks_dev(mi355_ctx*, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                           const float* h9s, const uint8_t* keep, int find_masks, int* n_chips, mi355_chip_info** chips,
                           std::vector<size_t>& chip_off, std::vector<size_t>& mask_off, int* canvas_w, int* canvas_h, int imgs_on_device = 0,
                           std::vector<int>* owned_bbox = nullptr, int defer_pixels = 0, int row_lo = 0, int row_hi = 0x7fffffff, uint8_t* cover_only = nullptr);      // row_lo .. row_hi: a stripe of the canvas (see warp.hip); find_masks: per chip {min col, min row, max col, max row} of its non-zero mask bytes (max < min: none)
// defer_pixels: the chips' validity masks (and ownership) are made at once, their PIXELS only where asked for afterwards, chip by chip
// (the blender needs them inside a chip's active window only); columns / rows inclusive, clipped to the chip
int mi_chip_pixels_prepare(mi355_ctx*, int n, const int* chips, const int* win4);      // entry e = chip chips[e] inside win4[4e..]: arguments to the device
int mi_chip_pixels_launch(mi355_ctx*, int first, int count);                          // one launch for entries first .. first + count - 1
int mi_mosaic_blended(mi355_ctx*, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                      const uint8_t* keep, int band, uint8_t** out, int* ow, int* oh, int* ows);
int mi_mosaic_blended_dev(mi355_ctx*, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                          const uint8_t* keep, int band, uint8_t* d_canvas, int cw, int ch, int cws, int row0 = 0, int rows = -1, uint8_t* cover_only = nullptr);      // rows >= 0: the stripe row0 .. row0 + rows - 1 of the canvas only
int mi_blend_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int* cw, int* ch);
int mi_multiband_blend(mi355_ctx*, const uint8_t* const* chips, const uint8_t* const* masks, const mi355_chip_info* info, int n,
                       int W, int H, int band, uint8_t** out, int* ow, int* oh, int* ows);
int mi_chips_and_masks(mi355_ctx*, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                       const float* h9s, const uint8_t* keep, int find_masks, int* n_chips, mi355_chip_info** chips,
                       uint8_t*** chip_imgs, uint8_t*** masks, int* cw, int* ch);
int mi_ransac_tables(mi355_ctx*, uint32_t seed, const uint16_t** d_tables);      // the draw tables of a seed (built on a side stream; NULL: prefetch)
int mi_ransac_batch(mi355_ctx*, const mi355_sfpoint* d_p1, const mi355_sfpoint* d_p2, const int* d_n, const int* h_n,
                    int n_pairs, int stride, float dist, int sample_times, uint32_t seed, mi355_pair_result* d_out, int min_keep = -1);
int mi_ransac_big(mi355_ctx*, const mi355_sfpoint* p1, const mi355_sfpoint* p2, int n, float dist, int sample_times, uint32_t seed,
                  mi355_sfpoint* in1, mi355_sfpoint* in2, int* n_in, float* H, int* ok);
int mi_match_pairs_dev(mi355_ctx*, const int32_t* pairs, int n_pairs, float dist, uint32_t seed, mi355_pair_result* d_out);
int mi_bf_match(mi355_ctx*, int img_i, int img_j, int sorted, mi355_dmatch* matches, int32_t* d2, int32_t* second, int maxm, int* nm);
int mi_select_grid(mi355_ctx*, const mi355_dmatch* sorted, int n, const float* kp1, int nk1, const float* kp2, int nk2,
                   int nMatch, int width, int height, int gx, int gy, mi355_sfpoint* v1, mi355_sfpoint* v2, int* n_out);
int mi_set_features(mi355_ctx*, int img_id, const mi355_keypoint* kp, const float* desc, int n, int w, int h);
int mi_finish_features(mi355_ctx*, Features& f, const int* d_n = nullptr, hipStream_t st = nullptr);
int mi_finish_features_batch(mi355_ctx*, Features* const* fs, int nf, const int* d_n, int n_stride, hipStream_t st, int max_rows = 2048);   // all frames of a SIFT batch, one launch   // builds xy / int8 rows / norms from kp + d8 on device
int mi_resolve_features(mi355_ctx*);               // waits for in-flight SIFT frames and adopts their keypoint counts
int mi_resolve_features_of(mi355_ctx*, const int* ids, int n);   // the same for the given frames only (waits for their batches' events)
int mi_sift_extract_dev(mi355_ctx*, int img_id, const uint8_t* d_bgr, int w, int h, int ws, int* n_kp);
void mi_sift_release(mi355_ctx*);
void mi_comm_release(mi355_ctx*);
void mi_surf_release(mi355_ctx*);

// host helpers
int  mi_inverse_matrix_host(const float* src, int order, float* dst, float eps);   // matrix.h:147-296 (host side of the warps)
void mi_glibc_draw_table(uint32_t seed, int n, int max_draws, uint16_t* out4);       // mosaicimage.h:1777-1813
