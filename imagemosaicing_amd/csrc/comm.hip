// csrc/comm.hip -- the multi-GPU exchanges of the path, inside the C ABI: one process per GPU, RCCL over xGMI.
//
// The reference shards its i-loop over threads of ONE process (thread k takes images k mod T, MosaicWithoutPos.cpp:4861 /
// :5066); every thread can read every image's features because they sit in d:/feature_temp files (:5073-5076, 5100-5103),
// and results meet in m_vecMatchPairs under a mutex (PushMatchPairs, :10137-10145).  With one process per GPU the same two
// hand-offs become collectives:
//   mi355_allgather_features   after detect+describe of the rank's own frames (k mod G == rank): every rank receives every
//                              frame's keypoints + descriptors (fixed-size records, 312 KB per frame), so that rank r can match
//                              (i, j) for any j of the reference's window j in (i, i+182) (:5083-5084)
//   mi355_allgather_results    after match+select+RANSAC of the rank's own pairs: the accepted pair records (H + inlier lists)
//                              of all ranks land on every rank's host -- or, with a root, on the host of the one rank that runs the
//                              reference's driver on the inlier lists (ncclSend / ncclRecv) -- in pinned memory the ctx keeps
//   mi355_allgather_moments    the same pairs as 184-byte second moments: what the replicated alignment of every rank starts from
//   mi355_exchange_frames      after the alignment: the frames a rank's canvas stripe reads and the rank does not hold, from their
//                              owners (ncclSend / ncclRecv groups) -- frames are uploaded to ONE GPU each, not to all eight
// All run on the ctx stream; nothing else of the data path crosses ranks.
//
// librccl is bound at run time (dlopen / dlsym) the first time a communicator is created: a process that already holds an RCCL
// (PyTorch-ROCm bundles its own librccl.so.1 next to its own HIP runtime) must use THAT copy -- a second one from /opt/rocm
// would bring a second HIP runtime into the process -- and a single-GPU user of the library does not need RCCL installed.
// The pack / install halves are separate entry points so that a caller with its own transport (the gloo CPU tests, MPI)
// moves the same records.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) { api.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (api.so) break; }      // the copy the process already uses
        if (!api.so) for (const char* n : names) { api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.so) break; }
        if (!api.so) api.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!api.so) { api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.so, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.so, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.so, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.so, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.so, "ncclGetErrorString"));
        api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.so, "ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.so, "ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(api.so, "ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(api.so, "ncclGroupEnd"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.so, "ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.so, "ncclCommUserRank"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString || !api.CommCount || !api.CommUserRank || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd) api.err = "librccl lacks a required symbol";
    });
    return &api;
}

#define MI_NCCL(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) { ctx->set_error(std::string(#call) + ": " + rccl_api()->GetErrorString(r_)); return MI355_ERR_DEVICE; } \
    } while (0)

static_assert(sizeof(mi355_feature_header) == 16, "feature header");
static_assert(MI355_FEATURE_RECORD_BYTES >= 2048 * 28 + 2048 * 128 && MI355_FEATURE_RECORD_BYTES % 256 == 0, "feature record");

// record layout: [0, 57344) keypoints (2048 x 28 B), [57344, 319488) descriptors u8 (2048 x 128); rows beyond n_kp are zero
constexpr size_t REC_KP_BYTES = (size_t)2048 * sizeof(mi355_keypoint), REC_D8_OFF = REC_KP_BYTES, REC_D8_BYTES = (size_t)2048 * 128;

struct PackSrc { const uint8_t* kp; const uint8_t* d8; int n; };

// one workgroup column per record: 16-byte moves, zero fill beyond the frame's n keypoints
__global__ __launch_bounds__(256) void pack_features_kernel(const PackSrc* src, uint8_t* payload) {
    const PackSrc s = src[blockIdx.y];
    uint4* dst = reinterpret_cast<uint4*>(payload + (size_t)blockIdx.y * MI355_FEATURE_RECORD_BYTES);
    const size_t kp_bytes = (size_t)s.n * sizeof(mi355_keypoint), d8_bytes = (size_t)s.n * 128;
    const size_t total16 = MI355_FEATURE_RECORD_BYTES / 16;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total16; q += (size_t)gridDim.x * 256) {
        const size_t b = q * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b < REC_KP_BYTES) {
            if (b + 16 <= kp_bytes) v = *reinterpret_cast<const uint4*>(s.kp + b);
            else if (b < kp_bytes) { uint8_t t[16] = {0}; for (size_t i = 0; b + i < kp_bytes; i++) t[i] = s.kp[b + i]; v = *reinterpret_cast<uint4*>(t); }
        } else if (b < REC_D8_OFF + REC_D8_BYTES) {
            const size_t o = b - REC_D8_OFF;
            if (o + 16 <= d8_bytes) v = *reinterpret_cast<const uint4*>(s.d8 + o);      // d8_bytes is a multiple of 128
        }
        dst[q] = v;
    }
}

// accepted pair records to the front, order kept (what the reference pushes to the driver, MosaicWithoutPos.cpp:5201-5227).  Three launches:
// accepted records per block of 256, an exclusive scan of the block counts by one workgroup, the moves (every lane of a wave helps moving the
// wave's accepted records, 604 x 16 B each).  (One workgroup walking all blocks in turn -- a strided flag read, two barriers and the moves per
// 256 records -- took 5.5 ms for C4's 74 029 records, 0.6 ms for a rank's share of them.)
constexpr int CB = 256;
__global__ __launch_bounds__(CB) void compact_count_kernel(const mi355_pair_result* in, int n, int* blk) {
    __shared__ int s_w[CB / 64];
    const int i = blockIdx.x * CB + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(i < n && in[i].accepted != 0);
    if (lane == 0) s_w[wv] = __builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < CB / 64; q++) t += s_w[q]; blk[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void compact_scan_kernel(int* blk /* counts in, exclusive offsets out */, int nblk, int* n_out) {
    __shared__ int s_w[16], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int i = b0 + tid;
        const int v = i < nblk ? blk[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int off = s_base, tot = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) off += s_w[q]; tot += s_w[q]; }
        if (i < nblk) blk[i] = off + inc - v;
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}
__global__ __launch_bounds__(CB) void compact_move_kernel(const mi355_pair_result* in, int n, const int* blk_off, mi355_pair_result* out) {
    __shared__ int s_w[CB / 64];
    const int i0 = blockIdx.x * CB, i = i0 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(i < n && in[i].accepted != 0);
    if (lane == 0) s_w[wv] = __builtin_popcountll(m);
    __syncthreads();
    int off = blk_off[blockIdx.x];
    for (int q = 0; q < wv; q++) off += s_w[q];
    for (unsigned long long r = m; r; r &= r - 1ull) {
        const int l = __builtin_ctzll(r);
        const uint4* s = reinterpret_cast<const uint4*>(in + (i0 + wv * 64 + l));
        uint4* d = reinterpret_cast<uint4*>(out + (off + __builtin_popcountll(m & ((1ull << l) - 1ull))));
        for (int q = lane; q < (int)(sizeof(mi355_pair_result) / 16); q += 64) d[q] = s[q];
    }
}
// The second moments of a pair's inlier coordinates (mi355_pair_moments): a wave per pair record, lane t < 21 owns ONE of the 21 sums and walks
// the inliers in order -- the chain of separately rounded double products and sums host_io.cpp pair_moments_host forms (this file is compiled
// with -ffp-contract=off like the host file), so the alignment gives the same bits from either.
__global__ __launch_bounds__(256) void pair_moments_kernel(const mi355_pair_result* in, int n, mi355_pair_moments* out) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), t = threadIdx.x & 63;
    if (p >= n) return;
    const mi355_pair_result& e = in[p];
    mi355_pair_moments& o = out[p];
    // a malformed record (n_in beyond the 400 slots of the lists) keeps its n_in and gets no sums: mi355_global_affine_align_moments then
    // answers MI355_ERR_ARG exactly like the record forms do (host_io.cpp), instead of aligning on a plausible-looking clamp
    const int nin = (e.accepted && e.n_in > 0) ? e.n_in : 0;
    const int cnt = nin <= MI355_MAX_SELECTED ? nin : 0;
    if (t == 0) { o.i = e.i; o.j = e.j; o.n_in = nin; o._pad = 0; }
    if (t >= 21) return;
    // operands of this lane's product out of c = (xa, ya, 1, xb, yb, 1): aa 00 10 11 20 21 22 | ab row-major | bb like aa
    int iu, iv;
    if (t < 6)       { const int i = t < 1 ? 0 : (t < 3 ? 1 : 2), j = t - (i * (i + 1)) / 2; iu = i; iv = j; }
    else if (t < 15) { const int q = t - 6; iu = q / 3; iv = 3 + q % 3; }
    else             { const int q = t - 15, i = q < 1 ? 0 : (q < 3 ? 1 : 2), j = q - (i * (i + 1)) / 2; iu = 3 + i; iv = 3 + j; }
    auto pick = [](int k, double xa, double ya, double xb, double yb) { return k == 0 ? xa : (k == 1 ? ya : (k == 3 ? xb : (k == 4 ? yb : 1.0))); };
    double s = 0.0;
    for (int k = 0; k < cnt; k++) {
        const double xa = (double)e.a[k].x, ya = (double)e.a[k].y, xb = (double)e.b[k].x, yb = (double)e.b[k].y;
        const double pr = pick(iu, xa, ya, xb, yb) * pick(iv, xa, ya, xb, yb);
        s = s + pr;
    }
    if (t < 6) o.aa[t] = s; else if (t < 15) o.ab[t - 6] = s; else o.bb[t - 15] = s;
}
static_assert(sizeof(mi355_pair_moments) == 184, "moment record");
static_assert(sizeof(mi355_pair_result) % 16 == 0, "pair record moves in 16-byte pieces");

}  // namespace

// caller holds the ctx lock; d_n_out: device int that receives the number of accepted records
static int compact_accepted(mi355_ctx* ctx, const mi355_pair_result* d_in, int n, mi355_pair_result* d_out, int* d_n_out) {
    const int nblk = (n + CB - 1) / CB;
    DevBuf& dblk = ctx->buf("ag_res_blocks");
    MI_HIP(dblk.reserve(sizeof(int) * (size_t)nblk));
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblk), dim3(CB), 0, ctx->stream, d_in, n, dblk.as<int>());
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, dblk.as<int>(), nblk, d_n_out);
    hipLaunchKernelGGL(compact_move_kernel, dim3(nblk), dim3(CB), 0, ctx->stream, d_in, n, dblk.as<int>(), d_out);
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

struct mi355_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

// ---- pack / install (transport-agnostic halves) ---------------------------------------------------------------------------
extern "C" int mi355_pack_features_dev(mi355_ctx* ctx, const int32_t* img_ids, int n, mi355_feature_header* hdr, void* d_payload) {
    LOCKED_PROLOGUE
    if (n < 0 || (n > 0 && (!img_ids || !hdr || !d_payload))) return MI355_ERR_ARG;
    if (n == 0) return MI355_OK;
    { int rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; }
    std::vector<PackSrc> src(n);
    for (int k = 0; k < n; k++) {
        auto it = ctx->feats.find(img_ids[k]);
        if (it == ctx->feats.end()) { ctx->set_error("pack_features: no resident features for image " + std::to_string(img_ids[k])); return MI355_ERR_ARG; }
        const Features& f = it->second;
        if (f.n > 2048) { ctx->set_error("pack_features: more than 2048 keypoints"); return MI355_ERR_ARG; }
        hdr[k].img_id = img_ids[k]; hdr[k].n_kp = f.n; hdr[k].w = f.w; hdr[k].h = f.h;
        src[k] = PackSrc{f.kp.as<uint8_t>(), f.d8.as<uint8_t>(), f.n};
    }
    DevBuf& dsrc = ctx->buf("pack_src");
    MI_HIP(dsrc.reserve(sizeof(PackSrc) * n));
    MI_HIP(hipMemcpyAsync(dsrc.p, src.data(), sizeof(PackSrc) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pack_features_kernel, dim3(16, n), dim3(256), 0, ctx->stream, dsrc.as<PackSrc>(), reinterpret_cast<uint8_t*>(d_payload));
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(ctx->stream));           // `src` goes out of scope
    return MI355_OK;
}

// One launch installs every received record: keypoints and descriptors move into the image's resident buffers and the matcher's
// operands (xy, int8 rows, squared norms) are derived on the way -- the per-frame form (two copies + finish_features per frame)
// costs ~1500 API calls per step at C4.
struct InstallDst { const uint8_t* rec; mi355_keypoint* kp; uint8_t* d8; float2* xy; int8_t* s8; int* n8; int n, npad; };

__global__ __launch_bounds__(256) void install_features_kernel(const InstallDst* tab) {
    const InstallDst t = tab[blockIdx.y];
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;     // 8 lanes per descriptor row, 16 bytes each
    if (row >= t.npad) return;
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);           // padding rows: zeros after the shift
    if (row < t.n) {
        v = *reinterpret_cast<const uint4*>(t.rec + REC_D8_OFF + (size_t)row * 128 + part * 16);
        *reinterpret_cast<uint4*>(t.d8 + (size_t)row * 128 + part * 16) = v;
    }
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int b = 0; b < 4; b++) { const int u = (int)((w[q] >> (8 * b)) & 0xffu) - 128; s += u * u; }
    // the matcher's int8 operand (match.hip): u - 128 = u ^ 0x80 as a byte
    *reinterpret_cast<uint4*>(t.s8 + (size_t)row * 128 + part * 16) = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0) t.n8[row] = s;
    if (row < t.n && part < 7) {
        const unsigned kw = reinterpret_cast<const unsigned*>(t.rec + (size_t)row * sizeof(mi355_keypoint))[part];
        reinterpret_cast<unsigned*>(t.kp + row)[part] = kw;
        if (part < 2) reinterpret_cast<unsigned*>(t.xy + row)[part] = kw;               // x, y are the first two fields
    }
}

static int install_features(mi355_ctx* ctx, const mi355_feature_header* hdr, const void* d_payload, int n, const int32_t* skip_ids, int n_skip) {
    std::vector<InstallDst> tab;
    tab.reserve(n);
    bool resolved = false;
    for (int k = 0; k < n; k++) {
        const mi355_feature_header& hk = hdr[k];
        if (hk.img_id < 0) continue;                     // padding record of a rank with fewer frames
        bool skip = false;
        for (int q = 0; q < n_skip; q++) if (skip_ids[q] == hk.img_id) { skip = true; break; }
        if (skip) continue;
        if (hk.n_kp < 0 || hk.n_kp > 2048 || hk.w <= 0 || hk.h <= 0) { ctx->set_error("install_features: bad record header"); return MI355_ERR_ARG; }
        auto it = ctx->feats.find(hk.img_id);
        if (it != ctx->feats.end() && it->second.pending && !resolved) { int rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; resolved = true; }
        Features& f = ctx->feats[hk.img_id];
        f.n = hk.n_kp; f.w = hk.w; f.h = hk.h; f.pending = false; f.h_cnt = nullptr;
        f.npad = ((f.n + 255) / 256) * 256;
        if (f.npad == 0) f.npad = 256;
        // sized for 2048 keypoints once: no allocation in steady state when the counts change from step to step
        MI_HIP(f.kp.reserve(sizeof(mi355_keypoint) * 2048));
        MI_HIP(f.d8.reserve((size_t)128 * 2048));
        MI_HIP(f.xy.reserve(sizeof(float2) * 2048));
        MI_HIP(f.s8.reserve((size_t)128 * 2048));
        MI_HIP(f.n8.reserve(sizeof(int) * 2048));
        InstallDst t;
        t.rec = reinterpret_cast<const uint8_t*>(d_payload) + (size_t)k * MI355_FEATURE_RECORD_BYTES;
        t.kp = f.kp.as<mi355_keypoint>(); t.d8 = f.d8.as<uint8_t>(); t.xy = f.xy.as<float2>(); t.s8 = f.s8.as<int8_t>(); t.n8 = f.n8.as<int>();
        t.n = f.n; t.npad = f.npad;
        tab.push_back(t);
    }
    if (tab.empty()) return MI355_OK;
    DevBuf& dtab = ctx->buf("install_tab");
    MI_HIP(dtab.reserve(sizeof(InstallDst) * tab.size()));
    MI_HIP(hipMemcpyAsync(dtab.p, tab.data(), sizeof(InstallDst) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(install_features_kernel, dim3(2048 / 32, (unsigned)tab.size()), dim3(256), 0, ctx->stream, dtab.as<InstallDst>());
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(ctx->stream));           // `tab` goes out of scope; the caller may reuse d_payload
    return MI355_OK;
}

extern "C" int mi355_install_features_dev(mi355_ctx* ctx, const mi355_feature_header* hdr, const void* d_payload, int n) {
    LOCKED_PROLOGUE
    if (n < 0 || (n > 0 && (!hdr || !d_payload))) return MI355_ERR_ARG;
    return install_features(ctx, hdr, d_payload, n, nullptr, 0);
}

// ---- communicator ------------------------------------------------------------------------------------------------------------
extern "C" int mi355_comm_unique_id(uint8_t id128[128]) {
    if (!id128) return MI355_ERR_ARG;
    RcclApi* api = rccl_api();
    if (!api->err.empty()) return MI355_ERR_DEVICE;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    if (api->GetUniqueId(&id) != ncclSuccess) return MI355_ERR_DEVICE;
    memcpy(id128, &id, 128);
    return MI355_OK;
}

extern "C" int mi355_comm_available(void) {
    return rccl_api()->err.empty() ? MI355_OK : MI355_ERR_DEVICE;
}

extern "C" int mi355_comm_info(mi355_ctx* ctx, int* rank, int* n_ranks) {
    LOCKED_PROLOGUE
    if (!rank || !n_ranks) return MI355_ERR_ARG;
    *rank = 0; *n_ranks = 0;
    if (!ctx->comm) { ctx->set_error("comm_info: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
    RcclApi* api = rccl_api();
    MI_NCCL(api->CommUserRank(ctx->comm->comm, rank));
    MI_NCCL(api->CommCount(ctx->comm->comm, n_ranks));
    return MI355_OK;
}

extern "C" int mi355_comm_init(mi355_ctx* ctx, const uint8_t id128[128], int rank, int world) {
    LOCKED_PROLOGUE
    if (!id128 || world < 1 || rank < 0 || rank >= world) return MI355_ERR_ARG;
    RcclApi* api = rccl_api();
    if (!api->err.empty()) { ctx->set_error(api->err); return MI355_ERR_DEVICE; }
    if (ctx->comm) { ctx->set_error("comm_init: the ctx already has a communicator"); return MI355_ERR_ARG; }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    mi355_comm* c = new (std::nothrow) mi355_comm();
    if (!c) return MI355_ERR_NOMEM;
    c->rank = rank; c->world = world;
    const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { ctx->set_error(std::string("ncclCommInitRank: ") + api->GetErrorString(r)); delete c; return MI355_ERR_DEVICE; }
    ctx->comm = c;
    return MI355_OK;
}

void mi_comm_release(mi355_ctx* ctx) {
    if (!ctx->comm) return;
    if (ctx->comm->comm) (void)rccl_api()->CommDestroy(ctx->comm->comm);
    delete ctx->comm;
    ctx->comm = nullptr;
}

extern "C" int mi355_comm_destroy(mi355_ctx* ctx) {
    LOCKED_PROLOGUE
    (void)hipStreamSynchronize(ctx->stream);
    mi_comm_release(ctx);
    return MI355_OK;
}

// ---- collectives ---------------------------------------------------------------------------------------------------------------
extern "C" int mi355_allgather_features(mi355_ctx* ctx, const int32_t* img_ids, int n_local, int n_max_per_rank) {
    LOCKED_PROLOGUE
    if (!ctx->comm) { ctx->set_error("allgather_features: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
    if (n_max_per_rank < 1) return MI355_ERR_ARG;        // the same value on every rank: a bad one fails everywhere alike
    RcclApi* api = rccl_api();
    const int world = ctx->comm->world, rank = ctx->comm->rank;
    const size_t nm = (size_t)n_max_per_rank;
    DevBuf& dhdr = ctx->buf("ag_feat_hdr");              // [world][n_max] headers, [world][n_max] records; this rank's block is the send buffer
    DevBuf& dpay = ctx->buf("ag_feat_payload");
    MI_HIP(dhdr.reserve(sizeof(mi355_feature_header) * nm * world));
    MI_HIP(dpay.reserve((size_t)MI355_FEATURE_RECORD_BYTES * nm * world));
    std::vector<mi355_feature_header> hdr(nm * world);
    for (auto& hk : hdr) { hk.img_id = -1; hk.n_kp = 0; hk.w = 0; hk.h = 0; }
    mi355_feature_header* my_hdr = hdr.data() + nm * rank;
    uint8_t* my_pay = dpay.as<uint8_t>() + (size_t)MI355_FEATURE_RECORD_BYTES * nm * rank;
    // Everything that can fail on THIS rank alone happens in `local`; its outcome travels in the headers (img_id == -2), and the
    // rank takes part in both collectives whatever it was -- a rank that returned early would leave the others in ncclAllGather.
    auto local = [&]() -> int {
        if (n_local < 0 || n_max_per_rank < n_local || (n_local > 0 && !img_ids)) { ctx->set_error("allgather_features: bad arguments"); return MI355_ERR_ARG; }
        { int rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; }
        if (n_local == 0) return MI355_OK;
        std::vector<PackSrc> src(n_local);
        for (int k = 0; k < n_local; k++) {
            auto it = ctx->feats.find(img_ids[k]);
            if (it == ctx->feats.end()) { ctx->set_error("allgather_features: no resident features for image " + std::to_string(img_ids[k])); return MI355_ERR_ARG; }
            const Features& f = it->second;
            if (f.n > 2048) { ctx->set_error("allgather_features: more than 2048 keypoints"); return MI355_ERR_ARG; }       // as mi355_pack_features_dev
            my_hdr[k].img_id = img_ids[k]; my_hdr[k].n_kp = f.n; my_hdr[k].w = f.w; my_hdr[k].h = f.h;
            src[k] = PackSrc{f.kp.as<uint8_t>(), f.d8.as<uint8_t>(), f.n};
        }
        DevBuf& dsrc = ctx->buf("pack_src");
        MI_HIP(dsrc.reserve(sizeof(PackSrc) * n_local));
        MI_HIP(hipMemcpyAsync(dsrc.p, src.data(), sizeof(PackSrc) * n_local, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(pack_features_kernel, dim3(16, n_local), dim3(256), 0, ctx->stream, dsrc.as<PackSrc>(), my_pay);
        MI_HIP(hipGetLastError());
        MI_HIP(hipStreamSynchronize(ctx->stream));       // `src` goes out of scope
        return MI355_OK;
    };
    const int rc_local = local();
    std::string err_local;
    if (rc_local != MI355_OK) {
        err_local = ctx->err;
        for (size_t k = 0; k < nm; k++) { my_hdr[k].img_id = -2; my_hdr[k].n_kp = 0; my_hdr[k].w = 0; my_hdr[k].h = 0; }
    }
    MI_HIP(hipMemcpyAsync(dhdr.as<mi355_feature_header>() + nm * rank, my_hdr, sizeof(mi355_feature_header) * nm, hipMemcpyHostToDevice, ctx->stream));
    // in-place all-gathers (send buffer = this rank's block of the receive buffer)
    MI_NCCL(api->AllGather(dhdr.as<mi355_feature_header>() + nm * rank, dhdr.p, sizeof(mi355_feature_header) * nm, ncclChar, ctx->comm->comm, ctx->stream));
    MI_NCCL(api->AllGather(my_pay, dpay.p, (size_t)MI355_FEATURE_RECORD_BYTES * nm, ncclChar, ctx->comm->comm, ctx->stream));
    MI_HIP(hipMemcpyAsync(hdr.data(), dhdr.p, sizeof(mi355_feature_header) * nm * world, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    if (rc_local != MI355_OK) { ctx->set_error(err_local); return rc_local; }
    for (int r = 0; r < world; r++)
        if (hdr[nm * r].img_id == -2) { ctx->set_error("allgather_features: rank " + std::to_string(r) + " failed before the exchange"); return MI355_ERR_FAILED; }
    return install_features(ctx, hdr.data(), dpay.p, (int)(nm * world), img_ids, n_local);      // own frames are resident already
}

extern "C" int mi355_allgather_results(mi355_ctx* ctx, const mi355_pair_result* d_local, int n_local, int flags, int root,
                                       const mi355_pair_result** all, int* n_all) {
    LOCKED_PROLOGUE
    const int accepted_only = flags & MI355_GATHER_ACCEPTED_ONLY;
    const bool no_wait = (flags & MI355_GATHER_NO_WAIT) != 0 && root >= 0;
    if (!ctx->comm) { ctx->set_error("allgather_results: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
    if (all) *all = nullptr;
    if (n_all) *n_all = 0;
    // bad arguments of ONE rank travel as a negative count: every rank then returns an error after the first collective
    RcclApi* api = rccl_api();
    const int world = ctx->comm->world, rank = ctx->comm->rank;
    const bool bad_local = n_local < 0 || (n_local > 0 && !d_local) || !all || !n_all || root >= world;
    if (bad_local) n_local = 0;
    // 1. counts (after the optional compaction)
    DevBuf& dcnt = ctx->buf("ag_res_counts");
    MI_HIP(dcnt.reserve(sizeof(int) * (size_t)(world + 1)));
    int* d_counts = dcnt.as<int>();
    DevBuf& dcomp = ctx->buf("ag_res_compact");
    const mi355_pair_result* d_send = d_local;
    int n_send = n_local;
    if (bad_local) {
        n_send = -1;
        MI_HIP(hipMemcpyAsync(d_counts + rank, &n_send, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    } else if (accepted_only && n_local > 0) {
        MI_HIP(dcomp.reserve(sizeof(mi355_pair_result) * (size_t)n_local));
        { const int rc = compact_accepted(ctx, d_local, n_local, dcomp.as<mi355_pair_result>(), d_counts + rank); if (rc != MI355_OK) return rc; }
        d_send = dcomp.as<mi355_pair_result>();
    } else {
        MI_HIP(hipMemcpyAsync(d_counts + rank, &n_send, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    MI_NCCL(api->AllGather(d_counts + rank, d_counts, sizeof(int), ncclChar, ctx->comm->comm, ctx->stream));
    std::vector<int> counts(world);
    MI_HIP(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * world, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    int n_max = 1; size_t total = 0;
    if (bad_local) { ctx->set_error("allgather_results: bad arguments"); return MI355_ERR_ARG; }
    for (int r = 0; r < world; r++) { if (counts[r] < 0) { ctx->set_error("allgather_results: rank " + std::to_string(r) + " failed before the exchange"); return MI355_ERR_FAILED; } if (counts[r] > n_max) n_max = counts[r]; total += (size_t)counts[r]; }
    n_send = counts[rank];
    const size_t REC = sizeof(mi355_pair_result);
    DevBuf& dall = ctx->buf("ag_res_all");
    HostBuf& hall = ctx->hbuf("ag_res_host");            // pinned, kept by the ctx: the caller's view of the result until the next call
    if (root < 0) {
        // 2a. payload to every rank, padded to the largest rank's count
        MI_HIP(dall.reserve(REC * (size_t)n_max * world));
        mi355_pair_result* my = dall.as<mi355_pair_result>() + (size_t)n_max * rank;
        if (n_send > 0) MI_HIP(hipMemcpyAsync(my, d_send, REC * (size_t)n_send, hipMemcpyDeviceToDevice, ctx->stream));
        MI_NCCL(api->AllGather(my, dall.p, REC * (size_t)n_max, ncclChar, ctx->comm->comm, ctx->stream));
        // 3a. to the host, rank-major, padding dropped
        MI_HIP(hall.reserve(REC * (total > 0 ? total : 1)));
        size_t o = 0;
        for (int r = 0; r < world; r++) {
            if (counts[r] > 0) MI_HIP(hipMemcpyAsync(hall.as<mi355_pair_result>() + o, dall.as<mi355_pair_result>() + (size_t)n_max * r, REC * (size_t)counts[r], hipMemcpyDeviceToHost, ctx->stream));
            o += (size_t)counts[r];
        }
        MI_HIP(hipStreamSynchronize(ctx->stream));
        *all = hall.as<mi355_pair_result>(); *n_all = (int)total;
        return MI355_OK;
    }
    // 2b. payload to the root alone: every other rank's block lands at its rank-major offset (no padding), one copy takes it to the host
    if (rank == root) {
        MI_HIP(dall.reserve(REC * (total > 0 ? total : 1)));
        std::vector<size_t> off(world, 0);
        for (int r = 1; r < world; r++) off[r] = off[r - 1] + (size_t)counts[r - 1];
        MI_NCCL(api->GroupStart());
        for (int r = 0; r < world; r++)
            if (r != root && counts[r] > 0) MI_NCCL(api->Recv(dall.as<mi355_pair_result>() + off[r], REC * (size_t)counts[r], ncclChar, r, ctx->comm->comm, ctx->stream));
        MI_NCCL(api->GroupEnd());
        if (n_send > 0) MI_HIP(hipMemcpyAsync(dall.as<mi355_pair_result>() + off[rank], d_send, REC * (size_t)n_send, hipMemcpyDeviceToDevice, ctx->stream));
        MI_HIP(hall.reserve(REC * (total > 0 ? total : 1)));
        if (total > 0) MI_HIP(hipMemcpyAsync(hall.p, dall.p, REC * total, hipMemcpyDeviceToHost, ctx->stream));
        if (!no_wait) MI_HIP(hipStreamSynchronize(ctx->stream));      // MI355_GATHER_NO_WAIT: the copy runs beside the caller's host work (the replicated alignment)
        *all = hall.as<mi355_pair_result>();
    } else if (n_send > 0) {
        MI_NCCL(api->GroupStart());
        MI_NCCL(api->Send(d_send, REC * (size_t)n_send, ncclChar, root, ctx->comm->comm, ctx->stream));
        MI_NCCL(api->GroupEnd());
    }
    *n_all = (int)total;
    return MI355_OK;
}

// device-side compaction alone (callers with their own transport): accepted records to the front of d_out, count to the host
extern "C" int mi355_compact_accepted_dev(mi355_ctx* ctx, const mi355_pair_result* d_in, int n, mi355_pair_result* d_out, int* n_out) {
    LOCKED_PROLOGUE
    if (n < 0 || !n_out || (n > 0 && (!d_in || !d_out))) return MI355_ERR_ARG;
    *n_out = 0;
    if (n == 0) return MI355_OK;
    DevBuf& dcnt = ctx->buf("ag_res_counts");
    MI_HIP(dcnt.reserve(sizeof(int) * 64));
    { const int rc = compact_accepted(ctx, d_in, n, d_out, dcnt.as<int>()); if (rc != MI355_OK) return rc; }
    MI_HIP(hipMemcpyAsync(n_out, dcnt.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

extern "C" int mi355_pair_moments_dev(mi355_ctx* ctx, const mi355_pair_result* d_results, int n, mi355_pair_moments* d_out) {
    LOCKED_PROLOGUE
    if (n < 0 || (n > 0 && (!d_results || !d_out))) return MI355_ERR_ARG;
    if (n == 0) return MI355_OK;
    hipLaunchKernelGGL(pair_moments_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, d_results, n, d_out);
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

// mi355_allgather_results' protocol (counts first, a rank with bad arguments sends -1, payload padded to the largest count) on the moments of
// this rank's accepted pairs
extern "C" int mi355_allgather_moments(mi355_ctx* ctx, const mi355_pair_result* d_local, int n_local, const mi355_pair_moments** all, int* n_all) {
    LOCKED_PROLOGUE
    if (!ctx->comm) { ctx->set_error("allgather_moments: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
    if (all) *all = nullptr;
    if (n_all) *n_all = 0;
    const bool bad_local = n_local < 0 || (n_local > 0 && !d_local) || !all || !n_all;
    if (bad_local) n_local = 0;
    RcclApi* api = rccl_api();
    const int world = ctx->comm->world, rank = ctx->comm->rank;
    DevBuf& dcnt = ctx->buf("ag_res_counts");
    MI_HIP(dcnt.reserve(sizeof(int) * (size_t)(world + 1)));
    int* d_counts = dcnt.as<int>();
    DevBuf& dcomp = ctx->buf("ag_res_compact");
    int n_send = n_local;
    if (bad_local) n_send = -1;
    if (bad_local || n_local == 0) MI_HIP(hipMemcpyAsync(d_counts + rank, &n_send, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    else {
        MI_HIP(dcomp.reserve(sizeof(mi355_pair_result) * (size_t)n_local));
        const int rc = compact_accepted(ctx, d_local, n_local, dcomp.as<mi355_pair_result>(), d_counts + rank);
        if (rc != MI355_OK) return rc;
    }
    MI_NCCL(api->AllGather(d_counts + rank, d_counts, sizeof(int), ncclChar, ctx->comm->comm, ctx->stream));
    std::vector<int> counts(world);
    MI_HIP(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * world, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));                  // (n_send's host copy is consumed)
    int n_max = 1; size_t total = 0;
    if (bad_local) { ctx->set_error("allgather_moments: bad arguments"); return MI355_ERR_ARG; }
    for (int r = 0; r < world; r++) { if (counts[r] < 0) { ctx->set_error("allgather_moments: rank " + std::to_string(r) + " failed before the exchange"); return MI355_ERR_FAILED; } if (counts[r] > n_max) n_max = counts[r]; total += (size_t)counts[r]; }
    n_send = counts[rank];
    DevBuf& dall = ctx->buf("ag_mom_all");
    MI_HIP(dall.reserve(sizeof(mi355_pair_moments) * (size_t)n_max * world));
    mi355_pair_moments* my = dall.as<mi355_pair_moments>() + (size_t)n_max * rank;
    if (n_send > 0) {
        hipLaunchKernelGGL(pair_moments_kernel, dim3((n_send + 3) / 4), dim3(256), 0, ctx->stream, dcomp.as<mi355_pair_result>(), n_send, my);
        MI_HIP(hipGetLastError());
    }
    MI_NCCL(api->AllGather(my, dall.p, sizeof(mi355_pair_moments) * (size_t)n_max, ncclChar, ctx->comm->comm, ctx->stream));
    HostBuf& hall = ctx->hbuf("ag_mom_host");
    MI_HIP(hall.reserve(sizeof(mi355_pair_moments) * (total > 0 ? total : 1)));
    size_t o = 0;
    for (int r = 0; r < world; r++) {
        if (counts[r] > 0) MI_HIP(hipMemcpyAsync(hall.as<mi355_pair_moments>() + o, dall.as<mi355_pair_moments>() + (size_t)n_max * r, sizeof(mi355_pair_moments) * (size_t)counts[r], hipMemcpyDeviceToHost, ctx->stream));
        o += (size_t)counts[r];
    }
    MI_HIP(hipStreamSynchronize(ctx->stream));
    *all = hall.as<mi355_pair_moments>(); *n_all = (int)total;
    return MI355_OK;
}

// ---- frame ownership (primary form) ---------------------------------------------------------------------------------------
// The reference composites in ONE address space: every image is in host memory when MosaicImagesRefined / LaplacianPyramidBlending walk them
// (MosaicWithoutPos.cpp:4663, 4671; MosaicImage.cpp:2306-2460).  With one process per GPU a frame lives where it was extracted (k mod G,
// :4861); a canvas stripe reads the frames that cross it.  Which ones is known on every rank once the transforms are (the alignment is
// replicated), so the whole schedule is a table every rank walks alike: no negotiation, no counts, only the transfers themselves.
extern "C" int mi355_exchange_frames(mi355_ctx* ctx, const uint8_t* const* d_frames, const int* h, const int* ws, int n, const int32_t* owner,
                                     const uint8_t* need, int flags, const uint8_t** d_out, uint64_t* bytes_recv, uint64_t* bytes_sent) {
    LOCKED_PROLOGUE
    if (!ctx->comm) { ctx->set_error("exchange_frames: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
    if (n < 0 || (n > 0 && (!d_frames || !h || !ws || !need || !d_out))) { ctx->set_error("exchange_frames: bad arguments"); return MI355_ERR_ARG; }
    RcclApi* api = rccl_api();
    const int world = ctx->comm->world, rank = ctx->comm->rank;
    const bool own_too = (flags & MI355_EXCHANGE_OWN_THROUGH_RCCL) != 0;
    auto own = [&](int k) { return owner ? owner[k] : k % world; };
    // What can fail on this rank alone is checked BEFORE anything is posted -- geometry, "a rank holds the frames it owns", memory for its landing
    // area -- and the verdict travels with the cover rows (one more byte per rank): a rank that left early would leave its peers waiting inside
    // ncclRecv; like the other exchanges, every rank returns the error together.  The landing area depends on the rank's OWN row only.
    const bool local_rows = (flags & MI355_EXCHANGE_NEED_IS_LOCAL) != 0;
    const uint8_t* my_row = local_rows ? need : need + (size_t)rank * n;
    std::vector<size_t> slot(n > 0 ? n : 1, (size_t)-1);
    size_t arena = 0;
    uint64_t rb = 0, sb = 0;
    DevBuf& dar = ctx->buf("frame_exchange");
    auto local = [&]() -> int {
        for (int k = 0; k < n; k++) {
            const int o = own(k);
            if (o < 0 || o >= world || h[k] < 1 || ws[k] < 1) { ctx->set_error("exchange_frames: bad owner / geometry of frame " + std::to_string(k)); return MI355_ERR_ARG; }
            d_out[k] = nullptr;
            if (o == rank && !d_frames[k]) { ctx->set_error("exchange_frames: this rank owns frame " + std::to_string(k) + " but holds no pointer to it"); return MI355_ERR_ARG; }
            if (!my_row[k]) continue;
            if (o == rank && !own_too) { d_out[k] = d_frames[k]; continue; }
            const size_t bytes = (size_t)ws[k] * h[k];
            slot[k] = arena; arena += (bytes + 255) & ~(size_t)255;
            if (o != rank) rb += bytes;
        }
        MI_HIP(dar.reserve(arena + 256));
        return MI355_OK;
    };
    const int rc_local = local();
    const std::string err_local = rc_local != MI355_OK ? ctx->err : std::string();
    std::vector<uint8_t> table;
    {
        // one all-gather: [cover row (n bytes, only when the rows are local) | status byte], padded to 16 bytes per rank
        const size_t nrow = local_rows ? (size_t)n : 0, row = (nrow + 1 + 15) & ~(size_t)15;
        DevBuf& dn = ctx->buf("frame_need_rows");
        MI_HIP(dn.reserve(row * world + 16));
        std::vector<uint8_t> mine(row, 0), padded(row * world);
        if (nrow) memcpy(mine.data(), need, nrow);
        mine[nrow] = rc_local == MI355_OK ? 0 : 1;
        MI_HIP(hipMemcpyAsync(dn.as<uint8_t>() + row * rank, mine.data(), row, hipMemcpyHostToDevice, ctx->stream));
        MI_NCCL(api->AllGather(dn.as<uint8_t>() + row * rank, dn.p, row, ncclChar, ctx->comm->comm, ctx->stream));
        MI_HIP(hipMemcpyAsync(padded.data(), dn.p, row * world, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
        if (rc_local != MI355_OK) { ctx->set_error(err_local); return rc_local; }
        for (int r = 0; r < world; r++) if (padded[row * r + nrow]) { ctx->set_error("exchange_frames: rank " + std::to_string(r) + " failed before the exchange"); return MI355_ERR_FAILED; }
        if (local_rows) {
            table.resize((size_t)n * world);
            for (int r = 0; r < world; r++) memcpy(table.data() + (size_t)r * n, padded.data() + row * r, (size_t)n);
            need = table.data();
        }
    }
    for (int k = 0; k < n; k++) {                         // what this rank sends
        if (own(k) != rank) continue;
        const size_t bytes = (size_t)ws[k] * h[k];
        for (int r = 0; r < world; r++) if (r != rank && need[(size_t)r * n + k]) sb += bytes;
    }
    for (int k = 0; k < n; k++) if (slot[k] != (size_t)-1) d_out[k] = dar.as<uint8_t>() + slot[k];
    constexpr int RUN = 64;                               // frames per ncclGroup: bounds the operations one group carries (C5: ~480 receives per rank in all)
    for (int k0 = 0; k0 < n; k0 += RUN) {
        const int k1 = k0 + RUN < n ? k0 + RUN : n;
        bool any = false;
        for (int k = k0; k < k1 && !any; k++) {
            const int o = own(k);
            for (int r = 0; r < world; r++) if (need[(size_t)r * n + k] && (r != o || own_too) && (o == rank || r == rank)) { any = true; break; }
        }
        if (!any) continue;
        MI_NCCL(api->GroupStart());
        for (int k = k0; k < k1; k++) {
            const int o = own(k);
            const size_t bytes = (size_t)ws[k] * h[k];
            for (int r = 0; r < world; r++) {
                if (!need[(size_t)r * n + k] || (r == o && !own_too)) continue;
                if (o == rank) MI_NCCL(api->Send(d_frames[k], bytes, ncclChar, r, ctx->comm->comm, ctx->stream));
                if (r == rank) MI_NCCL(api->Recv(dar.as<uint8_t>() + slot[k], bytes, ncclChar, o, ctx->comm->comm, ctx->stream));
            }
        }
        MI_NCCL(api->GroupEnd());
    }
    if (bytes_recv) *bytes_recv = rb;
    if (bytes_sent) *bytes_sent = sb;
    return MI355_OK;
}
