// csrc/sift.hip -- K1..K5: SIFT detect + describe on gfx950, replacing the body of SiftExtraction_Thread
// (MosaicWithoutPos.cpp:4832-4887: SiftFeatureDetector(2000,3,0.01,20).detect + SiftDescriptorExtractor.compute).
//
// The algorithm is OpenCV 2.4.0's cv::SIFT as the reference's own binary runs it (oracle/oracle_sift.c says how that was
// established and how the reference's committed run pins it): NO image doubling, 16-BIT FIXED-POINT pyramids (gray x 48), Gaussian
// levels filtered 16S -> 32F -> 16S with separately rounded products and sums in the filter engine's tap order, integer DoG,
// |DoG| > 20, Cramer's rule for the 3x3 fit.  This file implements that definition, independently, for the GPU, and the parity
// tests compare keypoints and descriptors bit for bit with the oracle.
//
// Host side: frames collect into batches (SiftWork, sift_run_batch at the end of this file); every launch below covers all frames
// of a batch.  Kernels:
//   blur16_stream<R,BGR>  separable Gaussian of one level (>= 512 columns): a wave walks down a strip of 256 columns, rows arrive
//                         once from HBM (8 bytes per lane: four 16-bit samples), the row-filtered rows of the last 2R+1 steps live in
//                         registers (the column pass needs its taps centre first, then symmetric pairs: a gather, not a scatter),
//                         no workgroup barrier; BGR = true forms gray x 48 of the caller's frame on the fly (base level, no gray
//                         image is ever stored); level 3 also writes the decimated base of the next octave
//   blur16_tile<R,BGR>    the same filter for small levels: 64x32 tile + halo in LDS, row pass then column pass
//   downsample16          next octave seed = every second pixel of level 3 (when the blur did not write it on the way out)
//   extrema_stream / extrema_kernel   DoG never materialised in HBM: the 6 Gaussian levels are read once, 26-neighbour
//                         test on the 5 DoG planes (registers / LDS), candidates leave with their 3x3x3 neighbourhood
//   refine                one lane per candidate: quadratic fit, contrast / edge tests, duplicate claim bitmap
//   resp_threshold        response threshold above which nfeatures + 256 refined points lie (radix select)
//   orient                one wave per refined point above the threshold: 36-bin histogram in order-free fixed point,
//                         smoothing + peaks via lane shuffles
//   topk                  one workgroup per frame: radix select of the nfeatures-th response, bitonic sort of the
//                         survivors by the total order (response desc, octave, layer, row, col, bin)
//   describe              one workgroup per keypoint: trilinear contributions quantised to 2^-10 and added with
//                         64-bit LDS atomics (order-free by definition), normalise / clip / renormalise -> u8
#include "common.h"
#include <memory>
#include "detmath.h"
#include <cmath>
#include <type_traits>

namespace {

constexpr int N_LAYERS = 3, N_LEVELS = 6, IMG_BORDER = 5, MAX_INTERP = 5, ORI_BINS = 36, MAX_OCT = 16;
constexpr int FIXPT_SCALE = 48;                 // SIFT_FIXPT_SCALE of the reference's OpenCV build: pyramid samples are gray x 48 in 16 bits
constexpr float DOG_THRESHOLD_P1 = 21.0f;       // |DoG| > floor(0.5 * 0.01 / 3 * 255 * 48) = 20, on integers: |DoG| >= 21
constexpr float HIST_Q = 1024.0f;               // order-free histogram accumulation: contributions quantised to 2^-10 (oracle_sift.c)
constexpr int MAX_R = 16;
constexpr int SIFT_BATCH_MAX = MI355_SIFT_BATCH_MAX;
typedef int16_t lvl_t;                          // one pyramid sample

__host__ __device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

typedef float v2f __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(4))) uint2_a4 { unsigned x, y; };     // two dwords at a 4-byte aligned address (global_load_dwordx2 asks for no more)
typedef float v4f __attribute__((ext_vector_type(4)));

// four / one 16-bit samples as floats (exact)
__device__ __forceinline__ v4f ld4(const lvl_t* p) {              // p 8-byte aligned
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    v4f o;
    o.x = (float)(int)(short)(q.x & 0xffffu); o.y = (float)((int)q.x >> 16);
    o.z = (float)(int)(short)(q.y & 0xffffu); o.w = (float)((int)q.y >> 16);
    return o;
}
__device__ __forceinline__ float ld1(const lvl_t* p) { return (float)(int)*p; }
// saturate_cast<short>(float): round half to even, clamp
__device__ __forceinline__ int sat16(float v) {
    int q = (int)rintf(v);
    q = q < -32768 ? -32768 : (q > 32767 ? 32767 : q);
    return q;
}
__device__ __forceinline__ float gray48(const uint8_t* p) {       // 8-bit BGR2GRAY fixed point, x 48 (convertTo(CV_16S, 48))
    return (float)(((1868 * (int)p[0] + 9617 * (int)p[1] + 4899 * (int)p[2] + 8192) >> 14) * FIXPT_SCALE);
}

// XCD-aware remap: hardware places block b on XCD b % 8; give every XCD a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// compile-time row loop (the unrolled body must see its ring slot as a constant; #pragma unroll gives up on bodies this large)
template <int J, int NPP, class F> __device__ __forceinline__ bool static_rows(F& f) {
    if constexpr (J < NPP) { if (!f(std::integral_constant<int, J>{})) return false; return static_rows<J + 1, NPP>(f); }
    else return true;
}

// ---------- K1/K2: separable Gaussian, 16S -> 32F -> 16S ------------------------------------------------------------------------
// Arithmetic (oracle_sift.c gauss_blur16; OpenCV's RowFilter<short,float> + SymmColumnFilter<Cast<float,short>>), products and sums
// rounded separately (this translation unit is compiled with -ffp-contract=off; nothing below may become an fma):
//   row     t(x) = k[0] * S(x - R);  t(x) += k[i] * S(x - R + i)            for i = 1 .. 2R
//   column  s    = k[R] * t(y);      s    += k[R + j] * (t(y + j) + t(y - j)) for j = 1 .. R;   out = round-half-even(s) in 16 bits
// borders reflect-101.
struct Blur16Args {
    const lvl_t* src;                        // source level (w x h), frame f at src + f * fstride
    const uint8_t* bgr[SIFT_BATCH_MAX];      // BGR = true: the caller's frames (w x h, row stride bgr_ws[f] bytes)
    int bgr_ws[SIFT_BATCH_MAX];
    lvl_t* dst;                              // frame f at dst + f * fstride
    lvl_t* ds;                               // not NULL: also write the 2x decimated level (even rows, even columns): the next octave's base
    int w, h;
    int tiles_x, tiles_y;
    float k[2 * MAX_R + 1];
    float kp[2 * (MAX_R + 1)];               // blur16_stream: tap pairs (k[t], k[t-1]) for t = 0 .. R, k[-1] = 0
    size_t fstride;                          // samples between the frames of a batch
    int nb;                                  // frames in the launch
};

constexpr int T16W = 64, T16H = 32;
template <int R, bool BGR>
__global__ __launch_bounds__(256) void blur16_tile(Blur16Args a) {
    constexpr int ROWS = T16H + 2 * R, COLS = T16W + 2 * R, PIN = COLS | 1, PMID = T16W + 4;
    __shared__ float s_in[ROWS * PIN];
    __shared__ __attribute__((aligned(16))) float s_mid[ROWS * PMID];
    const int tid = threadIdx.x, fr = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, a.tiles_x * a.tiles_y);
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * T16W, y0 = ty * T16H;
    const lvl_t* src = BGR ? nullptr : a.src + (size_t)fr * a.fstride;
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    lvl_t* dst = a.dst + (size_t)fr * a.fstride;
    for (int idx = tid; idx < ROWS * COLS; idx += 256) {
        const int ry = idx / COLS, rx = idx - ry * COLS;
        const int gy = reflect101(y0 - R + ry, a.h), gx = reflect101(x0 - R + rx, a.w);
        s_in[ry * PIN + rx] = BGR ? gray48(bgr + (size_t)gy * bws + 3 * gx) : ld1(src + (size_t)gy * a.w + gx);
    }
    __syncthreads();
    // row pass: 4 adjacent outputs per item from a sliding window
    for (int item = tid; item < ROWS * (T16W / 4); item += 256) {
        const int xg = item / ROWS, row = item - xg * ROWS;
        const float* in = s_in + row * PIN + 4 * xg;
        float e[2 * R + 4];
#pragma unroll
        for (int m = 0; m < 2 * R + 4; m++) e[m] = in[m];
        float acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = a.k[0] * e[q];
#pragma unroll
        for (int i = 1; i <= 2 * R; i++) {
#pragma unroll
            for (int q = 0; q < 4; q++) { const float p = a.k[i] * e[q + i]; acc[q] = acc[q] + p; }
        }
        *reinterpret_cast<v4f*>(s_mid + row * PMID + 4 * xg) = (v4f){acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    // column pass: centre tap, then the symmetric pairs outwards; one output row x 4 columns per item
    for (int item = tid; item < T16H * (T16W / 4); item += 256) {
        const int yy = item / (T16W / 4), xg = item - yy * (T16W / 4);
        const float* mid = s_mid + (yy + R) * PMID + 4 * xg;
        v4f acc = *reinterpret_cast<const v4f*>(mid) * a.k[R];
#pragma unroll
        for (int j = 1; j <= R; j++) {
            const v4f sm = *reinterpret_cast<const v4f*>(mid + j * PMID) + *reinterpret_cast<const v4f*>(mid - j * PMID);
            const v4f p = sm * a.k[R + j];
            acc = acc + p;
        }
        const int gx = x0 + 4 * xg, gy = y0 + yy;
        if (gy >= a.h) continue;
        const int o0 = sat16(acc.x), o1 = sat16(acc.y), o2 = sat16(acc.z), o3 = sat16(acc.w);
        lvl_t* d = dst + (size_t)gy * a.w + gx;
        if (gx + 3 < a.w && (a.w & 3) == 0) {
            *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)(o0 & 0xffff) | ((unsigned)o1 << 16), (unsigned)(o2 & 0xffff) | ((unsigned)o3 << 16));
        } else {
            if (gx < a.w) d[0] = (lvl_t)o0;
            if (gx + 1 < a.w) d[1] = (lvl_t)o1;
            if (gx + 2 < a.w) d[2] = (lvl_t)o2;
            if (gx + 3 < a.w) d[3] = (lvl_t)o3;
        }
        if (a.ds && !(gy & 1) && (gy >> 1) < (a.h >> 1)) {
            lvl_t* q = a.ds + (size_t)fr * a.fstride + (size_t)(gy >> 1) * (a.w >> 1);
            if ((gx >> 1) < (a.w >> 1)) q[gx >> 1] = (lvl_t)o0;
            if (((gx + 2) >> 1) < (a.w >> 1)) q[(gx + 2) >> 1] = (lvl_t)o2;
        }
    }
}

// ---- blur16_stream: barrier-free streaming variant for the big levels ---------------------------------------------------------------
// Same arithmetic.  One WAVE owns a strip of 256 columns (4 per lane) and walks down L output rows of it: every input row is read
// once from HBM (prefetched two rows ahead into registers), exchanged with the neighbour lanes through a wave-private LDS row (no
// workgroup barrier anywhere: LDS operations of one wave execute in order), filtered horizontally from a rolling register window, and
// kept in a register ring of the last 2R+2 row results; an output row leaves one step after its last row has arrived (its column pass
// covers the LDS round trip of the next row's window), taps centre first, then the pairs (y + j, y - j).
// The row loop itself is hand-scheduled gfx950 assembly (blur16_asm.inc, generated by gen_blur16_asm.py, which says what the hand
// schedule does that hipcc's did not: counted vmcnt waits, tap-staggered packed products without register moves, ...).  The HIP code
// below only works out the wave's geometry.
#include "build/blur16_asm.inc"     // generated next to the objects (build.py generate()); the path is explicit so that a stale copy left in csrc/ by an older checkout can never be picked up
template <int R, bool BGR, bool DS>
__device__ __forceinline__ void blur16_stream_body(const Blur16Args& a, int L, int nstrip, int nseg) {
    constexpr int RA = (R + 3) / 4 * 4, SW = 256, BW = SW + 2 * RA;
    constexpr int BUF = (BW + 8 + 63) & ~63;          // + 8 floats: dump area for lanes that have no halo sample to write
    static_assert(2 * RA + R <= 64, "halo lanes");
    __shared__ __attribute__((aligned(16))) float s_buf[4][2][BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave-uniform by construction; readfirstlane tells the compiler, so that segment, strip and row pointers live in scalar registers.
    // Every return below is taken by whole waves (the tests are on `unit`): the assembly body runs with all 64 lanes on and leaves exec = -1
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= a.nb) return;
    unit -= fr * per;
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    if (seg >= nseg) return;
    const int x0 = strip * SW, y0 = seg * L;          // L is even (launcher): the decimated rows are the even rows of a segment
    const int lact = (a.h - y0 < L) ? a.h - y0 : L;
    const int wv = a.w - x0;                          // valid columns of this strip (> R, checked by the launcher)
    // columns: the lane's own four (clamped into the row for a partial last strip); one more sample for the lanes < 2 RA (the halos,
    // reflect-101) and, in a partial last strip, for the lanes 2 RA .. 2 RA + R - 1: the reflected columns right of the image,
    // written AFTER the main samples (LDS operations of one wave execute in order) over whatever the clamped main loads left there
    const int xm = x0 + 4 * lane;
    const int xl = xm < a.w - 4 ? xm : a.w - 4;
    int hcol, hpos;
    if (lane < RA) { hcol = x0 - RA + lane; hpos = lane; }
    else if (lane < 2 * RA) { hcol = x0 + SW + (lane - RA); hpos = SW + lane; }
    else if (wv < SW && lane < 2 * RA + R) { hcol = x0 + wv + (lane - 2 * RA); hpos = RA + wv + (lane - 2 * RA); }
    else { hcol = x0; hpos = BW + (lane & 7); }
    hcol = reflect101(hcol, a.w);
    const size_t fro = (size_t)fr * a.fstride;
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    const int rowb = BGR ? bws : 2 * a.w;             // bytes per source row
    int gy = y0 - R, dir = 1;                         // first source row of the reflect-101 walk
    if (gy < 0) { gy = -gy; dir = -1; }
    const unsigned long long rp = (unsigned long long)(BGR ? reinterpret_cast<uintptr_t>(bgr) : reinterpret_cast<uintptr_t>(a.src + fro)) + (unsigned long long)gy * (unsigned)rowb;
    const unsigned moff = BGR ? 3u * xl : 2u * xl;    // 3 xl is a multiple of 12; rows are 4-byte aligned (launcher)
    unsigned hoff, hsh = 0;
    if (BGR) {                                        // the 8 bytes inside the row that hold the halo pixel's three
        const int b0 = 3 * hcol; int st = b0 & ~3; if (st + 8 > bws) st = bws - 8;
        hoff = (unsigned)st; hsh = 8u * (unsigned)(b0 - st);
    } else hoff = 2u * hcol;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(&s_buf[wave][0][0]);
    const unsigned lds_win = lds0 + 16u * lane, lds_main = lds0 + 4u * (RA + 4 * lane), lds_halo = lds0 + 4u * hpos;
    const unsigned long long dp = (unsigned long long)reinterpret_cast<uintptr_t>(a.dst + fro) + 2ull * (unsigned long long)y0 * a.w;
    const unsigned long long dsp = DS ? (unsigned long long)reinterpret_cast<uintptr_t>(a.ds + fro) + 2ull * (unsigned long long)(y0 >> 1) * (a.w >> 1) : 0ull;
    const unsigned doff = 2u * xm, dsoff = 2u * (xm >> 1);
    const unsigned long long smask = __ballot(xm < a.w);                 // lanes that store (partial last strip)
    const unsigned long long kp = (unsigned long long)reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(Blur16Args, kp);
    const int n = lact + 2 * R + 1;                   // steps: 2R + 1 rows fill the ring, then one output row per step
    const int hm1 = a.h - 1, dstr = 2 * a.w;
    (void)hsh;
#define BLUR_ASM_CALL(NAME) NAME(lds_win, lds_main, lds_halo, moff, hoff, doff, dsoff, rp, dp, dsp, kp, smask, rowb, gy, dir, hm1, dstr, n)
    if constexpr (BGR) { static_assert(R == 6, "base level"); blur16_asm_r6_bgr(lds_win, lds_main, lds_halo, moff, hoff, doff, dsoff, hsh, rp, dp, dsp, kp, smask, rowb, gy, dir, hm1, dstr, n); }
    else if constexpr (R == 5) BLUR_ASM_CALL(blur16_asm_r5);
    else if constexpr (R == 6) BLUR_ASM_CALL(blur16_asm_r6);
    else if constexpr (R == 8 && DS) BLUR_ASM_CALL(blur16_asm_r8_ds);
    else if constexpr (R == 8) BLUR_ASM_CALL(blur16_asm_r8);
    else if constexpr (R == 10) BLUR_ASM_CALL(blur16_asm_r10);
    else if constexpr (R == 13) BLUR_ASM_CALL(blur16_asm_r13);
#undef BLUR_ASM_CALL
}
// W waves per SIMD: the launcher sizes the grid to exactly W x 1024 waves, and the cap keeps the dispatcher from stacking them unevenly
template <int R, bool BGR, bool DS, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void blur16_stream(Blur16Args a, int L, int nstrip, int nseg) {
    blur16_stream_body<R, BGR, DS>(a, L, nstrip, nseg);
}

// next octave seed: every second pixel of level 3 (cv::resize INTER_NEAREST to half size)
__global__ __launch_bounds__(256) void downsample16(const lvl_t* src, int sw, lvl_t* dst, int dw, int dh, size_t fstride) {
    src += (size_t)blockIdx.z * fstride; dst += (size_t)blockIdx.z * fstride;      // blockIdx.z = frame of the batch
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x < dw && y < dh) dst[(size_t)y * dw + x] = src[(size_t)(2 * y) * sw + 2 * x];
}

// ---------- K3: DoG extrema -------------------------------------------------------------------------------------
struct OctaveDev { lvl_t* lv[N_LEVELS]; int w, h; };

#ifndef EXT_EH
#define EXT_EH 16
#endif
constexpr int EW = 64, EH = EXT_EH, ECAP = 128;
#ifndef REG_SHIFT_V
#define REG_SHIFT_V 0
#endif
constexpr int REG_SHIFT = REG_SHIFT_V;      // 2^REG_SHIFT consecutive tiles append to the same region: refine_kernel then walks spatially coherent runs
constexpr int NREG = 64, REG_STRIDE = 32;   // candidate list split into 64 regions, one counter per 128-byte line:
                                            // a single counter caps at ~1e8 returning atomics/s (one per tile = 0.5 ms)
// batched launches: frame f = blockIdx.y (z for refine) works on its own copy of every buffer, a fixed stride apart
struct BatchStride { size_t pyr, claimed, cand, refined, kps, cube, sel, mins; };   // elements of the respective type (sel: selected keypoints per frame; mins: keep-all's start-key table)
constexpr size_t CNT_STRIDE = 64, CCNT_STRIDE = (size_t)64 * 32, SEL_STRIDE = 2048;
constexpr int KEEPALL_MAX = MI355_SIFT_KEEPALL_MAX;      // keypoints per frame with nfeatures <= 0 (cv::SIFT's "keep all")
struct FrameOuts { mi355_keypoint* kp[SIFT_BATCH_MAX]; uint8_t* d8[SIFT_BATCH_MAX]; };

// ---------- K3b: sub-pixel refinement ---------------------------------------------------------------------------
struct Refined { int o, layer, r, c; float xi, xr, xc, contr, scl; unsigned start; };      // start: keep-all: (layer, row, column) of the extremum the fit started from, packed 2 + 14 + 14 bits

struct PyrDev { OctaveDev oc[MAX_OCT]; unsigned* claimed[MAX_OCT]; unsigned* mins[MAX_OCT]; int n_oct; };      // mins: keep-all only (one word per claim bit)

// DoG[lvl] = G[lvl + 1] - G[lvl]: 16-bit integers (cv::subtract into CV_16S; the samples are 0 .. 12240, nothing saturates), exact as float
__device__ __forceinline__ float dogv(const OctaveDev& oc, size_t foff, int lvl, int r, int c) {
    const size_t o = foff + (size_t)r * oc.w + c;
    return (float)((int)oc.lv[lvl + 1][o] - (int)oc.lv[lvl][o]);
}

// Matx33f::solve(b, DECOMP_LU) = Matx_FastSolveOp<float, 3, 1>: Cramer's rule in float, the expression of the vendored
// core/operations.hpp:742-750, 882-903 (products and sums rounded separately); det == 0 -> x = 0
__device__ __forceinline__ void solve3(const float a[3][3], const float b[3], float x[3]) {
    const float det = (a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2])) + a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (det == 0.0f) { x[0] = x[1] = x[2] = 0.0f; return; }
    const float d = 1.0f / det;
    x[0] = d * ((b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2])) + a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * ((a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0])) + a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * ((a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0])) + b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

// One Newton step of the quadratic fit at (L, R, C) and the acceptance tests, written once over an accessor dv(L, R, C)
// so that the in-tile first step (DoG planes in LDS, extrema_kernel) and the iterative path (global memory,
// refine_kernel) evaluate the very same expressions.
struct FitOff { float xi, xr, xc; };
template <class DV>
__device__ __forceinline__ FitOff fit_step(DV dv, int L, int R, int C) {
    const float img_scale = 1.0f / (float)(255 * FIXPT_SCALE);
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    float dD[3];
    dD[0] = (dv(L, R, C + 1) - dv(L, R, C - 1)) * deriv_scale;
    dD[1] = (dv(L, R + 1, C) - dv(L, R - 1, C)) * deriv_scale;
    dD[2] = (dv(L + 1, R, C) - dv(L - 1, R, C)) * deriv_scale;
    const float v2 = dv(L, R, C) * 2.0f;
    const float dxx = (dv(L, R, C + 1) + dv(L, R, C - 1) - v2) * second_scale;
    const float dyy = (dv(L, R + 1, C) + dv(L, R - 1, C) - v2) * second_scale;
    const float dss = (dv(L + 1, R, C) + dv(L - 1, R, C) - v2) * second_scale;
    const float dxy = (dv(L, R + 1, C + 1) - dv(L, R + 1, C - 1) - dv(L, R - 1, C + 1) + dv(L, R - 1, C - 1)) * cross_scale;
    const float dxs = (dv(L + 1, R, C + 1) - dv(L + 1, R, C - 1) - dv(L - 1, R, C + 1) + dv(L - 1, R, C - 1)) * cross_scale;
    const float dys = (dv(L + 1, R + 1, C) - dv(L + 1, R - 1, C) - dv(L - 1, R + 1, C) + dv(L - 1, R - 1, C)) * cross_scale;
    const float A[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    float X[3];
    solve3(A, dD, X);
    FitOff f; f.xi = -X[2]; f.xr = -X[1]; f.xc = -X[0];
    return f;
}
// contrast and edge tests at the converged location; contr is the interpolated response
template <class DV>
__device__ __forceinline__ bool fit_accept(DV dv, int L, int R, int C, FitOff f, float contrast_thr, float edge_thr, float& contr) {
    const float img_scale = 1.0f / (float)(255 * FIXPT_SCALE);
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    float dD0 = (dv(L, R, C + 1) - dv(L, R, C - 1)) * deriv_scale;
    float dD1 = (dv(L, R + 1, C) - dv(L, R - 1, C)) * deriv_scale;
    float dD2 = (dv(L + 1, R, C) - dv(L - 1, R, C)) * deriv_scale;
    const float t = ((0.0f + dD0 * f.xc) + dD1 * f.xr) + dD2 * f.xi;          // Matx::dot: s = 0; s += a[i] * b[i]
    contr = dv(L, R, C) * img_scale + t * 0.5f;
    if (fabsf(contr) * (float)N_LAYERS < contrast_thr) return false;
    const float v2 = dv(L, R, C) * 2.0f;
    const float dxx = (dv(L, R, C + 1) + dv(L, R, C - 1) - v2) * second_scale;
    const float dyy = (dv(L, R + 1, C) + dv(L, R - 1, C) - v2) * second_scale;
    const float dxy = (dv(L, R + 1, C + 1) - dv(L, R + 1, C - 1) - dv(L, R - 1, C + 1) + dv(L, R - 1, C - 1)) * cross_scale;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0.0f || (tr * tr) * edge_thr >= ((edge_thr + 1.0f) * (edge_thr + 1.0f)) * det) return false;
    return true;
}
// 0: converged, 1: dead (diverged / NaN), 2: keep iterating
__device__ __forceinline__ int fit_state(FitOff f) {
    if (fabsf(f.xi) < 0.5f && fabsf(f.xr) < 0.5f && fabsf(f.xc) < 0.5f) return 0;
    if (fabsf(f.xi) > 7.0e8f || fabsf(f.xr) > 7.0e8f || fabsf(f.xc) > 7.0e8f) return 1;
    if (!(f.xi == f.xi) || !(f.xr == f.xr) || !(f.xc == f.xc)) return 1;
    return 2;
}

__global__ __launch_bounds__(256) void extrema_kernel(OctaveDev oc, int octave, unsigned long long* cand, unsigned* count, unsigned cap, unsigned* overflow, BatchStride bs,
                                                      float* cube, unsigned cube_cap) {
    {
        const size_t f = blockIdx.y;
#pragma unroll
        for (int l = 0; l < N_LEVELS; l++) oc.lv[l] += f * bs.pyr;
        cand += f * bs.cand; count += f * CCNT_STRIDE; overflow += f * CNT_STRIDE; cube += f * bs.cube;
    }
    // staged window: rows y0-1 .. y0+EH, columns x0-4 .. x0+EW+3 (16-byte aligned so that interior tiles load float4)
    constexpr int PC = EW + 8;                 // 72 columns
    constexpr int P4 = PC / 4;                 // 18 float4 per row
    constexpr int RW = EH + 2;                 // 18 rows
    __shared__ v4f s_d4[5][RW * P4];
    __shared__ unsigned long long s_list[ECAP];   // candidates of this tile: one global atomic per workgroup
    __shared__ unsigned s_n, s_base;
    const int tid = threadIdx.x;
    // XCD-aware tile order: consecutive tiles (horizontal neighbours, which share their halo columns' cache lines) run on
    // the same XCD, i.e. behind the same L2 -- with the default round-robin every halo line came from HBM again
    // (measured: 2.2x the algorithmic bytes at the fabric).
    const int tiles_x = (oc.w + EW - 1) / EW, tiles_y = (oc.h + EH - 1) / EH;
    const int tile = xcd_remap(blockIdx.x, tiles_x * tiles_y);
    const int tyy = tile / tiles_x, txx = tile - tyy * tiles_x;
    const int x0 = txx * EW, y0 = tyy * EH;
    if (tid == 0) s_n = 0;
    const bool vec = ((oc.w & 3) == 0) && (x0 - 4 >= 0) && (x0 + EW + 4 <= oc.w) && ((reinterpret_cast<uintptr_t>(oc.lv[0]) & 7) == 0);
    if (vec) {
        // centre columns x0 .. x0+EW-1: whole 256-byte runs (two 128-byte lines per row and level), float4 per lane
        for (int idx = tid; idx < RW * (EW / 4); idx += 256) {
            const int ry = idx / (EW / 4), c4 = idx - ry * (EW / 4);
            int gy = y0 - 1 + ry;
            gy = gy < 0 ? 0 : (gy > oc.h - 1 ? oc.h - 1 : gy);      // clamped halo rows are never used by valid pixels (5 px border)
            const size_t o = (size_t)gy * oc.w + (x0 + 4 * c4);
            v4f g[N_LEVELS];
#pragma unroll
            for (int l = 0; l < N_LEVELS; l++) g[l] = ld4(oc.lv[l] + o);
#pragma unroll
            for (int l = 0; l < 5; l++) s_d4[l][ry * P4 + 1 + c4] = g[l + 1] - g[l];
        }
        // the two halo columns x0-1 and x0+EW (lines the neighbouring tiles fetch anyway)
        if (tid < 2 * RW) {
            const int ry = tid >> 1, side = tid & 1;
            int gy = y0 - 1 + ry;
            gy = gy < 0 ? 0 : (gy > oc.h - 1 ? oc.h - 1 : gy);
            const size_t o = (size_t)gy * oc.w + (side ? x0 + EW : x0 - 1);
            float g[N_LEVELS];
#pragma unroll
            for (int l = 0; l < N_LEVELS; l++) g[l] = ld1(oc.lv[l] + o);
#pragma unroll
            for (int l = 0; l < 5; l++) reinterpret_cast<float*>(s_d4[l])[ry * PC + (side ? 4 + EW : 3)] = g[l + 1] - g[l];
        }
    } else {
        for (int idx = tid; idx < RW * PC; idx += 256) {
            const int ry = idx / PC, rx = idx - ry * PC;
            int gy = y0 - 1 + ry, gx = x0 - 4 + rx;
            gy = gy < 0 ? 0 : (gy > oc.h - 1 ? oc.h - 1 : gy);
            gx = gx < 0 ? 0 : (gx > oc.w - 1 ? oc.w - 1 : gx);
            const size_t o = (size_t)gy * oc.w + gx;
            float g[N_LEVELS];
#pragma unroll
            for (int l = 0; l < N_LEVELS; l++) g[l] = ld1(oc.lv[l] + o);
#pragma unroll
            for (int l = 0; l < 5; l++) reinterpret_cast<float*>(s_d4[l])[idx] = g[l + 1] - g[l];
        }
    }
    __syncthreads();
    // one item per lane: 4 adjacent pixels of one row.  All 5 DoG planes' 3x6 windows go to registers (one 128-bit and two
    // 32-bit LDS reads per row), column-wise 3-row max/min are shared by the 4 pixels, and the 26-neighbour test becomes
    // "val >= max of the neighbours" / "val <= min of the neighbours" without data-dependent branches.
    for (int item = tid; item < EH * (EW / 4); item += 256) {
        const int ly = item / (EW / 4), xg = item - ly * (EW / 4);
        const int r = y0 + ly;
        float cm[5][6], cn[5][6], mid[5][6], top[5][6], bot[5][6];
#pragma unroll
        for (int pl = 0; pl < 5; pl++) {
            const float* base = reinterpret_cast<const float*>(s_d4[pl]) + (ly + 1) * PC + 4 * xg + 4;
#pragma unroll
            for (int rr = -1; rr <= 1; rr++) {
                const float* q = base + rr * PC;
                const v4f c4 = *reinterpret_cast<const v4f*>(q);
                float v[6] = {q[-1], c4.x, c4.y, c4.z, c4.w, q[4]};
#pragma unroll
                for (int j = 0; j < 6; j++) { if (rr == -1) top[pl][j] = v[j]; else if (rr == 0) mid[pl][j] = v[j]; else bot[pl][j] = v[j]; }
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                cm[pl][j] = fmaxf(fmaxf(top[pl][j], mid[pl][j]), bot[pl][j]);
                cn[pl][j] = fminf(fminf(top[pl][j], mid[pl][j]), bot[pl][j]);
            }
        }
        // all 12 tests of the item without a branch (same exact predicate as extrema_stream: |val| >= max(val > 0 ? mx : -mn, 21), DoG values are integers);
        // the rare hits are emitted afterwards
        unsigned hit = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = k + 1;
#pragma unroll
            for (int layer = 1; layer <= N_LAYERS; layer++) {
                const float val = mid[layer][j];
                // neighbours: full 3x3 of the two adjacent planes + the 8 in-plane ones
                float mx = fmaxf(fmaxf(cm[layer - 1][j - 1], cm[layer - 1][j]), cm[layer - 1][j + 1]);
                mx = fmaxf(mx, fmaxf(fmaxf(cm[layer + 1][j - 1], cm[layer + 1][j]), cm[layer + 1][j + 1]));
                mx = fmaxf(mx, fmaxf(fmaxf(cm[layer][j - 1], cm[layer][j + 1]), fmaxf(top[layer][j], bot[layer][j])));
                float mn = fminf(fminf(cn[layer - 1][j - 1], cn[layer - 1][j]), cn[layer - 1][j + 1]);
                mn = fminf(mn, fminf(fminf(cn[layer + 1][j - 1], cn[layer + 1][j]), cn[layer + 1][j + 1]));
                mn = fminf(mn, fminf(fminf(cn[layer][j - 1], cn[layer][j + 1]), fminf(top[layer][j], bot[layer][j])));
                const float q = val > 0.0f ? mx : -mn;
                hit |= fabsf(val) >= fmaxf(q, DOG_THRESHOLD_P1) ? (1u << ((layer - 1) * 4 + k)) : 0u;
            }
        }
        unsigned cmask = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int c = x0 + 4 * xg + k; cmask |= (c >= IMG_BORDER && c < oc.w - IMG_BORDER) ? (0x111u << k) : 0u; }
        hit = (r >= IMG_BORDER && r < oc.h - IMG_BORDER) ? (hit & cmask) : 0u;
        while (hit) {
            const int bb = __builtin_ctz(hit); hit &= hit - 1;
            const int layer = bb / 4 + 1, c = x0 + 4 * xg + (bb & 3);
            const unsigned long long rec = ((unsigned long long)octave << 48) | ((unsigned long long)layer << 40) | ((unsigned long long)r << 20) | (unsigned long long)c;
            const unsigned slot = atomicAdd(&s_n, 1u);
            if (slot < ECAP) s_list[slot] = rec;
            else {                                                                           // tile with > 128 extrema (flat image)
                const unsigned reg = ((unsigned)tile >> REG_SHIFT) & (NREG - 1);
                const unsigned g = atomicAdd(&count[reg * REG_STRIDE], 1u);
                if (g < cap) cand[(size_t)reg * cap + g] = rec; else *overflow = 1;
            }
        }
    }
    __syncthreads();
    const unsigned nloc = s_n < ECAP ? s_n : ECAP;
    const unsigned reg = ((unsigned)tile >> REG_SHIFT) & (NREG - 1);
    if (tid == 0 && nloc) s_base = atomicAdd(&count[reg * REG_STRIDE], nloc);
    __syncthreads();
    // Each candidate also leaves with its 3x3x3 DoG neighbourhood (27 floats in a 128-byte record, straight from the LDS
    // planes): the first Newton step of refine_kernel -- the only one for two candidates out of three -- then reads one line
    // instead of twelve cold ones scattered over four pyramid levels.  Bit 63 of the record says the cube exists.
    for (unsigned i = tid; i < nloc; i += 256) {
        const unsigned g = s_base + i;
        if (g < cap) cand[(size_t)reg * cap + g] = s_list[i] | (g < cube_cap ? (1ull << 63) : 0ull); else *overflow = 1;
    }
    for (unsigned idx = tid; idx < nloc * 27; idx += 256) {
        const unsigned ci = idx / 27, e = idx - ci * 27;
        const unsigned g = s_base + ci;
        if (g >= cube_cap) continue;
        const unsigned long long rec = s_list[ci];
        const int L0 = (int)((rec >> 40) & 0xff), R0 = (int)((rec >> 20) & 0xfffff), C0 = (int)(rec & 0xfffff);
        const int dl = (int)(e / 9) - 1, dr = (int)((e / 3) % 3) - 1, dc = (int)(e % 3) - 1;
        cube[((size_t)reg * 32 + e) * cube_cap + g] = reinterpret_cast<const float*>(s_d4[L0 + dl])[(R0 + dr - y0 + 1) * PC + (C0 + dc - x0) + 4];
    }
}

// ---- extrema_stream: the same test, streamed -------------------------------------------------------------------
// One WAVE owns a strip of 256 columns (4 per lane) and walks down its rows: every row of the six levels is read once
// (XD rows are in flight while the current one is processed), the five DoG planes of three consecutive rows live in
// registers, the neighbours' columns come through DPP wave shifts, and nothing waits on a workgroup barrier.  The hot
// loop issues no memory operation besides those row loads: a candidate (rare) is parked in a wave-private LDS list
// together with its 3x3x3 DoG neighbourhood, taken from the registers of the lane and of its two neighbours; the list
// goes to global memory when it is full and at the end of the segment.
constexpr int XCAP = 512;                       // candidate records buffered per wave (4 KB); the list is emptied when half full, between runs of the row loop
#ifndef MI355_XD
#define MI355_XD 2
#endif
constexpr int XD = MI355_XD;                           // rows in flight per wave
#ifndef MI355_XWAVES
#define MI355_XWAVES 3
#endif
constexpr int XWAVES = MI355_XWAVES;                       // waves per SIMD the streamed test is compiled for (the launcher sizes its grid to whole rounds of them)
constexpr int XSW = 248;                        // columns a wave is responsible for: lanes 1..62; lanes 0 and 63 carry the neighbours' columns
// The DoG values are 16-bit integers and stay PACKED, two columns per register, from the level rows to the test: v_pk_sub_i16 forms
// them straight from the loaded level words (no unpacking), v_pk_max_i16 / v_pk_min_i16 take the extremes of two columns at a time and
// the 16-bit compares read either half (SDWA).  Half the registers of the float form of round 3 (4 waves per SIMD instead of 2) and
// three quarters of its instructions.
typedef short s2 __attribute__((ext_vector_type(2)));
struct XRow { uint2 m[N_LEVELS]; };             // one row of the six levels: 4 packed 16-bit pixels per lane
struct XDog { s2 m[5][2]; };                    // five DoG planes: columns (0,1) and (2,3) of the lane
__device__ __forceinline__ s2 as_s2(unsigned v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ unsigned as_u(s2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pmin(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
// lane l gets v of lane l -+ 1; the end lane gets 0 (bound_ctrl): lanes 0 and 63 only carry the neighbours' columns, what they compute
// is never used, and without an 'old' value no v_mov has to precede the DPP move
__device__ __forceinline__ unsigned dpp_lower(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned dpp_upper(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }

// 26-neighbour test of the centre row B (rows A above, C below): bit (layer-1)*4 + k set for an extremum at column k.
// val >= every one of its 26 neighbours  <=>  val == the maximum of the 3x3x3 block around it (the block holds val itself), and that
// maximum is separable: the 3-row column extremes of a plane, their 3-wide horizontal extremes (neighbour lanes through DPP, the
// shifted column pairs through v_alignbit), the extremes of three adjacent planes.
// |val| > 20 && ((val > 0 && val >= all) || (val < 0 && val <= all))  <=>  val == max(M27, 21) || val == min(m27, -21)  (val <= M27 always).
__device__ __forceinline__ unsigned xtest_row(const XDog& A, const XDog& B, const XDog& C, int xm, int clo, int chi, bool row_ok) {
    s2 hx[5][2], hn[5][2];                         // extremes of the 3x3 block of every plane around the lane's columns (0,1) and (2,3)
#pragma unroll
    for (int p = 0; p < 5; p++) {
        auto h3 = [](s2 c01, s2 c23, bool mx, s2& o01, s2& o23) {
            const unsigned u01 = as_u(c01), u23 = as_u(c23);
            const unsigned l23 = dpp_lower(u23), r01 = dpp_upper(u01);                    // the left lane's columns (2,3), the right lane's (0,1)
            const s2 s_m0 = as_s2(__builtin_amdgcn_alignbit(u01, l23, 16));               // columns (-1, 0)
            const s2 s_12 = as_s2(__builtin_amdgcn_alignbit(u23, u01, 16));               // columns ( 1, 2)
            const s2 s_34 = as_s2(__builtin_amdgcn_alignbit(r01, u23, 16));               // columns ( 3, 4)
            if (mx) { o01 = pmax(pmax(s_m0, c01), s_12); o23 = pmax(pmax(s_12, c23), s_34); }
            else    { o01 = pmin(pmin(s_m0, c01), s_12); o23 = pmin(pmin(s_12, c23), s_34); }
        };
        h3(pmax(pmax(A.m[p][0], B.m[p][0]), C.m[p][0]), pmax(pmax(A.m[p][1], B.m[p][1]), C.m[p][1]), true, hx[p][0], hx[p][1]);
        h3(pmin(pmin(A.m[p][0], B.m[p][0]), C.m[p][0]), pmin(pmin(A.m[p][1], B.m[p][1]), C.m[p][1]), false, hn[p][0], hn[p][1]);
    }
    unsigned hit = 0;
    const s2 t21 = {21, 21}, tm21 = {-21, -21};
#pragma unroll
    for (int layer = 1; layer <= N_LAYERS; layer++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const s2 val = B.m[layer][r];
            const s2 mx = pmax(pmax(pmax(hx[layer - 1][r], hx[layer][r]), hx[layer + 1][r]), t21);
            const s2 mn = pmin(pmin(pmin(hn[layer - 1][r], hn[layer][r]), hn[layer + 1][r]), tm21);
            const bool h0 = (val.x == mx.x) | (val.x == mn.x), h1 = (val.y == mx.y) | (val.y == mn.y);      // lane masks: the | is scalar
            hit = (hit << 2) | (h0 ? 2u : 0u) | (h1 ? 1u : 0u);                                             // bit 11 - ((layer-1)*4 + k)
        }
    }
    if (!row_ok || hit == 0) return 0u;
    // (rare per lane) back to bit (layer-1)*4 + k, columns outside the strip / the border masked
    unsigned out = 0;
#pragma unroll
    for (int b = 0; b < 12; b++) out |= ((hit >> (11 - b)) & 1u) << b;
    unsigned cmask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) cmask |= (xm + k >= clo && xm + k < chi) ? (0x111u << k) : 0u;
    return out & cmask;
}

// The row loop's form of the same test, horizontal extremes first: the 3-wide extremes of a row's planes are formed ONCE, when the row enters
// the window (maximum and minimum share the neighbour lanes' columns and the shifted pairs), and the test of a centre row takes the extremes of
// its three rows.  Maxima commute, so the 3x3x3 extremes are the same numbers as xtest_row's (rows first); 105 instead of 130 instructions per
// row step for the block extremes, at 48 more registers (the kernel has them: three waves per SIMD).
struct XH { s2 hx[5][2], hn[5][2]; };           // per plane: 3-wide maxima / minima around the lane's columns (0,1) and (2,3)
struct XCen { s2 v[3][2]; };                    // the row's own values of the three tested planes
__device__ __forceinline__ void xrow_extremes(const XDog& d, XH& h, XCen& c) {
#pragma unroll
    for (int p = 0; p < 5; p++) {
        const unsigned u01 = as_u(d.m[p][0]), u23 = as_u(d.m[p][1]);
        const unsigned l23 = dpp_lower(u23), r01 = dpp_upper(u01);
        const s2 s_m0 = as_s2(__builtin_amdgcn_alignbit(u01, l23, 16));
        const s2 s_12 = as_s2(__builtin_amdgcn_alignbit(u23, u01, 16));
        const s2 s_34 = as_s2(__builtin_amdgcn_alignbit(r01, u23, 16));
        h.hx[p][0] = pmax(pmax(s_m0, d.m[p][0]), s_12); h.hx[p][1] = pmax(pmax(s_12, d.m[p][1]), s_34);
        h.hn[p][0] = pmin(pmin(s_m0, d.m[p][0]), s_12); h.hn[p][1] = pmin(pmin(s_12, d.m[p][1]), s_34);
    }
#pragma unroll
    for (int l = 0; l < 3; l++) { c.v[l][0] = d.m[l + 1][0]; c.v[l][1] = d.m[l + 1][1]; }
}
__device__ __forceinline__ unsigned xtest_rows(const XH& A, const XH& B, const XH& C, const XCen& cen, int xm, int clo, int chi, bool row_ok) {
    s2 hx[5][2], hn[5][2];
#pragma unroll
    for (int p = 0; p < 5; p++)
#pragma unroll
        for (int r = 0; r < 2; r++) { hx[p][r] = pmax(pmax(A.hx[p][r], B.hx[p][r]), C.hx[p][r]); hn[p][r] = pmin(pmin(A.hn[p][r], B.hn[p][r]), C.hn[p][r]); }
    unsigned hit = 0;
    const s2 t21 = {21, 21}, tm21 = {-21, -21};
#pragma unroll
    for (int layer = 1; layer <= N_LAYERS; layer++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const s2 val = cen.v[layer - 1][r];
            const s2 mx = pmax(pmax(pmax(hx[layer - 1][r], hx[layer][r]), hx[layer + 1][r]), t21);
            const s2 mn = pmin(pmin(pmin(hn[layer - 1][r], hn[layer][r]), hn[layer + 1][r]), tm21);
            const bool h0 = (val.x == mx.x) | (val.x == mn.x), h1 = (val.y == mx.y) | (val.y == mn.y);
            hit = (hit << 2) | (h0 ? 2u : 0u) | (h1 ? 1u : 0u);
        }
    }
    if (!row_ok || hit == 0) return 0u;
    unsigned out = 0;
#pragma unroll
    for (int b = 0; b < 12; b++) out |= ((hit >> (11 - b)) & 1u) << b;
    unsigned cmask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) cmask |= (xm + k >= clo && xm + k < chi) ? (0x111u << k) : 0u;
    return out & cmask;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XWAVES, XWAVES)))
void extrema_stream(OctaveDev oc, int octave, unsigned long long* cand, unsigned* count, unsigned cap, unsigned* overflow, BatchStride bs,
                    float* cube, unsigned cube_cap, int L, int nstrip, int nseg, int nb, int xsw /* columns per strip: multiple of 4, <= XSW */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave-uniform by construction; readfirstlane tells the compiler, so that rows, segments and their branches live in scalar registers
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= nb) return;
    unit -= fr * per;
    {
        const size_t f = (size_t)fr;
#pragma unroll
        for (int l = 0; l < N_LEVELS; l++) oc.lv[l] += f * bs.pyr;
        cand += f * bs.cand; count += f * CCNT_STRIDE; overflow += f * CNT_STRIDE; cube += f * bs.cube;
    }
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    const int xs = strip * xsw, y0 = seg * L;       // the strip's own columns are [xs, xs + xsw)
    const int lact = (oc.h - y0 < L) ? oc.h - y0 : L;
    const unsigned reg = (unsigned)unit & (NREG - 1);
    const int xm = xs - 4 + 4 * lane;               // lane 0: the four columns left of the strip, lane 63: the four right of it
    const int xl = xm < 0 ? 0 : (xm < oc.w - 4 ? xm : oc.w - 4);
    int clo = xs > IMG_BORDER ? xs : IMG_BORDER, chi = xs + xsw < oc.w - IMG_BORDER ? xs + xsw : oc.w - IMG_BORDER;
    if (lane == 0 || lane == 63) { clo = 0; chi = 0; }
    const int hm1 = oc.h - 1;
    auto load_row = [&](int r, XRow& q) {
        r = r < 0 ? 0 : (r > hm1 ? hm1 : r);                      // clamped rows are only ever neighbours of border pixels
        const size_t o = (size_t)r * oc.w + xl;
#pragma unroll
        for (int l = 0; l < N_LEVELS; l++) q.m[l] = *reinterpret_cast<const uint2*>(oc.lv[l] + o);
    };
    auto to_dog = [&](const XRow& q, XDog& d) {     // 16-bit differences of levels in [0, 255 * 48]: no overflow
#pragma unroll
        for (int p = 0; p < 5; p++) { d.m[p][0] = as_s2(q.m[p + 1].x) - as_s2(q.m[p].x); d.m[p][1] = as_s2(q.m[p + 1].y) - as_s2(q.m[p].y); }
    };
    // Candidates are rare per lane but not per wave (a 12 MP frame has ~2 per wave-row): a hit only appends its 8-byte record to a
    // wave-private LDS list (slots from a ballot, the list length is wave-uniform and lives in a scalar); the list goes to the region's
    // candidate list behind one region-counter atomic per flush.  The 3x3x3 DoG neighbourhoods refine starts from are gathered by
    // refine_kernel itself, one lane per candidate: gathered here (round 3), every flush stalled its wave for the round trip of
    // 36 scattered loads of rows that had long left the L2 -- a third of this kernel's time (73 -> 49 us per 12 MP frame without).
    __shared__ unsigned long long s_rec[4][XCAP];
    int nq = 0;                                      // wave-uniform
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    auto flush = [&]() {
        if (nq == 0) return;
        wave_sync();
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&count[reg * REG_STRIDE], (unsigned)nq);
        base = __shfl(base, 0);
        for (int idx = lane; idx < nq; idx += 64) {
            const unsigned g = base + (unsigned)idx;
            if (g >= cap) { *overflow = 1; continue; }
            cand[(size_t)reg * cap + g] = s_rec[wave][idx];                                             // bit 63 clear: refine_kernel gathers the neighbourhood itself
        }
        wave_sync();
        nq = 0;
    };
    // emit<IN_LOOP>: the row loop's form never touches global memory (false = the list is full, nothing of this row was kept); the slow
    // loop's form empties the list on the spot
    auto emit = [&](unsigned hit, int rc, auto in_loop) -> bool {
        constexpr bool IN_LOOP = decltype(in_loop)::value;
        const int nq0 = nq;
        while (__builtin_amdgcn_ballot_w64(hit != 0)) {
            const bool mine = hit != 0;
            const int bb = mine ? __builtin_ctz(hit) : 0;
            hit &= hit - 1;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            const int add = __builtin_popcountll(m);
            if (nq + add > XCAP) {
                if constexpr (IN_LOOP) { nq = nq0; return false; }
                else flush();
            }
            if (mine) {
                const int layer = bb / 4 + 1, c = xm + (bb & 3);
                s_rec[wave][nq + __builtin_popcountll(m & ((1ull << lane) - 1ull))] =
                    ((unsigned long long)octave << 48) | ((unsigned long long)layer << 40) | ((unsigned long long)rc << 20) | (unsigned long long)c;
            }
            nq += add;
        }
        return true;
    };
    // rows t-1 and t as row extremes, rows t+1 .. t+XD in flight
    XH win[3];
    XCen cen[3];
    XRow nxt[XD];
    {
        XRow q; XDog dq;
        load_row(y0 - 1, q); to_dog(q, dq); xrow_extremes(dq, win[0], cen[0]);
        load_row(y0, q); to_dog(q, dq); xrow_extremes(dq, win[1], cen[1]);
#pragma unroll
        for (int d = 0; d < XD; d++) load_row(y0 + 1 + d, nxt[d]);
    }
    // The row loop holds NO memory operation besides its row loads: the candidate list is emptied between runs of the loop, never
    // inside (a store or the returning atomic of flush() anywhere in the loop body made the compiler wait for ALL outstanding loads at
    // the top of every row, the prefetched rows included: 73 -> 55 us per 12 MP frame).  A run ends when the list is half full; a
    // group of rows that would overflow it (> 256 extrema in 6 rows of 248 columns: noise, not photographs) sends the wave into the
    // plain loop below for the rest of its segment.
    int tb = 0, t_slow = -1;
    for (bool done = false; !done;) {
        for (;;) {
            if (tb >= lact) { done = true; break; }
            if (nq > XCAP / 2) break;
            auto step = [&](auto jc) -> bool {
                constexpr int j = decltype(jc)::value;
                const int tt = tb + j;
                if (tt >= lact) return false;
                const int rc = y0 + tt;
                XH& A = win[j % 3]; XH& B = win[(j + 1) % 3]; XH& Cc = win[(j + 2) % 3];
                { XDog dq; to_dog(nxt[j % XD], dq); xrow_extremes(dq, Cc, cen[(j + 2) % 3]); }
                load_row(rc + 1 + XD, nxt[j % XD]);           // the bottom row of XD steps ahead, in flight meanwhile
                const unsigned hit = xtest_rows(A, B, Cc, cen[(j + 1) % 3], xm, clo, chi, rc >= IMG_BORDER && rc < oc.h - IMG_BORDER);
                if (!emit(hit, rc, std::true_type{})) { t_slow = tt; return false; }
                return true;
            };
            if (!static_rows<0, 3 * XD>(step)) { done = true; break; }
            tb += 3 * XD;
        }
        flush();
    }
    if (t_slow >= 0) {
        for (int tt = t_slow; tt < lact; tt++) {
            const int rc = y0 + tt;
            XRow q; XDog A, B, Cc;
            load_row(rc - 1, q); to_dog(q, A);
            load_row(rc, q); to_dog(q, B);
            load_row(rc + 1, q); to_dog(q, Cc);
            emit(xtest_row(A, B, Cc, xm, clo, chi, rc >= IMG_BORDER && rc < oc.h - IMG_BORDER), rc, std::false_type{});
        }
        flush();
    }
}

__global__ __launch_bounds__(256) void refine_kernel(PyrDev P, const unsigned long long* cand_all, const unsigned* cand_counts, unsigned cand_cap, unsigned* cand_total,
                                                     float contrast_thr, float edge_thr, float sigma,
                                                     Refined* out, unsigned* out_count, unsigned out_cap, unsigned* out_resp, BatchStride bs,
                                                     const float* cube_all, unsigned cube_cap) {
    const size_t fr = blockIdx.z, foff = fr * bs.pyr;             // frame of the batch
    cube_all += fr * bs.cube;
    cand_all += fr * bs.cand; cand_counts += fr * CCNT_STRIDE; cand_total += fr * CNT_STRIDE;
    out += fr * bs.refined; out_count += fr * CNT_STRIDE; out_resp += fr * bs.refined;
    // blockIdx.y = region of the candidate list (extrema_kernel spreads its appends over NREG counters)
    const unsigned reg = blockIdx.y;
    unsigned n = cand_counts[reg * REG_STRIDE];
    if (n > cand_cap) n = cand_cap;
    const unsigned long long* cand = cand_all + (size_t)reg * cand_cap;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {      // total for the host: one region per lane
        unsigned tot = cand_counts[threadIdx.x * REG_STRIDE];
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (threadIdx.x == 0) cand_total[0] = tot;
    }
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long pk = cand[i];
        const int o = (int)((pk >> 48) & 0xff);
        int L = (int)((pk >> 40) & 0xff), R = (int)((pk >> 20) & 0xfffff), C = (int)(pk & 0xfffff);
        const unsigned start = ((unsigned)L << 28) | ((unsigned)R << 14) | (unsigned)C;      // generation order inside the octave: layer, row, column of the extremum
        const OctaveDev& oc = P.oc[o];
        auto dv = [&](int l, int r, int c) { return dogv(oc, foff, l, r, c); };
        FitOff f = {0.0f, 0.0f, 0.0f};
        int it = 0;
        bool alive = true, accepted = false;
        float contr = 0.0f;
        // Every Newton step works on the 3x3x3 DoG neighbourhood of its location held in registers: the one extrema_kernel saved next to the
        // candidate (bit 63; straight from its LDS planes; first step only), or gathered here -- three rows of four levels, the three columns
        // inside the four that start at the even column (C - 1) & ~1: twelve 8-byte loads instead of the ~50 two-byte ones the fit and the
        // acceptance tests read through dogv().  Round 4 gathered for the first step only and sent the candidates that move on (one in three)
        // through dogv() for their later steps; a mover's next location shares most of its 64-byte sectors with the last one, so gathering
        // again costs little memory and a quarter of the load instructions.  (Odd level widths: dogv() throughout.)
        const bool from_cube = (pk >> 63) != 0;
        const bool can_gather = (oc.w & 1) == 0 && (foff & 1) == 0;
        float cv[27];
        auto gather = [&](int l, int r, int c) {
            const int a0 = (c - 1) & ~1, sh = ((c - 1) & 1) * 16;
            int v[4][9];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const lvl_t* lp = oc.lv[l - 1 + q] + foff;
#pragma unroll
                for (int dr = 0; dr < 3; dr++) {
                    const uint2_a4 wv = *reinterpret_cast<const uint2_a4*>(lp + (size_t)(r - 1 + dr) * oc.w + a0);      // 4-byte aligned: a0, the row pitch and the frame offset are even
                    const unsigned long long ww = (((unsigned long long)wv.y << 32) | wv.x) >> sh;
                    v[q][dr * 3 + 0] = (int)(short)(ww & 0xffffu); v[q][dr * 3 + 1] = (int)(short)((ww >> 16) & 0xffffu); v[q][dr * 3 + 2] = (int)(short)((ww >> 32) & 0xffffu);
                }
            }
#pragma unroll
            for (int dl = 0; dl < 3; dl++)
#pragma unroll
                for (int e = 0; e < 9; e++) cv[dl * 9 + e] = (float)(v[dl + 1][e] - v[dl][e]);
        };
        if (from_cube || can_gather) {
            for (; it < MAX_INTERP; it++) {
                if (it == 0 && from_cube) {
                    const float* cb = cube_all + (size_t)reg * 32 * cube_cap + i;       // [element][candidate]: coalesced across the lanes
#pragma unroll
                    for (int e = 0; e < 27; e++) cv[e] = cb[(size_t)e * cube_cap];
                } else if (can_gather) gather(L, R, C);
                else break;                                                            // (a saved cube on an odd-width level: the later steps through dogv())
                const int L0 = L, R0 = R, C0 = C;
                auto dvc = [&](int l, int r, int c) { return cv[(l - L0 + 1) * 9 + (r - R0 + 1) * 3 + (c - C0 + 1)]; };
                f = fit_step(dvc, L, R, C);
                const int stt = fit_state(f);
                if (stt == 1) { alive = false; break; }
                if (stt == 0) { accepted = fit_accept(dvc, L, R, C, f, contrast_thr, edge_thr, contr); alive = accepted; break; }
                C += (int)rintf(f.xc); R += (int)rintf(f.xr); L += (int)rintf(f.xi);
                if (L < 1 || L > N_LAYERS || C < IMG_BORDER || C >= oc.w - IMG_BORDER || R < IMG_BORDER || R >= oc.h - IMG_BORDER) { alive = false; break; }
            }
            if (!alive || (can_gather && !accepted)) continue;                       // dead, rejected, or still moving after MAX_INTERP steps
        }
        if (!accepted) {
            for (; it < MAX_INTERP; it++) {
                f = fit_step(dv, L, R, C);
                const int stt = fit_state(f);
                if (stt == 0) break;
                if (stt == 1) { alive = false; break; }
                C += (int)rintf(f.xc); R += (int)rintf(f.xr); L += (int)rintf(f.xi);
                if (L < 1 || L > N_LAYERS || C < IMG_BORDER || C >= oc.w - IMG_BORDER || R < IMG_BORDER || R >= oc.h - IMG_BORDER) { alive = false; break; }
            }
            if (!alive || it >= MAX_INTERP) continue;
            if (!fit_accept(dv, L, R, C, f, contrast_thr, edge_thr, contr)) continue;
        }
        // duplicates: several start points may converge to one location -> first claim wins (all claims carry identical values)
        const size_t bit = ((size_t)R * oc.w + C) * 4 + (size_t)L;
        const unsigned mask = 1u << (bit & 31);
        if (!P.mins[o]) {
            const unsigned old = atomicOr(&P.claimed[o][fr * bs.claimed + (bit >> 5)], mask);
            if (old & mask) continue;
        }
        // one atomic per wave, not per point: ~67 000 points of a 12 MP frame on ONE counter serialise in the L2 (0.65 ms per batch measured)
        const unsigned long long act = __ballot(1);
        const int leader = __builtin_ctzll(act), lane_id = (int)(threadIdx.x & 63);
        unsigned slot = 0;
        if (lane_id == leader) slot = atomicAdd(out_count, (unsigned)__builtin_popcountll(act));
        slot = __shfl(slot, leader) + (unsigned)__builtin_popcountll(act & ((1ull << lane_id) - 1ull));
        if (slot < out_cap) {
            Refined rr;
            rr.o = o; rr.layer = L; rr.r = R; rr.c = C; rr.xi = f.xi; rr.xr = f.xr; rr.xc = f.xc; rr.contr = contr;
            rr.scl = sigma * det_exp2f(((float)L + f.xi) / (float)N_LAYERS);
            rr.start = start;
            out[slot] = rr;
            out_resp[slot] = __float_as_uint(fabsf(contr));
            // keep-all (nfeatures <= 0): OpenCV's list is in generation order and of several start points that converge to one location the
            // FIRST in that order keeps it -- every start point is stored, the location remembers its smallest start key, keepall_live_kernel
            // keeps the record that holds it.  The key is entered only once the record has its slot: keepall_reset_kernel walks the STORED
            // records, so "entry below all ones <=> a stored record points at it" holds also for a frame that overflows the list
            if (P.mins[o]) atomicMin(&P.mins[o][fr * bs.mins + bit], start);
        } else if (!P.mins[o]) atomicAnd(&P.claimed[o][fr * bs.claimed + (bit >> 5)], ~mask);      // list full (the frame fails): keep "bit set <=> record stored"
    }
}

// The claim bitmaps (4 bits per pixel of every octave: 32 MB per 12 MP frame) are zero between batches: instead of clearing
// them wholesale per batch, the bits this batch set -- exactly one per stored refined record -- are taken back.
__device__ __forceinline__ void unclaim_body(const PyrDev& P, const Refined* ref, const unsigned* ref_count, unsigned ref_cap, const BatchStride& bs, size_t fr, unsigned first, unsigned step) {
    ref += fr * bs.refined; ref_count += fr * CNT_STRIDE;
    unsigned n = *ref_count;
    if (n > ref_cap) n = ref_cap;
    for (unsigned i = first; i < n; i += step) {
        const Refined rr = ref[i];
        const size_t bit = ((size_t)rr.r * P.oc[rr.o].w + rr.c) * 4 + (size_t)rr.layer;
        atomicAnd(&P.claimed[rr.o][fr * bs.claimed + (bit >> 5)], ~(1u << (bit & 31)));
    }
}

// ---- keep-all (nfeatures <= 0, cv::SIFT's default of its own: every keypoint, in OpenCV's generation order) ------------------------------
// The reference's one committed run was made this way (tests/test_sift_reference_run.py).  No response threshold, no top-k: every refined
// point that holds its location's smallest start key is oriented, and the keypoints leave in the order (octave, start layer / row / column,
// orientation bin) -- the order of OpenCV's vector when retainBest does not run.  Not a hot path: the ranks are counted by brute force.
__global__ __launch_bounds__(256) void keepall_live_kernel(PyrDev P, const Refined* ref, const unsigned* ref_count, unsigned ref_cap, unsigned* ctrl, unsigned* list, BatchStride bs) {
    const size_t fr = blockIdx.y;
    ref += fr * bs.refined; ref_count += fr * CNT_STRIDE; ctrl += fr * CNT_STRIDE; list += fr * bs.refined;
    unsigned n = *ref_count;
    if (n > ref_cap) n = ref_cap;
    const int lane = threadIdx.x & 63;
    for (unsigned i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {      // uniform trip count per wave
        const unsigned i = i0 + threadIdx.x;
        bool take = false;
        if (i < n) {
            const Refined rr = ref[i];
            const size_t bit = ((size_t)rr.r * P.oc[rr.o].w + rr.c) * 4 + (size_t)rr.layer;
            take = P.mins[rr.o][fr * bs.mins + bit] == rr.start;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
        if (m) {
            const int first = __builtin_ctzll(m);
            unsigned base = 0;
            if (lane == first) base = atomicAdd(&ctrl[2], (unsigned)__builtin_popcountll(m));
            base = __shfl(base, first);
            if (take) list[base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        }
    }
}
// the start-key table is all ones between batches: the locations this batch touched are put back
__global__ __launch_bounds__(256) void keepall_reset_kernel(PyrDev P, const Refined* ref, const unsigned* ref_count, unsigned ref_cap, BatchStride bs) {
    const size_t fr = blockIdx.y;
    ref += fr * bs.refined; ref_count += fr * CNT_STRIDE;
    unsigned n = *ref_count;
    if (n > ref_cap) n = ref_cap;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Refined rr = ref[i];
        P.mins[rr.o][fr * bs.mins + ((size_t)rr.r * P.oc[rr.o].w + rr.c) * 4 + (size_t)rr.layer] = 0xffffffffu;
    }
}

// ---------- K4: orientation ---------------------------------------------------------------------------------------
struct KpRec {
    unsigned resp_bits; int o, layer, r, c, bin;
    float ptx, pty, scl, angle, xi;
    unsigned start;                     // keep-all: the refined point's start key
};

// Only the nfeatures strongest keypoints survive, and the response is known before the orientation: a 65536-bin
// histogram of the response bits (filled by refine_kernel) gives a threshold T such that at least `want` refined
// points have response >= T.  Pass 0 orients those; if that turns out to yield fewer than nfeatures keypoints
// (points without any histogram peak), top-k raises a flag and pass 1 orients the rest -- same final result as
// orienting everything, ~40x less work in the common case.
// The selection is one workgroup per frame walking ~67 000 responses three times (latency, not bandwidth: 32 workgroups on the whole
// chip), taking the claim bits back is 2 M scattered atomics per batch (bandwidth): as two launches they cost their sum (112 + 88 us per
// batch of 32), in ONE grid -- workgroup 0 of a frame selects, the others unclaim -- the longer of the two.
constexpr int UNCLAIM_WGS = 32;                       // workgroups per frame beside the selecting one
__global__ __launch_bounds__(1024) void select_unclaim_kernel(PyrDev P, const Refined* ref, BatchStride bs,
                                                              const unsigned* resp /* |response| bits of the refined points */, const unsigned* ref_count, unsigned ref_cap,
                                                              unsigned want, unsigned* ctrl /* [0]=T bits [1]=fallback flag [2]=points >= T */, size_t resp_stride,
                                                              unsigned* list /* indices of the points >= T, what orientation pass 0 walks */) {
    if (blockIdx.x > 0) { unclaim_body(P, ref, ref_count, ref_cap, bs, blockIdx.y, (blockIdx.x - 1) * 1024 + threadIdx.x, (gridDim.x - 1) * 1024); return; }
    // two-pass radix select over the 16-bit key (8 exponent + 8 mantissa bits) of the responses;
    // T = lower edge of the first key (from the top) at which the count of points with key >= it reaches `want`
    resp += (size_t)blockIdx.y * resp_stride; ref_count += (size_t)blockIdx.y * CNT_STRIDE; ctrl += (size_t)blockIdx.y * CNT_STRIDE;
    list += (size_t)blockIdx.y * resp_stride;
    __shared__ unsigned s_h[16][256];               // a private histogram per wave: LDS conflicts stay inside one wave
    __shared__ unsigned s_sel[2], s_cnt;
    const int tid = threadIdx.x, wv = tid >> 6;
    unsigned n = *ref_count;
    if (n > ref_cap) n = ref_cap;
    if (n <= want) {                                  // everything is oriented
        if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; ctrl[2] = n; }
        for (unsigned i = tid; i < n; i += 1024) list[i] = i;
        return;
    }
    const unsigned nr = (n + 1023u) & ~1023u;
    unsigned above = 0, hi_sel = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = tid; i < 16 * 256; i += 1024) (&s_h[0][0])[i] = 0;
        __syncthreads();
        for (unsigned i0 = tid; i0 < nr; i0 += 8 * 1024) {
            unsigned vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const unsigned i = i0 + 1024u * u; vv[u] = i < n ? resp[i] : 0u; }      // 8 loads in flight (one at a time: 65 round trips per pass)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (i0 + 1024u * u >= n) continue;
                const unsigned key = (vv[u] >> 15) & 0xffffu;
                if (pass == 0) atomicAdd(&s_h[wv][key >> 8], 1u);
                else if ((key >> 8) == hi_sel) atomicAdd(&s_h[wv][key & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid < 256) { unsigned t = 0; for (int w = 0; w < 16; w++) t += s_h[w][tid]; s_h[0][tid] = t; }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = above; int b = 255;
            for (; b >= 0; b--) { if (cum + s_h[0][b] >= want) break; cum += s_h[0][b]; }
            if (b < 0) b = 0;
            s_sel[0] = (unsigned)b; s_sel[1] = cum;
        }
        __syncthreads();
        if (pass == 0) { hi_sel = s_sel[0]; above = s_sel[1]; }
        else if (tid == 0) { ctrl[0] = ((hi_sel << 8) | s_sel[0]) << 15; ctrl[1] = 0; s_cnt = 0; }
        __syncthreads();
    }
    // the points at or above the threshold, compacted (order free: the total order is established by top-k), so that the
    // orientation pass hands exactly one point to each wave instead of letting 4096 waves look for ~2300 among ~94 000
    const unsigned T = ((hi_sel << 8) | s_sel[0]) << 15;
    const int lane = tid & 63;
    for (unsigned i0 = tid; i0 < nr; i0 += 8 * 1024) {
        unsigned vv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const unsigned i = i0 + 1024u * u; vv[u] = i < n ? resp[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned i = i0 + 1024u * u;
            const bool take = i < n && vv[u] >= T;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(take);
            if (m) {
                const int first = __builtin_ctzll(m);
                unsigned base = 0;
                if (lane == first) base = atomicAdd(&s_cnt, (unsigned)__builtin_popcountll(m));      // (LDS: a returning global atomic per wave and trip was a round trip each)
                base = __shfl(base, first);
                if (take) list[base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
            }
        }
    }
    __syncthreads();
    if (tid == 0) ctrl[2] = s_cnt;
}

constexpr int OCAP = 256;                     // emitted keypoints buffered per workgroup between flushes
__global__ __launch_bounds__(256) void orient_kernel(PyrDev P, const Refined* ref, const unsigned* ref_count, unsigned ref_cap,
                                                     KpRec* out, unsigned* out_resp, unsigned* out_count, unsigned out_cap,
                                                     const unsigned* ctrl, int pass, BatchStride bs, const unsigned* list) {
    const size_t fr = blockIdx.y, foff = fr * bs.pyr;             // frame of the batch
    ref += fr * bs.refined; ref_count += fr * CNT_STRIDE; out += fr * bs.kps; out_resp += fr * bs.kps; out_count += fr * CNT_STRIDE; ctrl += fr * CNT_STRIDE;
    list += fr * bs.refined;
    __shared__ unsigned long long s_hq[4][ORI_BINS];
    const unsigned Tbits = ctrl[0];
    if (pass == 1 && (ctrl[1] == 0 || Tbits == 0)) return;      // fallback pass not needed
    __shared__ float s_hist[4][ORI_BINS + 4];
    __shared__ KpRec s_out[OCAP];             // one global atomic per flush instead of one per keypoint
    __shared__ unsigned s_on, s_obase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned n = *ref_count;
    if (n > ref_cap) n = ref_cap;
    if (pass == 0) n = ctrl[2] < n ? ctrl[2] : n;               // pass 0 walks the compacted list of points >= T
    if (tid == 0) s_on = 0;
    __syncthreads();
    for (unsigned base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {        // uniform trip count per workgroup
        unsigned k = base + wv;
        bool take = false;
        if (k < n) {
            if (pass == 0) { k = list[k]; take = true; }
            else take = __float_as_uint(fabsf(ref[k].contr)) < Tbits;
        }
        if (take) {
            const Refined rr = ref[k];
            const OctaveDev& oc = P.oc[rr.o];
            const lvl_t* img = oc.lv[rr.layer] + foff;
            const int radius = (int)rintf(4.5f * rr.scl);
            const float osig = 1.5f * rr.scl;
            const float expf_scale = -1.0f / (2.0f * osig * osig);
            const int side = 2 * radius + 1, S = side * side;
            if (lane < ORI_BINS) s_hq[wv][lane] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // 4 samples per lane per trip: the 16 gradient loads of a trip are issued before any is consumed
            // lane l takes the samples [l * per, (l + 1) * per) in turn: one division per lane, then a running row / column
            const int per = (S + 63) / 64;
            int sc = lane * per, ic = sc / side, jc = sc - ic * side;
            for (int k0 = 0; k0 < per; k0 += 4) {
                float dxv[4], dyv[4]; int d2v[4]; bool okv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int s = k0 + u < per ? sc : S;
                    const int i = ic - radius, j = jc - radius;
                    sc++; jc++;
                    if (jc == side) { jc = 0; ic++; }
                    const int y = rr.r + i, x = rr.c + j;
                    okv[u] = (s < S) && !(y <= 0 || y >= oc.h - 1 || x <= 0 || x >= oc.w - 1);
                    d2v[u] = i * i + j * j;
                    dxv[u] = 0.0f; dyv[u] = 0.0f;
                    if (okv[u]) {
                        dxv[u] = (float)((int)img[(size_t)y * oc.w + x + 1] - (int)img[(size_t)y * oc.w + x - 1]);
                        dyv[u] = (float)((int)img[(size_t)(y - 1) * oc.w + x] - (int)img[(size_t)(y + 1) * oc.w + x]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!okv[u]) continue;
                    const float dx = dxv[u], dy = dyv[u];
                    const float wgt = det_expf((float)d2v[u] * expf_scale);
                    const float ang = det_atan2deg(dy, dx);
                    const float mag = sqrtf(dx * dx + dy * dy);
                    int b = (int)rintf(((float)ORI_BINS / 360.0f) * ang);
                    if (b >= ORI_BINS) b -= ORI_BINS;
                    if (b < 0) b += ORI_BINS;
                    const float t = wgt * mag;
                    atomicAdd(&s_hq[wv][b], (unsigned long long)(long long)(int)rintf(t * HIST_Q));      // order-free by definition; t <= sqrt(2) x 255 x 48: 32 bits hold t x 2^10
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane < ORI_BINS) s_hist[wv][lane] = (float)(long long)s_hq[wv][lane] * (1.0f / HIST_Q);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            float hs = 0.0f;
            if (lane < ORI_BINS) {
                const float m2 = s_hist[wv][(lane + ORI_BINS - 2) % ORI_BINS], m1 = s_hist[wv][(lane + ORI_BINS - 1) % ORI_BINS];
                const float p1 = s_hist[wv][(lane + 1) % ORI_BINS], p2 = s_hist[wv][(lane + 2) % ORI_BINS];
                hs = ((m2 + p2) * (1.0f / 16.0f) + (m1 + p1) * (4.0f / 16.0f)) + s_hist[wv][lane] * (6.0f / 16.0f);
            }
            float omax = hs;                                            // max over lanes (order independent)
            for (int off = 32; off > 0; off >>= 1) { const float o2 = __shfl_xor(omax, off); omax = o2 > omax ? o2 : omax; }
            const float hl = __shfl(hs, lane > 0 ? lane - 1 : ORI_BINS - 1);
            const float hr = __shfl(hs, lane < ORI_BINS - 1 ? lane + 1 : 0);
            const float mag_thr = omax * 0.8f;
            if (lane < ORI_BINS && hs > hl && hs > hr && hs >= mag_thr) {
                float bf = (float)lane + (0.5f * (hl - hr)) / ((hl - 2.0f * hs) + hr);
                bf = bf < 0.0f ? (float)ORI_BINS + bf : (bf >= (float)ORI_BINS ? bf - (float)ORI_BINS : bf);
                KpRec kr;
                kr.resp_bits = __float_as_uint(fabsf(rr.contr));
                kr.o = rr.o; kr.layer = rr.layer; kr.r = rr.r; kr.c = rr.c; kr.bin = lane;
                kr.ptx = (float)rr.c + rr.xc; kr.pty = (float)rr.r + rr.xr; kr.scl = rr.scl; kr.xi = rr.xi; kr.start = rr.start;
                kr.angle = (360.0f / (float)ORI_BINS) * bf;
                const unsigned slot = atomicAdd(&s_on, 1u);            // LDS counter; <= 18 peaks x 4 waves per trip
                if (slot < OCAP) s_out[slot] = kr;
            }
        }
        __syncthreads();
        // flush when the next trip (<= 4 x 18 peaks) might not fit, and after the last trip
        const bool last = base + gridDim.x * 4 >= n;
        if (s_on > OCAP - 72 || last) {
            const unsigned cntl = s_on < OCAP ? s_on : OCAP;
            if (tid == 0 && cntl) s_obase = atomicAdd(out_count, cntl);
            __syncthreads();
            for (unsigned i = tid; i < cntl; i += 256) {
                const unsigned g = s_obase + i;
                if (g < out_cap) { out[g] = s_out[i]; out_resp[g] = s_out[i].resp_bits; }
            }
            __syncthreads();
            if (tid == 0) s_on = 0;
            __syncthreads();
        }
    }
}

// ---------- K4b: strongest nfeatures in the total order -------------------------------------------------------------
constexpr int TOPK_CAP = 4096;      // survivors handed to the sort (nfeatures + ties)

struct SelRec { float ptx, pty, scl, angle; int o, layer; };

__device__ __forceinline__ unsigned long long tie_key(const KpRec& k) {
    return ((unsigned long long)k.o << 52) | ((unsigned long long)k.layer << 48) | ((unsigned long long)k.r << 28) | ((unsigned long long)k.c << 8) | (unsigned long long)k.bin;
}

__global__ __launch_bounds__(1024) void topk_kernel(const KpRec* kps, const unsigned* resp, const unsigned* kp_count, unsigned kp_cap, int nfeatures,
                                                    FrameOuts outs, SelRec* out_sel, int* out_n, int* overflow, unsigned* ctrl, int pass, BatchStride bs) {
    const size_t fr = blockIdx.x;                                 // one workgroup per frame of the batch
    kps += fr * bs.kps; resp += fr * bs.kps; kp_count += fr * CNT_STRIDE; out_sel += fr * bs.sel; out_n += fr * CNT_STRIDE; overflow += fr * CNT_STRIDE; ctrl += fr * CNT_STRIDE;
    mi355_keypoint* out_kp = outs.kp[0];
#pragma unroll
    for (int q = 1; q < SIFT_BATCH_MAX; q++) if ((int)fr == q) out_kp = outs.kp[q];
    __shared__ unsigned s_hist[256];
    if (pass == 1 && (ctrl[1] == 0 || ctrl[0] == 0)) return;     // fallback pass not needed
    __shared__ unsigned s_misc[8];
    __shared__ unsigned long long s_k0[TOPK_CAP];      // ~resp_bits (descending response first)
    __shared__ unsigned long long s_k1[TOPK_CAP];      // tie key
    __shared__ unsigned s_idx[TOPK_CAP];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned N = *kp_count;
    if (N > kp_cap) { N = kp_cap; if (tid == 0) *overflow = 1; }
    const unsigned K = (unsigned)nfeatures;
    unsigned thresh = 0;                                // select everything with resp_bits >= thresh
    if (N > K) {
        // radix select of the K-th largest resp_bits, MSB first; equal digits inside a wave are merged with
        // ballots before touching the LDS histogram (responses share their exponent byte: one hot bin)
        unsigned prefix = 0, want = K;
        const unsigned Nr = (N + 1023u) & ~1023u;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += 1024) s_hist[i] = 0;
            __syncthreads();
            const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (unsigned i0 = tid; i0 < Nr; i0 += 8 * 1024) {
                unsigned vv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const unsigned i = i0 + 1024u * u; vv[u] = i < N ? resp[i] : 0u; }      // 8 loads in flight
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const unsigned i = i0 + 1024u * u;
                    if (i0 + 1024u * u - tid >= Nr) break;                     // wave-uniform
                    const unsigned v = vv[u];
                    const bool valid = (i < N) && ((v & himask) == prefix);
                    const unsigned d = (v >> shift) & 255u;
                    unsigned long long peers = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const unsigned long long m = __ballot(valid && ((d >> b) & 1u));
                        peers &= ((d >> b) & 1u) ? m : ~m;
                    }
                    if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&s_hist[d], (unsigned)__popcll(peers));
                }
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0; int b = 255;
                for (; b >= 0; b--) { if (cum + s_hist[b] >= want) break; cum += s_hist[b]; }
                if (b < 0) b = 0;
                s_misc[0] = (unsigned)b; s_misc[1] = want - cum;
            }
            __syncthreads();
            prefix |= s_misc[0] << shift; want = s_misc[1];
            __syncthreads();
        }
        thresh = prefix;
    }
    if (tid == 0) s_misc[2] = 0;
    for (int i = tid; i < TOPK_CAP; i += 1024) { s_k0[i] = ~0ull; s_k1[i] = ~0ull; s_idx[i] = 0xffffffffu; }
    __syncthreads();
    for (unsigned i0 = tid; i0 < N; i0 += 8 * 1024) {
        unsigned vv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const unsigned i = i0 + 1024u * u; vv[u] = i < N ? resp[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned i = i0 + 1024u * u;
            if (i < N && vv[u] >= thresh) {
                const KpRec k = kps[i];
                const unsigned slot = atomicAdd(&s_misc[2], 1u);
                if (slot < TOPK_CAP) { s_k0[slot] = (unsigned long long)(~k.resp_bits); s_k1[slot] = tie_key(k); s_idx[slot] = i; }
            }
        }
    }
    __syncthreads();
    unsigned M = s_misc[2];
    if (M > TOPK_CAP) { M = TOPK_CAP; if (tid == 0) *overflow = 1; }
    // retainBest keeps every keypoint tied with the nfeatures-th response; the feature record holds SEL_STRIDE: more than that (nfeatures
    // close to 2048 and a tie at the boundary) is reported as an overflow rather than cut off silently
    if (N > K && M > (unsigned)SEL_STRIDE && tid == 0) *overflow = 1;
    // bitonic sort by (k0, k1) of the smallest power of two >= M (padding keys are all-ones and stay last)
    int SN = 64;
    while ((unsigned)SN < M) SN <<= 1;
    for (int k = 2; k <= SN; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < SN; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a0 = s_k0[i], a1 = s_k1[i], b0 = s_k0[ixj], b1 = s_k1[ixj];
                    const bool gt = (a0 > b0) || (a0 == b0 && a1 > b1);
                    const bool up = (i & k) == 0;
                    if (gt == up) {
                        s_k0[i] = b0; s_k1[i] = b1; s_k0[ixj] = a0; s_k1[ixj] = a1;
                        const unsigned t = s_idx[i]; s_idx[i] = s_idx[ixj]; s_idx[ixj] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
    // KeyPointsFilter::retainBest keeps everything whose response is >= the K-th strongest: the M survivors of the threshold are exactly
    // those (ties with the K-th included), up to the feature record's capacity
    const unsigned keep = N > K ? (M < (unsigned)SEL_STRIDE ? M : (unsigned)SEL_STRIDE) : M;
    for (unsigned i = tid; i < keep; i += 1024) {
        const KpRec k = kps[s_idx[i]];
        // image coordinates: kpt.pt = (c + xc) * (1 << octave), kpt.size = sigma * 2^((layer + xi) / 3) * (1 << octave) * 2 (no doubled octave)
        const float s2 = (float)(1 << k.o);
        mi355_keypoint kp;
        kp.x = k.ptx * s2; kp.y = k.pty * s2; kp.size = (k.scl * s2) * 2.0f; kp.angle = k.angle;
        kp.response = __uint_as_float(k.resp_bits);
        kp.octave = (k.o & 255) | (k.layer << 8) | (((int)rintf((k.xi + 0.5f) * 255.0f)) << 16);
        kp.class_id = -1;
        out_kp[i] = kp;
        SelRec sr; sr.ptx = k.ptx; sr.pty = k.pty; sr.scl = k.scl; sr.angle = k.angle; sr.o = k.o; sr.layer = k.layer;
        out_sel[i] = sr;
    }
    if (tid == 0) {
        *out_n = (int)keep;
        // pass 0 oriented only the refined points with response >= T: fewer than nfeatures keypoints came out while
        // weaker refined points exist -> orient those too (pass 1) and select again
        if (pass == 0 && ctrl[0] != 0 && keep < K) ctrl[1] = 1;
    }
}

// keep-all: a keypoint's place in the output = the number of keypoints with a smaller (octave, start key, orientation bin); the keys are
// distinct (one refined record per start point, one keypoint per bin).  Every workgroup walks all keys of its frame in tiles through LDS.
__global__ __launch_bounds__(256) void keepall_output_kernel(const KpRec* kps, const unsigned* kp_count, unsigned kp_cap, FrameOuts outs, SelRec* out_sel, int* out_n, int* overflow, BatchStride bs) {
    const size_t fr = blockIdx.y;
    kps += fr * bs.kps; kp_count += fr * CNT_STRIDE; out_sel += fr * bs.sel; out_n += fr * CNT_STRIDE; overflow += fr * CNT_STRIDE;
    mi355_keypoint* out_kp = outs.kp[0];
#pragma unroll
    for (int q = 1; q < SIFT_BATCH_MAX; q++) if ((int)fr == q) out_kp = outs.kp[q];
    unsigned N = *kp_count;
    if (N > kp_cap || N > (unsigned)KEEPALL_MAX) {                     // more keypoints than the record holds: the frame fails (OpenCV would keep them all)
        if (blockIdx.x == 0 && threadIdx.x == 0) { *overflow = 1; *out_n = 0; }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_n = (int)N;
    auto key_of = [](const KpRec& k) { return ((unsigned long long)k.o << 40) | ((unsigned long long)k.start << 6) | (unsigned long long)k.bin; };
    __shared__ unsigned long long s_key[1024];
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= N) return;
    KpRec mine;
    unsigned long long mk = ~0ull;
    if (i < N) { mine = kps[i]; mk = key_of(mine); }
    unsigned rank = 0;
    for (unsigned t0 = 0; t0 < N; t0 += 1024) {
        __syncthreads();
        for (unsigned q = threadIdx.x; q < 1024; q += 256) s_key[q] = t0 + q < N ? key_of(kps[t0 + q]) : ~0ull;
        __syncthreads();
        const unsigned cntq = N - t0 < 1024 ? N - t0 : 1024;
        for (unsigned q = 0; q < cntq; q++) rank += s_key[q] < mk ? 1u : 0u;
    }
    if (i >= N) return;
    const KpRec& k = mine;
    const float s2 = (float)(1 << k.o);
    mi355_keypoint kp;
    kp.x = k.ptx * s2; kp.y = k.pty * s2; kp.size = (k.scl * s2) * 2.0f; kp.angle = k.angle;
    kp.response = __uint_as_float(k.resp_bits);
    kp.octave = (k.o & 255) | (k.layer << 8) | (((int)rintf((k.xi + 0.5f) * 255.0f)) << 16);
    kp.class_id = -1;
    out_kp[rank] = kp;
    SelRec sr; sr.ptx = k.ptx; sr.pty = k.pty; sr.scl = k.scl; sr.angle = k.angle; sr.o = k.o; sr.layer = k.layer;
    out_sel[rank] = sr;
}

// ---------- K5: descriptors -----------------------------------------------------------------------------------------
// One WAVE per keypoint, four keypoints per workgroup (16 frames x 2000 keypoints as 256-thread workgroups of their own were
// dispatch- and latency-bound: each lane looped over ~20 samples with four dependent 2-byte loads each).  A lane takes four
// samples per trip and issues their 16 gradient loads before any is consumed; the histogram lives in wave-private LDS.
__global__ __launch_bounds__(256) void describe_kernel(PyrDev P, const SelRec* sel, const int* n_sel, FrameOuts outs, BatchStride bs) {
    const size_t fr = blockIdx.y, foff = fr * bs.pyr;             // frame of the batch
    sel += fr * bs.sel; n_sel += fr * CNT_STRIDE;
    uint8_t* desc = outs.d8[0];
#pragma unroll
    for (int q = 1; q < SIFT_BATCH_MAX; q++) if ((int)fr == q) desc = outs.d8[q];
    constexpr int d = 4, n = 8, HB = (d + 2) * (d + 2) * (n + 2);
    __shared__ unsigned long long s_hq4[4][HB];
    __shared__ float s_dst4[4][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kidx = blockIdx.x * 4 + wv;
    if (kidx >= *n_sel) return;
    unsigned long long* s_hq = s_hq4[wv];
    float* s_dst = s_dst4[wv];
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    const SelRec k = sel[kidx];
    const OctaveDev& oc = P.oc[k.o];
    const lvl_t* img = oc.lv[k.layer] + foff;
    const int rows = oc.h, cols = oc.w;
    for (int i = lane; i < HB; i += 64) s_hq[i] = 0ull;
    wave_sync();
    const int px = (int)rintf(k.ptx), py = (int)rintf(k.pty);
    float sin_t, cos_t;
    det_sincosdeg(k.angle, sin_t, cos_t);
    const float bins_per_deg = (float)n / 360.0f;
    const float exp_scale = -1.0f / ((float)(d * d) * 0.5f);
    const float hist_width = 3.0f * k.scl;
    const int radius = (int)rintf(hist_width * 1.4142135623730951f * (float)(d + 1) * 0.5f);
    cos_t = cos_t / hist_width; sin_t = sin_t / hist_width;
    const int side = 2 * radius + 1, S = side * side;
    // Only about half of the window's samples fall inside the rotated 4 x 4 grid.  The cheap part -- the sample's grid coordinates and the
    // test -- runs for all of them; the samples that pass are queued (window row, column packed into one word) in a wave-private LDS list
    // and the expensive part (four loads, atan2, exp, eight fixed-point atomics) runs on full waves of queued samples: 34 -> 24 us per
    // 12 MP frame.  The sums are order-free, so the order of the queue does not matter.
    __shared__ int s_q4[4][64 * 5];
    int* s_q = s_q4[wv];
    int qn = 0;                                       // wave-uniform
    auto heavy = [&](int packed) {
        const int i = (int)((unsigned)packed >> 16) - 32768, j = (int)((unsigned)packed & 0xffffu) - 32768;
        const float c_rot = (float)j * cos_t - (float)i * sin_t;
        const float r_rot = (float)j * sin_t + (float)i * cos_t;
        float rbin = r_rot + (float)(d / 2) - 0.5f;
        float cbin = c_rot + (float)(d / 2) - 0.5f;
        const int r = py + i, c = px + j;
        const float dx = (float)((int)img[(size_t)r * cols + c + 1] - (int)img[(size_t)r * cols + c - 1]);
        const float dy = (float)((int)img[(size_t)(r - 1) * cols + c] - (int)img[(size_t)(r + 1) * cols + c]);
        const float ori = det_atan2deg(dy, dx);
        const float mag = sqrtf(dx * dx + dy * dy) * det_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        float obin = (ori - k.angle) * bins_per_deg;
        const float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
        rbin -= r0f; cbin -= c0f; obin -= o0f;
        const int r0 = (int)r0f, c0 = (int)c0f;
        int o0 = (int)o0f;
        if (o0 < 0) o0 += n;
        if (o0 >= n) o0 -= n;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
        const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
        const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
        const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
        const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
        // order-free accumulation: rint(v * 2^10) as 64-bit integers (two's complement add == unsigned add)
#define FIXQ(v) ((unsigned long long)(long long)(int)rintf((v) * HIST_Q))     /* |v| <= sqrt(2) x 255 x 48: v x 2^10 fits 32 bits (one v_cvt_i32_f32 instead of the 64-bit conversion sequence) */
        atomicAdd(&s_hq[idx], FIXQ(v_rco000)); atomicAdd(&s_hq[idx + 1], FIXQ(v_rco001));
        atomicAdd(&s_hq[idx + (n + 2)], FIXQ(v_rco010)); atomicAdd(&s_hq[idx + (n + 3)], FIXQ(v_rco011));
        atomicAdd(&s_hq[idx + (d + 2) * (n + 2)], FIXQ(v_rco100)); atomicAdd(&s_hq[idx + (d + 2) * (n + 2) + 1], FIXQ(v_rco101));
        atomicAdd(&s_hq[idx + (d + 3) * (n + 2)], FIXQ(v_rco110)); atomicAdd(&s_hq[idx + (d + 3) * (n + 2) + 1], FIXQ(v_rco111));
#undef FIXQ
    };
    // lane l tests the samples [l * per, (l + 1) * per) in turn (one division per lane, then running row / column)
    const int per = (S + 63) / 64;
    int sc = lane * per, ic = sc / side, jc = sc - ic * side;
    for (int k0 = 0; k0 < per; k0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = k0 + u;
            const int i = ic - radius, j = jc - radius;
            const float c_rot = (float)j * cos_t - (float)i * sin_t;
            const float r_rot = (float)j * sin_t + (float)i * cos_t;
            const float rbin = r_rot + (float)(d / 2) - 0.5f;
            const float cbin = c_rot + (float)(d / 2) - 0.5f;
            const int r = py + i, c = px + j;
            const bool ok = kk < per && sc < S && (rbin > -1.0f && rbin < (float)d && cbin > -1.0f && cbin < (float)d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
            if (ok) s_q[qn + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int)(((unsigned)(i + 32768) << 16) | (unsigned)(j + 32768));
            qn += __builtin_popcountll(m);
            sc++; jc++;
            if (jc == side) { jc = 0; ic++; }
        }
        wave_sync();
        while (qn >= 64) {
            qn -= 64;
            heavy(s_q[qn + lane]);
        }
        wave_sync();
    }
    if (lane < qn) heavy(s_q[lane]);
    wave_sync();
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int t = lane + 64 * half;
        const int cell = t >> 3, q = t & 7, i = cell >> 2, j = cell & 3;
        const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
        float v = (float)(long long)s_hq[idx + q] * (1.0f / HIST_Q);
        if (q < 2) v = v + (float)(long long)s_hq[idx + n + q] * (1.0f / HIST_Q);       // circular orientation wrap
        s_dst[t] = v;
    }
    wave_sync();
    // sequential norms: the accumulation order is part of the definition.  The squares are formed by the 64 lanes, the
    // running sum visits them in index order through lane broadcasts
    float a = s_dst[lane], b = s_dst[lane + 64];
    auto seq_sum = [&](float va, float vb) {
        const float sa = va * va, sb = vb * vb;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; q++) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa), q));
#pragma unroll
        for (int q = 0; q < 64; q++) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sb), q));
        return acc;
    };
    const float thr = sqrtf(seq_sum(a, b)) * 0.2f;
    a = a < thr ? a : thr; b = b < thr ? b : thr;
    const float nn = sqrtf(seq_sum(a, b));
    const float fac = 512.0f / (nn > 1.1920929e-7f ? nn : 1.1920929e-7f);
    const float va = rintf(a * fac), vb = rintf(b * fac);
    desc[(size_t)kidx * 128 + lane] = (uint8_t)(va < 0.0f ? 0 : (va > 255.0f ? 255 : (int)va));
    desc[(size_t)kidx * 128 + lane + 64] = (uint8_t)(vb < 0.0f ? 0 : (vb > 255.0f ? 255 : (int)vb));
}

// ---------- host ---------------------------------------------------------------------------------------------------
int gauss_kernel_host(double sigma, float* k) {
    const int ksize = ((int)lrint(sigma * 8.0 + 1.0)) | 1;
    const int r = ksize / 2;
    // cv::getGaussianKernel(ksize, sigma, CV_32F): each exp() rounded to float, the floats summed in double, taps (float)(tap / sum)
    double sum = 0.0;
    const double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; i++) { const double x = (double)i - (double)(ksize - 1) * 0.5; k[i] = (float)std::exp(scale2x * x * x); sum += (double)k[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; i++) k[i] = (float)((double)k[i] * sum);
    return r;
}

// does this level go through blur16_stream?  (8-byte aligned rows, last strip wider than the largest radius, enough strips x frames to fill the chip;
// the decimated copy needs an even height: the even rows of every segment are then the even rows of the image)
inline bool blur_streams(const Blur16Args& a, bool bgr, int R, int stream_mode) {
    const bool has_r = bgr ? R == 6 : (R == 5 || R == 6 || R == 8 || R == 10 || R == 13);
    bool ok = stream_mode && has_r && (a.w & 3) == 0 && ((a.w & 255) == 0 || (a.w & 255) > MAX_R) && a.w >= 512 && a.h >= 64;
    if (bgr) { for (int f = 0; f < a.nb; f++) ok = ok && ((uintptr_t)a.bgr[f] & 3) == 0 && (a.bgr_ws[f] & 3) == 0 && a.bgr_ws[f] >= 3 * a.w; }
    else ok = ok && ((uintptr_t)a.src & 7) == 0;
    if (a.ds) ok = ok && R == 8 && (a.h & 1) == 0;
    return ok && ((uintptr_t)a.dst & 7) == 0 && (a.fstride & 3) == 0;
}
inline void stream_grid(int w, int h, int& L, int& nstrip, int& nseg, int nb = 1, int waves = 2) {
    static const int units_env = [] { const char* e = getenv("MI355_STREAM_UNITS"); return e ? atoi(e) : 0; }();
    const int units_target = units_env ? units_env : 1024 * waves;
    static const int stream_minl = [] { const char* e = getenv("MI355_STREAM_MINL"); return e ? atoi(e) : 64; }();
    nstrip = (w + 255) / 256;
    nseg = (units_target + nstrip * nb - 1) / (nstrip * nb);
    L = (h + nseg - 1) / nseg;
    if (L < stream_minl) L = stream_minl;
    L = (L + 1) & ~1;
    nseg = (h + L - 1) / L;
}
template <bool BGR>
bool launch_blur(hipStream_t st, int R, const Blur16Args& a_in, int stream_mode, bool* streamed = nullptr) {
    Blur16Args a = a_in;
    const int nb = a.nb > 1 ? a.nb : 1;
    if (streamed) *streamed = false;
    if (blur_streams(a, BGR, R, stream_mode)) {
        // barrier-free streaming kernel over the whole chip: W waves per SIMD (W x 1024 waves, one round), segments of >= 64 rows, all frames of
        // a batch in one launch.  The register ring of the row results (4 x (2R + 2) registers) decides W: R <= 8 fits 128 registers, R = 10 / 13 168
        static const int w4 = [] { const char* e = getenv("MI355_STREAM_W4"); return e ? atoi(e) : 8; }();
        const int waves = (R <= w4 && R <= 8) ? 4 : 3;      // (grids sized for one wave per SIMD fewer, to leave registers to the other batches' keypoint kernels, measured the same: profiles/r05_pipeline_layouts.txt)
        double ksum = 0.0;
        for (int t = 0; t <= 2 * R; t++) ksum += std::fabs((double)a.k[t]);
        if (ksum < 2.6) {                            // the kernel's rounding assumes results in [0, 32767]: samples <= 255 * 48, taps positive and normalised
            for (int t = 0; t <= R; t++) { a.kp[2 * t] = a.k[t]; a.kp[2 * t + 1] = t ? a.k[t - 1] : 0.0f; }
            int L, nstrip, nseg;
            stream_grid(a.w, a.h, L, nstrip, nseg, nb, waves);
            const int units = nstrip * nseg * nb;
            const dim3 grid((units + 3) / 4), block(256);
            if (streamed) *streamed = true;
#define LAUNCH(RR, DS, WW) hipLaunchKernelGGL((blur16_stream<RR, BGR, DS, WW>), grid, block, 0, st, a, L, nstrip, nseg); return true
            if constexpr (BGR) { if (waves == 4) { LAUNCH(6, false, 4); } else { LAUNCH(6, false, 3); } }
            else {
                switch (R) {
                    case 5: if (waves == 4) { LAUNCH(5, false, 4); } else { LAUNCH(5, false, 3); }
                    case 6: if (waves == 4) { LAUNCH(6, false, 4); } else { LAUNCH(6, false, 3); }
                    case 8: if (a.ds) { if (waves == 4) { LAUNCH(8, true, 4); } else { LAUNCH(8, true, 3); } }
                            else { if (waves == 4) { LAUNCH(8, false, 4); } else { LAUNCH(8, false, 3); } }
                    case 10: LAUNCH(10, false, 3);
                    case 13: LAUNCH(13, false, 3);
                    default: break;
                }
            }
#undef LAUNCH
        }
    }
    const dim3 grid(a.tiles_x * a.tiles_y, nb), block(256);
    switch (R) {
#define CASE(RR) case RR: hipLaunchKernelGGL((blur16_tile<RR, BGR>), grid, block, 0, st, a); return true;
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
#undef CASE
        default: return false;
    }
}

}  // namespace

// One batch work area: room for `nb` frames (every per-frame buffer nb times, a fixed stride apart) and one stream.
// Frames handed to mi_sift_extract_dev() collect in `pend`; a full batch (or a flush) is enqueued as
//   phase 1  per frame : the chip-filling kernels of the big octaves (base level, blur_stream, extrema)
//   phase 2  per octave: the small octaves of ALL frames of the batch in one launch each (grid dimension = frame)
//   phase 3  per stage : refine / threshold / orientation / top-k / descriptors of ALL frames in one launch each
// so that the latency-bound launches (30 tiny blurs, single-workgroup selections) are paid once per batch instead of
// once per frame, on ONE in-order stream -- no reliance on how the runtime maps streams to hardware queues.
struct SiftWork {
    int w = 0, h = 0;                        // input frame size the buffers are sized for
    int nb = 0;                              // frames the buffers hold
    int n_oct = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;               // recorded after the last launch of the latest batch
    DevBuf pyr;                              // all Gaussian levels
    DevBuf claimed;                          // duplicate claim bitmaps
    DevBuf cand, refined, kps, kresp, sel, counters, rhist, ccnt;
    DevBuf olist;                            // per frame: indices of the refined points at or above the response threshold
    DevBuf mins; bool keepall = false;       // keep-all (nfeatures <= 0): smallest start key per refined location, all ones between batches
    DevBuf cube; unsigned cube_cap = 0;       // 3x3x3 DoG neighbourhoods of the first cube_cap candidates of every region (128 B each)
    PyrDev P;                                // pointers of frame 0
    BatchStride bs;
    unsigned cand_cap = 0, ref_cap = 0, kp_cap = 0;
    float kern[N_LEVELS][2 * MAX_R + 1];
    int radius[N_LEVELS];
    float kern0[2 * MAX_R + 1]; int radius0 = 0;
    struct Pend { int img_id; const uint8_t* d_bgr; int ws; hipEvent_t ev; };   // ev: recorded once the batch is enqueued (optional)
    std::vector<Pend> pend;
};

constexpr int SIFT_SLOTS_MAX = 4;            // batch work areas (each with its own stream) that may be in flight
#define SIFT_SLOTS (ctx->sift_nslots)

void mi_sift_release(mi355_ctx* ctx) {
    for (SiftWork* s : ctx->sift_slots) {
        if (!s) continue;
        if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
        if (s->done) (void)hipEventDestroy(s->done);
        s->pyr.release(); s->claimed.release(); s->cand.release(); s->refined.release(); s->kps.release(); s->kresp.release(); s->sel.release(); s->counters.release(); s->rhist.release(); s->ccnt.release(); s->cube.release(); s->olist.release(); s->mins.release();
        delete s;
    }
    ctx->sift_slots.clear();
    if (ctx->sift_in_ev) { (void)hipEventDestroy(ctx->sift_in_ev); ctx->sift_in_ev = nullptr; }
    for (int* p : ctx->pinned_chunks) (void)hipHostFree(p);
    ctx->pinned_chunks.clear();
    ctx->pinned_used = 0;
}

static int sift_run_batch(mi355_ctx* ctx, SiftWork* s);

// enqueues every partly filled batch
int mi_sift_flush(mi355_ctx* ctx) {
    int rc = MI355_OK;
    for (SiftWork* s : ctx->sift_slots) if (s && !s->pend.empty()) { const int r = sift_run_batch(ctx, s); if (r != MI355_OK) rc = r; }
    return rc;
}

// adopts the keypoint count of a finished frame from pinned memory
static int adopt_counts(mi355_ctx* ctx, int img_id, Features& f) {
    f.pending = false;
    if (!f.h_cnt) { f.n = 0; return MI355_OK; }               // its batch failed to launch
    const volatile int* c = f.h_cnt;
    for (int i = 0; i < 8; i++) ctx->last_counts[i] = c[i];
    if ((unsigned)c[0] > f.caps[0] || (unsigned)c[1] > f.caps[1] || (unsigned)c[2] > f.caps[2] || c[4]) {
        ctx->set_error("sift: candidate buffer overflow (image " + std::to_string(img_id) + " has more extrema than the buffers assume)");
        f.n = 0;
        return MI355_ERR_FAILED;
    }
    f.n = c[3];
    return MI355_OK;
}

// waits for every in-flight frame and adopts the keypoint counts that landed in pinned memory
int mi_resolve_features(mi355_ctx* ctx) {
    int rc = mi_sift_flush(ctx);
    bool any = false;
    for (auto& kv : ctx->feats) if (kv.second.pending) { any = true; break; }
    if (!any) { ctx->batch_events_used = 0; return rc; }
    for (SiftWork* s : ctx->sift_slots) if (s && s->stream) MI_HIP(hipStreamSynchronize(s->stream));
    for (auto& kv : ctx->feats) {
        if (!kv.second.pending) continue;
        const int r = adopt_counts(ctx, kv.first, kv.second);
        if (r != MI355_OK) rc = r;
    }
    ctx->batch_events_used = 0;
    return rc;
}

// the same for the given frames only: the host waits for THEIR batches (batch events), later batches keep running -- a caller
// that matches pairs while the rest of the survey is still in detect+describe overlaps the two (bench.py, window surveys)
int mi_resolve_features_of(mi355_ctx* ctx, const int* ids, int n) {
    int rc = MI355_OK;
    bool parked = false;
    for (int k = 0; k < n && !parked; k++) {
        auto it = ctx->feats.find(ids[k]);
        if (it != ctx->feats.end() && it->second.pending && !it->second.h_cnt && !it->second.ready) parked = true;
    }
    if (parked) rc = mi_sift_flush(ctx);                      // some of them still wait for their batch to fill: enqueue it
    for (int k = 0; k < n; k++) {
        auto it = ctx->feats.find(ids[k]);
        if (it == ctx->feats.end() || !it->second.pending) continue;
        Features& f = it->second;
        if (f.ready) MI_HIP(hipEventSynchronize(f.ready));
        else { const int r = mi_resolve_features(ctx); if (r != MI355_OK) rc = r; continue; }
        const int r = adopt_counts(ctx, it->first, f);
        if (r != MI355_OK) rc = r;
    }
    return rc;
}

static constexpr size_t PINNED_CHUNK = 4096;   // frames per pinned chunk

static int* pinned_slot(mi355_ctx* ctx) {
    const size_t chunk = ctx->pinned_used / PINNED_CHUNK, off = ctx->pinned_used % PINNED_CHUNK;
    if (chunk >= ctx->pinned_chunks.size()) {
        int* p = nullptr;
        if (hipHostMalloc((void**)&p, PINNED_CHUNK * 8 * sizeof(int), hipHostMallocDefault) != hipSuccess) return nullptr;
        ctx->pinned_chunks.push_back(p);
    }
    ctx->pinned_used++;
    return ctx->pinned_chunks[chunk] + off * 8;
}

static inline size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

static int sift_prepare(mi355_ctx* ctx, SiftWork* s, int w, int h, int nb, bool keepall) {
    if (s->w == w && s->h == h && s->nb == nb && s->keepall == keepall) return MI355_OK;
    MI_HIP(hipStreamSynchronize(s->stream));
    if (s->radius0 == 0) {
        // Gaussian kernels (double math on the host, like the oracle): sigma_i = sqrt((s k^i)^2 - (s k^(i-1))^2)
        const double sigma = 1.6, k = std::pow(2.0, 1.0 / N_LAYERS);
        for (int i = 1; i < N_LEVELS; i++) {
            const double sp = std::pow(k, (double)(i - 1)) * sigma, st = sp * k;
            s->radius[i] = gauss_kernel_host(std::sqrt(st * st - sp * sp), s->kern[i]);
        }
        // base level: createInitialImage(image, false, sigma) blurs with sqrtf(max(sigma^2 - 0.5^2, 0.01f)), computed in float
        const float sd = std::sqrt(std::max((float)sigma * (float)sigma - 0.25f, 0.01f));
        s->radius0 = gauss_kernel_host((double)sd, s->kern0);
    }
    // octave 0 is the image itself (no doubling in the reference's OpenCV build); nOctaves = cvRound(log2(min(w, h)) - 2)
    int nOct = (int)lrint(std::log((double)(w < h ? w : h)) / std::log(2.0) - 2.0);
    if (nOct > MAX_OCT) nOct = MAX_OCT;
    size_t fl = 0, cl = 0;
    int no = 0;
    for (int o = 0; o < nOct; o++) {
        const int ow = w >> o, oh = h >> o;
        if (ow < 2 * IMG_BORDER + 2 || oh < 2 * IMG_BORDER + 2) break;          // no keypoint can exist in smaller octaves
        fl += up64((size_t)ow * oh) * N_LEVELS;                     // every level starts on a 128-byte boundary
        cl += (((size_t)ow * oh * 4 + 31) / 32 + 63) & ~(size_t)63;
        no = o + 1;
    }
    if (no == 0) { ctx->set_error("sift: image too small"); return MI355_ERR_ARG; }
    // capacities.  Candidates are DoG extrema with |DoG| > 20: a plateau of equal values is the worst case (every pixel a tied
    // extremum); 3 layers x sum_o px0 / 4^o <= 4 px0 is provisioned.  Refined points / keypoints must survive the contrast test.
    const size_t px0 = (size_t)w * h;
    if (4 * px0 + 1024 > 0xfffffff0ull) { ctx->set_error("sift: image too large"); return MI355_ERR_ARG; }
    s->cand_cap = (unsigned)((4 * px0 + 1024 + NREG - 1) / NREG + ((size_t)MAX_OCT * 3 * EW * EH << REG_SHIFT) + 1024);   // + one full run of tiles per octave
    s->ref_cap = (unsigned)(px0 / 8 + 65536);
    s->kp_cap = (unsigned)(px0 / 8 + 65536);
    s->cube_cap = (unsigned)(px0 / 4 / NREG + 4096);
    if (s->cube_cap > s->cand_cap) s->cube_cap = s->cand_cap;
    s->bs.cube = (size_t)s->cube_cap * NREG * 32;
    s->bs.pyr = fl; s->bs.claimed = cl; s->bs.cand = (size_t)s->cand_cap * NREG; s->bs.refined = s->ref_cap; s->bs.kps = s->kp_cap;
    s->bs.sel = keepall ? (size_t)KEEPALL_MAX : SEL_STRIDE;
    size_t ml = 0;                                              // keep-all: one word per claim bit (4 per pixel of every octave)
    if (keepall) for (int o = 0; o < no; o++) ml += (size_t)(w >> o) * (h >> o) * 4;
    s->bs.mins = ml;
    const size_t B = (size_t)nb;
    MI_HIP(s->pyr.reserve(B * fl * sizeof(lvl_t)));
    MI_HIP(s->claimed.reserve(B * cl * sizeof(unsigned)));
    MI_HIP(hipMemsetAsync(s->claimed.p, 0, B * cl * sizeof(unsigned), s->stream));      // kept zero between batches (select_unclaim_kernel takes the bits back)
    MI_HIP(s->cand.reserve(B * s->bs.cand * sizeof(unsigned long long)));
    MI_HIP(s->refined.reserve(B * s->bs.refined * sizeof(Refined)));
    MI_HIP(s->kps.reserve(B * s->bs.kps * sizeof(KpRec)));
    MI_HIP(s->kresp.reserve(B * s->bs.kps * sizeof(unsigned)));
    MI_HIP(s->sel.reserve(B * s->bs.sel * sizeof(SelRec)));
    if (keepall) { MI_HIP(s->mins.reserve(B * ml * sizeof(unsigned))); MI_HIP(hipMemsetAsync(s->mins.p, 0xff, B * ml * sizeof(unsigned), s->stream)); }      // kept all ones between batches by keepall_reset_kernel
    else s->mins.release();
    MI_HIP(s->counters.reserve(B * CNT_STRIDE * sizeof(unsigned)));
    MI_HIP(s->rhist.reserve(B * s->bs.refined * sizeof(unsigned)));
    MI_HIP(s->olist.reserve(B * s->bs.refined * sizeof(unsigned)));      // |response| bits of the refined points (SoA next to `refined`)
    MI_HIP(s->ccnt.reserve(B * CCNT_STRIDE * sizeof(unsigned)));
    MI_HIP(s->cube.reserve(B * s->bs.cube * sizeof(float)));
    memset(&s->P, 0, sizeof(s->P));
    size_t fo = 0, co = 0, mo = 0;
    for (int o = 0; o < no; o++) {
        const int ow = w >> o, oh = h >> o;
        s->P.oc[o].w = ow; s->P.oc[o].h = oh;
        for (int i = 0; i < N_LEVELS; i++) { s->P.oc[o].lv[i] = s->pyr.as<lvl_t>() + fo; fo += up64((size_t)ow * oh); }
        s->P.claimed[o] = s->claimed.as<unsigned>() + co;
        co += (((size_t)ow * oh * 4 + 31) / 32 + 63) & ~(size_t)63;
        s->P.mins[o] = keepall ? s->mins.as<unsigned>() + mo : nullptr;
        mo += (size_t)ow * oh * 4;
    }
    s->P.n_oct = no; s->n_oct = no;
    s->w = w; s->h = h; s->nb = nb; s->keepall = keepall;
    return MI355_OK;
}


// Accepts one frame: it joins the current batch, which is enqueued once it holds ctx->sift_batch frames (or on a
// flush: any call that needs features, mi355_synchronize, or n_kp != NULL).  The frame memory must stay valid and
// unchanged until then.  Returns without waiting; the keypoint count is adopted later by mi_resolve_features().
int mi_sift_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int ws, int* n_kp) {
    if (ctx->p.n_octave_layers != N_LAYERS || ctx->p.sigma != 1.6f) { ctx->set_error("sift: this build implements nOctaveLayers=3, sigma=1.6 (the reference's SIFT(2000,3,0.01,20))"); return MI355_ERR_ARG; }
    if (ctx->p.nfeatures > 2048) { ctx->set_error("sift: nfeatures must be in [1,2048], or <= 0 for cv::SIFT's keep-all (up to 32768 keypoints per frame)"); return MI355_ERR_ARG; }
    const bool keepall = ctx->p.nfeatures <= 0;
    if (keepall && (w > 16384 || h > 16384)) { ctx->set_error("sift: keep-all frames are at most 16384 x 16384"); return MI355_ERR_ARG; }
    if ((size_t)w >= (1u << 20) || (size_t)h >= (1u << 20)) { ctx->set_error("sift: image too large"); return MI355_ERR_ARG; }
    if (w < 16 || h < 16) { ctx->set_error("sift: image too small"); return MI355_ERR_ARG; }
    if (ctx->sift_slots.empty()) {
        ctx->sift_slots.resize(SIFT_SLOTS_MAX, nullptr);
        MI_HIP(hipEventCreateWithFlags(&ctx->sift_in_ev, hipEventDisableTiming));
    }
    if (ctx->sift_next >= SIFT_SLOTS) ctx->sift_next = 0;
    const int slot = ctx->sift_next;
    if (!ctx->sift_slots[slot]) {
        ctx->sift_slots[slot] = new SiftWork();
        MI_HIP(hipStreamCreateWithFlags(&ctx->sift_slots[slot]->stream, hipStreamNonBlocking));
        MI_HIP(hipEventCreateWithFlags(&ctx->sift_slots[slot]->done, hipEventDisableTiming));
    }
    SiftWork* s = ctx->sift_slots[slot];
    int rc = MI355_OK;
    int nb = ctx->sift_batch < 1 ? 1 : (ctx->sift_batch > SIFT_BATCH_MAX ? SIFT_BATCH_MAX : ctx->sift_batch);
    {
        // A frame's work area is ~60 bytes per pixel (pyramid 16, worst-case candidate list 32, neighbourhood records 8, the rest 4): keep
        // slots x batch x that under 60 % of the device memory by shortening the batch for very large frames
        static size_t total_mem = [] { size_t fr = 0, tot = 0; return hipMemGetInfo(&fr, &tot) == hipSuccess ? tot : (size_t)0; }();
        const double per_frame = 60.0 * (double)w * (double)h;
        if (total_mem) {
            const int fit = (int)(0.6 * (double)total_mem / ((keepall ? 1.4 : 1.0) * per_frame) / (double)SIFT_SLOTS);      // keep-all: + 21 B per pixel of start keys
            if (fit < nb) nb = fit < 1 ? 1 : fit;
        }
        if (keepall && nb > 8) nb = 8;
    }
    if (!s->pend.empty() && (s->w != w || s->h != h || s->nb != nb || s->keepall != keepall)) { rc = sift_run_batch(ctx, s); if (rc != MI355_OK) return rc; }   // size change: close the batch
    rc = sift_prepare(ctx, s, w, h, nb, keepall);
    if (rc != MI355_OK) return rc;
    auto fit = ctx->feats.find(img_id);
    if (fit != ctx->feats.end() && fit->second.pending) { rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; }   // same id re-extracted while in flight
    Features& f = ctx->feats[img_id];
    f.w = w; f.h = h;
    MI_HIP(f.kp.reserve(sizeof(mi355_keypoint) * (size_t)(keepall ? KEEPALL_MAX : 2048)));
    MI_HIP(f.d8.reserve((size_t)128 * (size_t)(keepall ? KEEPALL_MAX : 2048)));
    f.pending = true; f.h_cnt = nullptr; f.ready = nullptr; f.n = 0;
    s->pend.push_back({img_id, d_bgr, ws, ctx->pend_event});
    if ((int)s->pend.size() >= nb) {
        rc = sift_run_batch(ctx, s);
        if (rc != MI355_OK) return rc;
    }
    if (ctx->pinned_used >= PINNED_CHUNK * 64) {           // recycle the pinned pool when nothing is pending any more
        rc = mi_resolve_features(ctx);
        if (rc != MI355_OK) return rc;
        ctx->pinned_used = 0;
    }
    if (n_kp) {
        rc = mi_resolve_features(ctx);
        if (rc != MI355_OK) return rc;
        *n_kp = f.n;
    }
    return MI355_OK;
}

static int sift_run_batch(mi355_ctx* ctx, SiftWork* s) {
    std::vector<SiftWork::Pend> pend;
    pend.swap(s->pend);
    const int n = (int)pend.size();
    if (n == 0) return MI355_OK;
    ctx->sift_next = (ctx->sift_next + 1) % SIFT_SLOTS;     // the next batch collects in the next work area
    const hipStream_t st = s->stream;
    const hipStream_t tt = st;                        // (the keypoint stages; on a stream of their own, at a lower queue priority or on CUs of their own they only lose: profiles/r05_pipeline_layouts.txt)
    const int w = s->w, h = s->h;
    const int nf = ctx->p.nfeatures;
    const BatchStride bs = s->bs;
    // the frames were produced on the caller's stream
    MI_HIP(hipEventRecord(ctx->sift_in_ev, ctx->stream));
    MI_HIP(hipStreamWaitEvent(st, ctx->sift_in_ev, 0));
    // (measurement mode) the previous batch's pyramid + extrema first.  The wait stands BEFORE the memsets: an event recorded right
    // after a wait takes the end of the stream's last command as its time, which would put the waiting into the first bracket
    const bool serial_heavy = ctx->serial_heavy != 0;
    if (serial_heavy && ctx->heavy_ev_valid) MI_HIP(hipStreamWaitEvent(st, ctx->heavy_ev, 0));
    unsigned* cnt = s->counters.as<unsigned>();      // per frame: [0] candidates [1] refined [2] keypoints [3] n_sel [4] overflow [8,9] ctrl
    MI_HIP(hipMemsetAsync(cnt, 0, (size_t)n * CNT_STRIDE * sizeof(unsigned), st));
    MI_HIP(hipMemsetAsync(s->ccnt.p, 0, (size_t)n * CCNT_STRIDE * sizeof(unsigned), st));
    FrameOuts outs;
    memset(&outs, 0, sizeof(outs));
    std::vector<Features*> fs(n);
    for (int k = 0; k < n; k++) {
        fs[k] = &ctx->feats[pend[k].img_id];
        outs.kp[k] = fs[k]->kp.as<mi355_keypoint>(); outs.d8[k] = fs[k]->d8.as<uint8_t>();
    }
    auto blur_args = [&](const OctaveDev& oc) {
        Blur16Args a; memset(&a, 0, sizeof(a));
        a.w = oc.w; a.h = oc.h; a.tiles_x = (oc.w + T16W - 1) / T16W; a.tiles_y = (oc.h + T16H - 1) / T16H;
        a.fstride = bs.pyr; a.nb = n;
        return a;
    };
    // ---- phases 1+2: the pyramid, octave by octave, every launch covering all n frames of the batch ----
    bool ds_fused = false;
    for (int o = 0; o < s->n_oct; o++) {
        const OctaveDev& oc = s->P.oc[o];
        const double level_bytes = (double)oc.w * oc.h * sizeof(lvl_t) * n;
        if (o == 0) {
            // base level straight from the caller's frames: gray x 48 formed on the fly, blurred with sqrt(1.6^2 - 0.5^2)
            Blur16Args a = blur_args(oc);
            for (int k = 0; k < n; k++) { a.bgr[k] = pend[k].d_bgr; a.bgr_ws[k] = pend[k].ws; }
            a.dst = oc.lv[0];
            memcpy(a.k, s->kern0, sizeof(float) * (2 * s->radius0 + 1));
            const bool streams = blur_streams(a, true, s->radius0, ctx->blur_stream);
            ProfScope ps(ctx, streams ? "gauss_stream" : "gauss", level_bytes + (double)w * h * 3.0 * n, st);      // read the u8 frames, write level 0
            if (!launch_blur<true>(st, s->radius0, a, ctx->blur_stream)) { ctx->set_error("sift: unsupported kernel radius"); return MI355_ERR_FAILED; }
        } else if (!ds_fused) {
            const OctaveDev& pv = s->P.oc[o - 1];
            ProfScope ps(ctx, "downsample", level_bytes * 2.0, st);
            hipLaunchKernelGGL(downsample16, dim3((oc.w + 63) / 64, (oc.h + 3) / 4, n), dim3(256), 0, st, pv.lv[N_LAYERS], pv.w, oc.lv[0], oc.w, oc.h, bs.pyr);
        }
        ds_fused = false;
        for (int i = 1; i < N_LEVELS; i++) {
            Blur16Args a = blur_args(oc);
            a.src = oc.lv[i - 1]; a.dst = oc.lv[i];
            memcpy(a.k, s->kern[i], sizeof(float) * (2 * s->radius[i] + 1));
            // the level that seeds the next octave writes its decimation on the way out (saves re-reading it)
            if (i == N_LAYERS && o + 1 < s->n_oct && (oc.w & 3) == 0 && (s->P.oc[o + 1].w == (oc.w >> 1)) && (s->P.oc[o + 1].h == (oc.h >> 1))) {
                a.ds = s->P.oc[o + 1].lv[0]; ds_fused = true;
                if (!blur_streams(a, false, s->radius[i], ctx->blur_stream)) {         // (odd height, unusual radius:) rather stream without the copy
                    Blur16Args b = a; b.ds = nullptr;
                    if (blur_streams(b, false, s->radius[i], ctx->blur_stream)) { a.ds = nullptr; ds_fused = false; }
                }
            }
            const bool streams = blur_streams(a, false, s->radius[i], ctx->blur_stream);
            ProfScope ps(ctx, streams ? "gauss_stream" : "gauss", level_bytes * 2.0, st);   // one read + one write of the level
            if (!launch_blur<false>(st, s->radius[i], a, ctx->blur_stream)) { ctx->set_error("sift: unsupported kernel radius"); return MI355_ERR_FAILED; }
        }
        {
            ProfScope ps(ctx, "extrema", level_bytes * 6.0, st);
            // the streamed test pays off on the big octaves of a full batch; smaller launches do not keep enough rows in flight and stay with the tiled kernel
            const bool xs = ctx->blur_stream && (oc.w & 3) == 0 && oc.w >= ctx->xstream_min_w && oc.h >= ctx->xstream_min_w * 3 / 4 && n >= ctx->xstream_min_frames;
            if (xs) {
                // no row halo to amortise here (3 + XD rows to prime a segment): many short segments balance the wave slots
                const int nstrip = (oc.w + XSW - 1) / XSW;
                const int xsw = ((oc.w + nstrip - 1) / nstrip + 3) & ~3;      // equal strips (<= 248 columns) instead of a nearly empty last one
                // whole rounds of the 1024 x XWAVES wave slots: the largest k <= 2 whose segments stay >= 64 rows
                int nseg = 1, L = oc.h;
                for (int k = 2; k >= 1; k--) {
                    const int ns = (1024 * XWAVES * k) / (nstrip * n);
                    if (ns < 1) continue;
                    const int l = (oc.h + ns - 1) / ns;
                    if (l >= 64 || k == 1) { L = l < 64 ? 64 : l; break; }
                }
                nseg = (oc.h + L - 1) / L;
                hipLaunchKernelGGL(extrema_stream, dim3((nstrip * nseg * n + 3) / 4), dim3(256), 0, st,
                                   oc, o, s->cand.as<unsigned long long>(), s->ccnt.as<unsigned>(), s->cand_cap, cnt + 4, bs, s->cube.as<float>(), s->cube_cap, L, nstrip, nseg, n, xsw);
            } else {
                hipLaunchKernelGGL(extrema_kernel, dim3(((oc.w + EW - 1) / EW) * ((oc.h + EH - 1) / EH), n), dim3(256), 0, st,
                                   oc, o, s->cand.as<unsigned long long>(), s->ccnt.as<unsigned>(), s->cand_cap, cnt + 4, bs, s->cube.as<float>(), s->cube_cap);
            }
        }
    }
    if (serial_heavy) {
        if (!ctx->heavy_ev) MI_HIP(hipEventCreateWithFlags(&ctx->heavy_ev, hipEventDisableTiming));
        MI_HIP(hipEventRecord(ctx->heavy_ev, st)); ctx->heavy_ev_valid = true;
    }
    // ---- phase 3: keypoint stages of all n frames ----
    {
        ProfScope ps(ctx, "refine", 0.0, tt);
        static const int refine_gx = [] { const char* e = getenv("MI355_REFINE_GX"); return e ? atoi(e) : 2; }();      // workgroups per candidate region: 32 x 64 regions x frames of mostly empty workgroups cost more to dispatch than the fits
        hipLaunchKernelGGL(refine_kernel, dim3(refine_gx, NREG, n), dim3(256), 0, tt, s->P, s->cand.as<unsigned long long>(), s->ccnt.as<unsigned>(), s->cand_cap, cnt + 0,
                           ctx->p.contrast_threshold, ctx->p.edge_threshold, 1.6f, s->refined.as<Refined>(), cnt + 1, s->ref_cap, s->rhist.as<unsigned>(), bs, s->cube.as<float>(), s->cube_cap);
    }
    if (s->keepall) {
        // keep-all: every refined point that holds its location's smallest start key is oriented; the keypoints leave in generation order
        {
            ProfScope ps(ctx, "kp_select", 0.0, tt);
            hipLaunchKernelGGL(keepall_live_kernel, dim3(256, n), dim3(256), 0, tt, s->P, s->refined.as<Refined>(), cnt + 1, s->ref_cap, cnt + 8, s->olist.as<unsigned>(), bs);
            hipLaunchKernelGGL(keepall_reset_kernel, dim3(256, n), dim3(256), 0, tt, s->P, s->refined.as<Refined>(), cnt + 1, s->ref_cap, bs);
        }
        {
            ProfScope ps(ctx, "orient", 0.0, tt);
            hipLaunchKernelGGL(orient_kernel, dim3(ctx->num_cu * 4, n), dim3(256), 0, tt, s->P, s->refined.as<Refined>(), cnt + 1, s->ref_cap,
                               s->kps.as<KpRec>(), s->kresp.as<unsigned>(), cnt + 2, s->kp_cap, cnt + 8, 0, bs, s->olist.as<unsigned>());
        }
        {
            ProfScope ps(ctx, "topk", 0.0, tt);
            hipLaunchKernelGGL(keepall_output_kernel, dim3((KEEPALL_MAX + 255) / 256, n), dim3(256), 0, tt, s->kps.as<KpRec>(), cnt + 2, s->kp_cap, outs, s->sel.as<SelRec>(),
                               reinterpret_cast<int*>(cnt + 3), reinterpret_cast<int*>(cnt + 4), bs);
        }
    } else {
    {
        ProfScope ps(ctx, "kp_select", 0.0, tt);
        hipLaunchKernelGGL(select_unclaim_kernel, dim3(1 + UNCLAIM_WGS, n), dim3(1024), 0, tt, s->P, s->refined.as<Refined>(), bs, s->rhist.as<unsigned>(), cnt + 1, s->ref_cap, (unsigned)nf + 256u, cnt + 8,
                           bs.refined, s->olist.as<unsigned>());
    }
    for (int pass = 0; pass < 2; pass++) {       // pass 1 (everything below the response threshold) exits at once unless top-k asked for it
        {
            ProfScope ps(ctx, "orient", 0.0, tt);
            hipLaunchKernelGGL(orient_kernel, dim3(ctx->num_cu * 4, n), dim3(256), 0, tt, s->P, s->refined.as<Refined>(), cnt + 1, s->ref_cap,
                               s->kps.as<KpRec>(), s->kresp.as<unsigned>(), cnt + 2, s->kp_cap, cnt + 8, pass, bs, s->olist.as<unsigned>());
        }
        {
            ProfScope ps(ctx, "topk", 0.0, tt);
            hipLaunchKernelGGL(topk_kernel, dim3(n), dim3(1024), 0, tt, s->kps.as<KpRec>(), s->kresp.as<unsigned>(), cnt + 2, s->kp_cap, nf,
                               outs, s->sel.as<SelRec>(), reinterpret_cast<int*>(cnt + 3), reinterpret_cast<int*>(cnt + 4), cnt + 8, pass, bs);
        }
    }
    }
    {
        ProfScope ps(ctx, "describe", 0.0, tt);
        hipLaunchKernelGGL(describe_kernel, dim3(((int)bs.sel + 3) / 4, n), dim3(256), 0, tt, s->P, s->sel.as<SelRec>(), reinterpret_cast<const int*>(cnt + 3), outs, bs);
    }
    MI_HIP(hipGetLastError());
    // the matcher's operands of all n frames in one launch, their counters in one strided copy (n launches + n copies of ~5 us each kept
    // the batch's stream, and a pipeline slot, busy for 0.3 ms per batch of 32)
    { int rc = mi_finish_features_batch(ctx, fs.data(), n, reinterpret_cast<const int*>(cnt + 3), (int)CNT_STRIDE, tt, s->keepall ? KEEPALL_MAX : 2048); if (rc != MI355_OK) return rc; }
    {
        if (ctx->pinned_used % PINNED_CHUNK + (size_t)n > PINNED_CHUNK) ctx->pinned_used += PINNED_CHUNK - ctx->pinned_used % PINNED_CHUNK;   // n slots in one chunk
        int* h0 = nullptr;
        for (int k = 0; k < n; k++) {
            int* hc = pinned_slot(ctx);
            if (!hc) { ctx->set_error("sift: pinned alloc failed"); return MI355_ERR_NOMEM; }
            if (k == 0) h0 = hc;
            Features& f = *fs[k];
            f.h_cnt = hc; f.pending = true; f.n = 0;
            f.caps[0] = 0xffffffffu; f.caps[1] = s->ref_cap; f.caps[2] = s->kp_cap;      // candidate overflow is flagged by the kernel (cnt[4])
        }
        MI_HIP(hipMemcpy2DAsync(h0, 8 * sizeof(int), cnt, CNT_STRIDE * sizeof(unsigned), 8 * sizeof(unsigned), (size_t)n, hipMemcpyDeviceToHost, tt));
    }
    MI_HIP(hipEventRecord(s->done, tt));
    {                                                       // the batch's own event: mi_resolve_features_of() waits for it, not for the streams
        if (ctx->batch_events_used >= ctx->batch_events.size()) {
            hipEvent_t e = nullptr;
            MI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->batch_events.push_back(e);
        }
        hipEvent_t e = ctx->batch_events[ctx->batch_events_used++];
        MI_HIP(hipEventRecord(e, tt));
        for (int k = 0; k < n; k++) fs[k]->ready = e;
    }
    for (int k = 0; k < n; k++) if (pend[k].ev) MI_HIP(hipEventRecord(pend[k].ev, tt));
    return MI355_OK;
}

// a frame parked with event `ev` still waits for its batch to fill: enqueue that batch now (the event gets recorded)
int mi_sift_flush_if_parked(mi355_ctx* ctx, hipEvent_t ev) {
    for (SiftWork* s : ctx->sift_slots) {
        if (!s) continue;
        for (const SiftWork::Pend& p : s->pend) if (p.ev == ev) return sift_run_batch(ctx, s);
    }
    return MI355_OK;
}
