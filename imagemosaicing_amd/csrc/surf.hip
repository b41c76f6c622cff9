// csrc/surf.hip -- the SURF variant of the path on gfx950: replaces the body of GetMatchedPairsOneToAllSurf
// (MosaicWithoutPos.cpp:5300-5533): SurfFeatureDetector(minHessian).detect + SurfDescriptorExtractor.compute (:5313-5335),
// FlannBasedMatcher.match on the 128-float descriptors (:5389-5391, here the exact 1-NN it approximates), std::sort (:5392), the
// distance-threshold selection (:5400-5424), CMosaicHarris::Ransac (:5459 -- the same arithmetic as Ransac2D, see
// tests/test_surf.py; csrc/ransac.hip) and the acceptance test (> 18 inliers, :5306, :5497).
//
// The algorithm is SURF (Bay et al. 2008) with the parameters of the cv::SURF object the reference builds (4 octaves, 2 layers,
// extended 128-float oriented descriptors); the exact arithmetic is the one stated at the top of oracle/oracle_surf.c (OpenCV
// 2.4.0's is not available: PARITY UNPINNED) -- this file implements that definition for the GPU and the parity tests compare
// keypoints and descriptors bit for bit.
//
// Kernels (all HBM / latency bound integer or short float work; no matrix shapes here):
//   surf_gray_rows      fixed-point gray + per-row running sums (wave scan), then surf_cols adds the rows up: the integral image,
//                       32-bit sums modulo 2^32 (a box sum is a difference and stays exact)
//   surf_det            one lane per sample and layer: 10 box sums from 40 integral-image reads, det / trace of the box Hessian
//   surf_maxima         26-neighbour maxima of the two middle layers of every octave + the 3x3 interpolation; a survivor leaves
//                       as ONE 64-bit key (response bits descending | octave, layer, row, column): the order is total and
//                       independent of the append order, and the few keypoints kept are re-derived from their key afterwards
//   surf_sort_*         bitonic sort of the keys (stages below 2048 apart inside LDS)
//   surf_finalize       the strongest max_kp keys -> keypoints (position, size, Laplacian sign)
//   surf_orient         one wave per keypoint: 113 Haar samples on a disc, compacted in disc order, 72 window sums (a lane per
//                       window walks the samples in order: sequential float sums, as defined)
//   surf_describe       one workgroup per keypoint: 21 x 21 area-averaged patch of the rotated window (a lane per cell), Haar
//                       differences, 4 x 4 x 8 sums (a lane per sum, samples in order), normalisation
//   surf_bf / surf_select   exact float 1-NN (a lane per query, train rows broadcast from LDS, fmaf chain over the 128 dims in
//                       order), sort by (distance, query) and the threshold walk
#include "common.h"
#include "detmath.h"
#include <cfloat>
#include <cmath>

namespace {

constexpr int S_OCT = 4, S_LAY = 2, S_NL = S_OCT * (S_LAY + 2);      // 16 layers
constexpr int ORI_R = 6, ORI_WIN = 60, ORI_INC = 5, PATCH = 20;
constexpr int SURF_MAX_KP = 1 << 21;      // = CAND_CAP: with max_kp at this value every Hessian maximum the candidate list can hold is kept, like the reference (MosaicWithoutPos.cpp:5313-5335)
constexpr unsigned CAND_CAP = 1u << 21;

struct SBox { int x1, y1, x2, y2; float w; };
struct LayerDesc {
    int size, step, rows, cols, si, sj, margin;      // samples si x sj written at (+margin, +margin)
    size_t off;                                      // offset of the layer's det (and trace) plane, in floats
    SBox dx[3], dy[3], dxy[4];
};
struct SurfLayers { LayerDesc l[S_NL]; };

__host__ void stretch(const int proto[][5], int n, int old_size, int size, SBox* out) {
    const float ratio = (float)size / (float)old_size;
    for (int k = 0; k < n; k++) {
        out[k].x1 = (int)rintf(ratio * (float)proto[k][0]); out[k].y1 = (int)rintf(ratio * (float)proto[k][1]);
        out[k].x2 = (int)rintf(ratio * (float)proto[k][2]); out[k].y2 = (int)rintf(ratio * (float)proto[k][3]);
        out[k].w = (float)proto[k][4] / ((float)(out[k].x2 - out[k].x1) * (float)(out[k].y2 - out[k].y1));
    }
}
// device twin for the orientation wavelets (size depends on the keypoint)
__device__ __forceinline__ void stretch2(const int (&proto)[2][5], int size, SBox* out) {
    const float ratio = (float)size / 4.0f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        out[k].x1 = (int)rintf(ratio * (float)proto[k][0]); out[k].y1 = (int)rintf(ratio * (float)proto[k][1]);
        out[k].x2 = (int)rintf(ratio * (float)proto[k][2]); out[k].y2 = (int)rintf(ratio * (float)proto[k][3]);
        out[k].w = (float)proto[k][4] / ((float)(out[k].x2 - out[k].x1) * (float)(out[k].y2 - out[k].y1));
    }
}
const int DX_P[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
const int DY_P[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
const int DXY_P[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};

template <int N>
__device__ __forceinline__ float haar(const uint32_t* S, int sw, int x, int y, const SBox* f) {
    double d = 0.0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t a = S[(size_t)(y + f[k].y1) * sw + x + f[k].x1], b = S[(size_t)(y + f[k].y1) * sw + x + f[k].x2];
        const uint32_t c = S[(size_t)(y + f[k].y2) * sw + x + f[k].x1], e = S[(size_t)(y + f[k].y2) * sw + x + f[k].x2];
        const int box = (int)(a + e - b - c);
        d += (double)box * (double)f[k].w;
    }
    return (float)d;
}

// ---- integral image ---------------------------------------------------------------------------------------------------------------
// one wave per row: gray bytes out, inclusive running sums of the row into S[(y+1)][1..w]; S row 0 / column 0 are zero
__global__ __launch_bounds__(256) void surf_gray_rows(const uint8_t* bgr, int ws, int w, int h, uint8_t* gray, uint32_t* S) {
    const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int sw = w + 1;
    const uint8_t* row = bgr + (size_t)y * ws;
    uint32_t* srow = S + (size_t)(y + 1) * sw;
    if (lane == 0) srow[0] = 0;
    uint32_t carry = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        uint32_t g = 0;
        if (x < w) {
            const uint8_t* p = row + 3 * x;
            g = (uint32_t)((1868 * (int)p[0] + 9617 * (int)p[1] + 4899 * (int)p[2] + 8192) >> 14);
            gray[(size_t)y * w + x] = (uint8_t)g;
        }
        uint32_t v = g;                                   // inclusive wave scan (integers: any order)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(v, off); if (lane >= off) v += t; }
        if (x < w) srow[x + 1] = carry + v;
        carry += __shfl(v, 63);
    }
    if (y == 0 && lane == 0) S[0] = 0;
}
// one lane per column: running sum down the rows, 8 independent loads in flight
__global__ __launch_bounds__(256) void surf_cols(uint32_t* S, int sw, int sh) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= sw) return;
    S[x] = 0;
    uint32_t acc = 0;
    int y = 1;
    for (; y + 8 <= sh; y += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = S[(size_t)(y + u) * sw + x];
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += v[u]; S[(size_t)(y + u) * sw + x] = acc; }
    }
    for (; y < sh; y++) { acc += S[(size_t)y * sw + x]; S[(size_t)y * sw + x] = acc; }
}

// ---- fast Hessian -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void surf_det(const uint32_t* S, int sw, const SurfLayers* L, float* det, float* trace, int z0) {
    const LayerDesc& q = L->l[z0 + blockIdx.z];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= q.si || j >= q.sj) return;
    const float vx = haar<3>(S, sw, j * q.step, i * q.step, q.dx);
    const float vy = haar<3>(S, sw, j * q.step, i * q.step, q.dy);
    const float vxy = haar<4>(S, sw, j * q.step, i * q.step, q.dxy);
    const size_t idx = q.off + (size_t)(i + q.margin) * q.cols + (j + q.margin);
    det[idx] = vx * vy - (0.81f * vxy) * vxy;
    trace[idx] = vx + vy;
}

// octave 0 (sampling step 1, sizes 9 / 15 / 21 / 27): the four layers of a 64 x 16 block of samples read the same 92 x 44 window of
// the integral image 160 times per sample -- staged in LDS once (16 KB) instead of 160 gathers through L1 per sample.  Same boxes,
// same double accumulation as surf_det.
constexpr int DT_W = 64, DT_H = 16, DT_HALO = 28, DT_TW = DT_W + DT_HALO, DT_TH = DT_H + DT_HALO;
template <int N>
__device__ __forceinline__ float haar_tile(const uint32_t* t, int x, int y, const SBox* f) {
    double d = 0.0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t a = t[(y + f[k].y1) * DT_TW + x + f[k].x1], b = t[(y + f[k].y1) * DT_TW + x + f[k].x2];
        const uint32_t c = t[(y + f[k].y2) * DT_TW + x + f[k].x1], e = t[(y + f[k].y2) * DT_TW + x + f[k].x2];
        const int box = (int)(a + e - b - c);
        d += (double)box * (double)f[k].w;
    }
    return (float)d;
}
__global__ __launch_bounds__(256) void surf_det_tile(const uint32_t* S, int sw, int sh, const SurfLayers* L, float* det, float* trace) {
    __shared__ uint32_t t[DT_TH * DT_TW];
    const int x0 = blockIdx.x * DT_W, y0 = blockIdx.y * DT_H;
    for (int e = threadIdx.x; e < DT_TH * DT_TW; e += 256) {
        const int r = e / DT_TW, c = e - r * DT_TW;
        const int y = y0 + r, x = x0 + c;
        t[e] = (y < sh && x < sw) ? S[(size_t)y * sw + x] : 0u;
    }
    __syncthreads();
    const int lj = threadIdx.x & 63, j = x0 + lj;
#pragma unroll
    for (int l = 0; l < S_LAY + 2; l++) {
        const LayerDesc& q = L->l[l];                      // octave 0: step 1, size <= 27
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int li = (threadIdx.x >> 6) * 4 + k, i = y0 + li;
            if (i >= q.si || j >= q.sj) continue;
            const float vx = haar_tile<3>(t, lj, li, q.dx);
            const float vy = haar_tile<3>(t, lj, li, q.dy);
            const float vxy = haar_tile<4>(t, lj, li, q.dxy);
            const size_t idx = q.off + (size_t)(i + q.margin) * q.cols + (j + q.margin);
            det[idx] = vx * vy - (0.81f * vxy) * vxy;
            trace[idx] = vx + vy;
        }
    }
}

__device__ void solve3(float A[3][3], float b[3], float x[3]) {        // oracle_surf.c surf_solve3
    int p0 = 0, p1 = 1, p2 = 2;
    {
        const float b0 = fabsf(A[0][0]), b1 = fabsf(A[1][0]), b2 = fabsf(A[2][0]);
        int m = 0; float best = b0;
        if (b1 > best) { best = b1; m = 1; }
        if (b2 > best) { best = b2; m = 2; }
        if (!(best > 1e-30f)) { x[0] = x[1] = x[2] = 0.0f; return; }
        if (m == 1) { p0 = 1; p1 = 0; } else if (m == 2) { p0 = 2; p2 = 0; }
    }
    auto elim = [&](int pr, int pk, int k) {
        const float f = A[pr][k] / A[pk][k];
        for (int c = k + 1; c < 3; c++) A[pr][c] = A[pr][c] - f * A[pk][c];
        b[pr] = b[pr] - f * b[pk];
    };
    elim(p1, p0, 0); elim(p2, p0, 0);
    {
        const float b1 = fabsf(A[p1][1]), b2 = fabsf(A[p2][1]);
        float best = b1;
        if (b2 > best) { best = b2; const int t = p1; p1 = p2; p2 = t; }
        if (!(best > 1e-30f)) { x[0] = x[1] = x[2] = 0.0f; return; }
    }
    elim(p2, p1, 1);
    if (!(fabsf(A[p2][2]) > 1e-30f)) { x[0] = x[1] = x[2] = 0.0f; return; }
    x[2] = b[p2] / A[p2][2];
    x[1] = (b[p1] - A[p1][2] * x[2]) / A[p1][1];
    x[0] = ((b[p0] - A[p0][1] * x[1]) - A[p0][2] * x[2]) / A[p0][0];
}

// interpolation of the maximum at (i, j) of middle layer (o, l); N9 is read from the three det planes.  false: rejected
__device__ bool surf_interp(const SurfLayers* L, const float* det, int o, int l, int i, int j, float& cx, float& cy, float& ksz) {
    const LayerDesc& a = L->l[o * (S_LAY + 2) + l - 1];
    const LayerDesc& b = L->l[o * (S_LAY + 2) + l];
    const LayerDesc& c = L->l[o * (S_LAY + 2) + l + 1];
    float N9[3][9];
    const size_t offs[3] = {a.off, b.off, c.off};
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int di = -1; di <= 1; di++)
#pragma unroll
            for (int dj = -1; dj <= 1; dj++) N9[q][(di + 1) * 3 + (dj + 1)] = det[offs[q] + (size_t)(i + di) * b.cols + (j + dj)];
    const int step = b.step, size = b.size;
    const int sum_i = step * (i - (size / 2) / step), sum_j = step * (j - (size / 2) / step);
    cx = (float)sum_j + (float)(size - 1) * 0.5f; cy = (float)sum_i + (float)(size - 1) * 0.5f;
    float bb[3] = {-(N9[1][5] - N9[1][3]) / 2.0f, -(N9[1][7] - N9[1][1]) / 2.0f, -(N9[2][4] - N9[0][4]) / 2.0f};
    float A[3][3];
    A[0][0] = (N9[1][3] - 2.0f * N9[1][4]) + N9[1][5];
    A[0][1] = (((N9[1][8] - N9[1][6]) - N9[1][2]) + N9[1][0]) / 4.0f;
    A[0][2] = (((N9[2][5] - N9[2][3]) - N9[0][5]) + N9[0][3]) / 4.0f;
    A[1][0] = A[0][1];
    A[1][1] = (N9[1][1] - 2.0f * N9[1][4]) + N9[1][7];
    A[1][2] = (((N9[2][7] - N9[2][1]) - N9[0][7]) + N9[0][1]) / 4.0f;
    A[2][0] = A[0][2]; A[2][1] = A[1][2];
    A[2][2] = (N9[0][4] - 2.0f * N9[1][4]) + N9[2][4];
    float xx[3];
    solve3(A, bb, xx);
    const bool ok = (xx[0] != 0.0f || xx[1] != 0.0f || xx[2] != 0.0f) && fabsf(xx[0]) <= 1.0f && fabsf(xx[1]) <= 1.0f && fabsf(xx[2]) <= 1.0f;
    if (!ok) return false;
    cx = cx + xx[0] * (float)step; cy = cy + xx[1] * (float)step;
    ksz = rintf((float)size + xx[2] * (float)(size - a.size));
    return true;
}

__device__ __forceinline__ unsigned long long surf_key(float v0, int o, int l, int i, int j) {
    // ascending key order = (response descending, octave, layer, row, column); rows / columns below 2^14
    return ((unsigned long long)(~__float_as_uint(v0)) << 32) | ((unsigned long long)o << 29) | ((unsigned long long)(l - 1) << 28) |
           ((unsigned long long)i << 14) | (unsigned long long)j;
}

__global__ __launch_bounds__(256) void surf_maxima(const SurfLayers* L, const float* det, float thr, unsigned long long* keys, unsigned* count, unsigned cap, int m0) {
    // the candidates of a workgroup reserve their slots with ONE global atomic (a single address takes ~1e8 atomics/s; one per
    // candidate cost 0.4-0.8 ms per 12 MP frame); the list is sorted afterwards, so the order inside it is free
    __shared__ unsigned s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int m = m0 + blockIdx.z, o = m / S_LAY, l = 1 + m % S_LAY;
    const LayerDesc& b = L->l[o * (S_LAY + 2) + l];
    const LayerDesc& c = L->l[o * (S_LAY + 2) + l + 1];
    const LayerDesc& a = L->l[o * (S_LAY + 2) + l - 1];
    const int margin = (c.size / 2) / b.step + 1;
    const int j = margin + blockIdx.x * 64 + (threadIdx.x & 63), i = margin + blockIdx.y * 4 + (threadIdx.x >> 6);
    bool cand = false;
    float v0 = 0.0f;
    if (i < b.rows - margin && j < b.cols - margin) {
        v0 = det[b.off + (size_t)i * b.cols + j];
        if (v0 > thr) {
            const size_t offs[3] = {a.off, b.off, c.off};
            bool is_max = true;
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int di = -1; di <= 1; di++)
#pragma unroll
                    for (int dj = -1; dj <= 1; dj++)
                        if (!(q == 1 && di == 0 && dj == 0)) { if (!(v0 > det[offs[q] + (size_t)(i + di) * b.cols + (j + dj)])) is_max = false; }
            if (is_max) {
                float cx, cy, ksz;
                cand = surf_interp(L, det, o, l, i, j, cx, cy, ksz);
            }
        }
    }
    unsigned local = 0;
    if (cand) local = atomicAdd(&s_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(count, s_n);
    __syncthreads();
    if (cand) { const unsigned slot = s_base + local; if (slot < cap) keys[slot] = surf_key(v0, o, l, i, j); }
}

// ---- bitonic sort of the keys (ascending), N a power of two, padding keys are all ones -----------------------------------------------
__global__ __launch_bounds__(1024) void surf_sort_local(unsigned long long* keys, unsigned N, unsigned k_lo, unsigned k_hi, unsigned j_start) {
    // handles, for this workgroup's 2048 keys: every (k, j) with k in [k_lo, k_hi] and j < 2048 (first k: j from j_start)
    __shared__ unsigned long long s[2048];
    const unsigned base = blockIdx.x * 2048u, tid = threadIdx.x;
    s[tid] = keys[base + tid]; s[tid + 1024] = keys[base + tid + 1024];
    __syncthreads();
    for (unsigned k = k_lo; k <= k_hi && k <= N; k <<= 1) {
        unsigned j = (k == k_lo) ? j_start : (k >> 1);
        if (j > 1024) j = 1024;
        for (; j > 0; j >>= 1) {
            const unsigned lo = ((tid & ~(j - 1)) << 1) | (tid & (j - 1)), hi = lo | j;      // tid-th comparator of stride j
            const unsigned long long A = s[lo], B = s[hi];
            const bool up = ((base + lo) & k) == 0;
            if ((A > B) == up) { s[lo] = B; s[hi] = A; }
            __syncthreads();
        }
    }
    keys[base + tid] = s[tid]; keys[base + tid + 1024] = s[tid + 1024];
}
__global__ __launch_bounds__(256) void surf_sort_global(unsigned long long* keys, unsigned N, unsigned k, unsigned j) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= N / 2) return;
    const unsigned lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
    const unsigned long long A = keys[lo], B = keys[hi];
    const bool up = (lo & k) == 0;
    if ((A > B) == up) { keys[lo] = B; keys[hi] = A; }
}
__global__ void surf_pad_keys(unsigned long long* keys, const unsigned* count, unsigned cap, unsigned N) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned n = *count; if (n > cap) n = cap;
    if (i < N && i >= n) keys[i] = ~0ull;
}

struct SurfKp { float x, y, size, angle, response; int octave, lap, valid; };

__global__ __launch_bounds__(256) void surf_finalize(const SurfLayers* L, const float* det, const float* trace, const unsigned long long* keys,
                                                     const unsigned* count, unsigned cap, int max_kp, SurfKp* out, int* n_out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    unsigned n = *count; if (n > cap) n = cap;
    const int keep = (int)n < max_kp ? (int)n : max_kp;
    if (q == 0) *n_out = keep;
    if (q >= keep) return;
    const unsigned long long key = keys[q];
    const int o = (int)((key >> 29) & 7u), l = 1 + (int)((key >> 28) & 1u), i = (int)((key >> 14) & 0x3fffu), j = (int)(key & 0x3fffu);
    SurfKp k;
    float cx = 0, cy = 0, ksz = 0;
    (void)surf_interp(L, det, o, l, i, j, cx, cy, ksz);               // accepted once already: same values
    const LayerDesc& b = L->l[o * (S_LAY + 2) + l];
    const float tr = trace[b.off + (size_t)i * b.cols + j];
    k.x = cx; k.y = cy; k.size = ksz; k.angle = 0.0f; k.response = __uint_as_float(~(unsigned)(key >> 32)); k.octave = o;
    k.lap = tr > 0.0f ? 1 : (tr < 0.0f ? -1 : 0); k.valid = 1;
    out[q] = k;
}

// ---- orientation ------------------------------------------------------------------------------------------------------------------
struct OriTable { int n; int x[128], y[128]; float w[128]; };

__global__ __launch_bounds__(256) void surf_orient(const uint32_t* S, int w, int h, const OriTable* T, SurfKp* kps, const int* n_kp) {
    __shared__ float sX[4][128], sY[4][128];
    __shared__ int sA[4][128];
    __shared__ float sMod[4][72], sSx[4][72], sSy[4][72];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wv;
    if (k >= *n_kp) return;
    SurfKp kp = kps[k];
    const int sw = w + 1;
    const float s = kp.size * 1.2f / 9.0f;
    const int gws = 2 * (int)rintf(2.0f * s);
    if (h + 1 < gws || w + 1 < gws) { if (lane == 0) kps[k].valid = 0; return; }
    const int OX[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}}, OY[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
    SBox ox[2], oy[2];
    stretch2(OX, gws, ox); stretch2(OY, gws, oy);
    // samples in disc order, compacted with ballots (two rounds of 64)
    int na = 0;
    for (int r0 = 0; r0 < T->n; r0 += 64) {
        const int q = r0 + lane;
        bool ok = false; float X = 0, Y = 0; int A = 0;
        if (q < T->n) {
            const int x = (int)rintf(kp.x + (float)T->x[q] * s - (float)(gws - 1) / 2.0f);
            const int y = (int)rintf(kp.y + (float)T->y[q] * s - (float)(gws - 1) / 2.0f);
            if (!(y < 0 || y >= (h + 1) - gws || x < 0 || x >= (w + 1) - gws)) {
                const float vx = haar<2>(S, sw, x, y, ox), vy = haar<2>(S, sw, x, y, oy);
                X = vx * T->w[q]; Y = vy * T->w[q];
                A = (int)rintf(det_atan2deg(Y, X));
                ok = true;
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        if (ok) { const int pos = na + __builtin_popcountll(m & ((1ull << lane) - 1ull)); sX[wv][pos] = X; sY[wv][pos] = Y; sA[wv][pos] = A; }
        na += __builtin_popcountll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (na == 0) { if (lane == 0) kps[k].valid = 0; return; }
    // a lane per window position (72 = 64 + 8), samples in order
    for (int wi = lane; wi < 360 / ORI_INC; wi += 64) {
        const int i = wi * ORI_INC;
        float sx = 0.0f, sy = 0.0f;
        for (int j = 0; j < na; j++) {
            int d = sA[wv][j] - i; d = d < 0 ? -d : d;
            if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sx = sx + sX[wv][j]; sy = sy + sY[wv][j]; }
        }
        sSx[wv][wi] = sx; sSy[wv][wi] = sy; sMod[wv][wi] = sx * sx + sy * sy;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float bestx = 0.0f, besty = 0.0f, best = 0.0f;
        for (int wi = 0; wi < 360 / ORI_INC; wi++) if (sMod[wv][wi] > best) { best = sMod[wv][wi]; bestx = sSx[wv][wi]; besty = sSy[wv][wi]; }
        kps[k].angle = det_atan2deg(-besty, bestx);
    }
}

// ---- descriptor -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void surf_describe(const uint8_t* gray, int w, int h, const float* DW, const SurfKp* kps, const int* n_kp, float* desc) {
    __shared__ float patch[(PATCH + 1) * (PATCH + 1)];
    __shared__ float sDX[PATCH * PATCH], sDY[PATCH * PATCH];
    __shared__ float sV[128];
    __shared__ float sScale;
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= *n_kp) return;
    const SurfKp kp = kps[k];
    float* out = desc + (size_t)k * 128;
    if (!kp.valid) { if (tid < 128) out[tid] = 0.0f; return; }
    const float s = kp.size * 1.2f / 9.0f;
    const int win = (int)((float)(PATCH + 1) * s);
    float sn, cs;
    det_sincosdeg(kp.angle, sn, cs);
    const float sin_dir = -sn, cos_dir = cs;
    const float off = -(float)(win - 1) / 2.0f;
    const float start_x = kp.x + off * cos_dir + off * sin_dir, start_y = kp.y - off * sin_dir + off * cos_dir;
    const float cell = (float)win / (float)(PATCH + 1);
    for (int c = tid; c < (PATCH + 1) * (PATCH + 1); c += 256) {
        const int pi = c / (PATCH + 1), pj = c - pi * (PATCH + 1);
        const float r0 = (float)pi * cell, r1 = (float)(pi + 1) * cell, c0 = (float)pj * cell, c1 = (float)(pj + 1) * cell;
        int ia = (int)floorf(r0), ib = (int)ceilf(r1) - 1, ja = (int)floorf(c0), jb = (int)ceilf(c1) - 1;
        if (ib > win - 1) ib = win - 1;
        if (jb > win - 1) jb = win - 1;
        float acc = 0.0f, wsum = 0.0f;
        for (int i = ia; i <= ib; i++) {
            const float lo = (float)i > r0 ? (float)i : r0, hi = (float)(i + 1) < r1 ? (float)(i + 1) : r1;
            const float wy = hi - lo;
            for (int j = ja; j <= jb; j++) {
                const float lo2 = (float)j > c0 ? (float)j : c0, hi2 = (float)(j + 1) < c1 ? (float)(j + 1) : c1;
                const float wgt = wy * (hi2 - lo2);
                const float px = (start_x + (float)i * sin_dir) + (float)j * cos_dir;
                const float py = (start_y + (float)i * cos_dir) - (float)j * sin_dir;
                int xi = (int)rintf(px), yi = (int)rintf(py);
                xi = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi);
                yi = yi < 0 ? 0 : (yi > h - 1 ? h - 1 : yi);
                acc = fmaf((float)gray[(size_t)yi * w + xi], wgt, acc);
                wsum = wsum + wgt;
            }
        }
        const float v = rintf(acc / wsum);
        patch[c] = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    }
    __syncthreads();
    for (int c = tid; c < PATCH * PATCH; c += 256) {
        const int i = c / PATCH, j = c - i * PATCH;
        const float dw = DW[c];
        const float p00 = patch[i * (PATCH + 1) + j], p01 = patch[i * (PATCH + 1) + j + 1], p10 = patch[(i + 1) * (PATCH + 1) + j], p11 = patch[(i + 1) * (PATCH + 1) + j + 1];
        sDX[c] = (((p01 - p00) + p11) - p10) * dw;
        sDY[c] = (((p10 - p00) + p11) - p01) * dw;
    }
    __syncthreads();
    if (tid < 128) {
        const int cellid = tid >> 3, q = tid & 7, ci = cellid >> 2, cj = cellid & 3;
        float v = 0.0f;
        for (int y = ci * 5; y < ci * 5 + 5; y++)
            for (int x = cj * 5; x < cj * 5 + 5; x++) {
                const float tx = sDX[y * PATCH + x], ty = sDY[y * PATCH + x];
                // v[0] += tx, v[1] += |tx| when ty >= 0, else v[2], v[3]; v[4] += ty, v[5] += |ty| when tx >= 0, else v[6], v[7]
                if (q < 4) { if ((ty >= 0) == (q < 2)) v = v + ((q & 1) ? fabsf(tx) : tx); }
                else { if ((tx >= 0) == (q < 6)) v = v + ((q & 1) ? fabsf(ty) : ty); }
            }
        sV[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double sq = 0.0;
        for (int q = 0; q < 128; q++) sq += (double)sV[q] * (double)sV[q];
        sScale = (float)(1.0 / (sqrt(sq) + DBL_EPSILON));
    }
    __syncthreads();
    if (tid < 128) out[tid] = sV[tid] * sScale;
}

// keypoints without an orientation sample are dropped: order-preserving compaction of keypoints + descriptors (one workgroup)
__global__ __launch_bounds__(1024) void surf_compact(const SurfKp* kps, const int* n_in, mi355_keypoint* kp_out, float2* xy, int* pos_out, int* n_out) {
    __shared__ int s_base, s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = *n_in;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const bool ok = i < n && kps[i].valid != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        if (lane == 0) s_w[wv] = __builtin_popcountll(m);
        __syncthreads();
        int off = s_base, tot = 0;
        for (int q = 0; q < 16; q++) { if (q < wv) off += s_w[q]; tot += s_w[q]; }
        if (ok) {
            const int pos = off + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            const SurfKp k = kps[i];
            mi355_keypoint o; o.x = k.x; o.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.lap;
            kp_out[pos] = o; xy[pos] = make_float2(k.x, k.y);
            pos_out[i] = pos;
        } else if (i < n) pos_out[i] = -1;
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}

// the 512-byte descriptor rows move with one workgroup per two keypoints, coalesced (a single workgroup copying them row by row
// per lane took 1.35 ms for 8192 keypoints -- the largest kernel of the extraction)
__global__ __launch_bounds__(256) void surf_move_desc(const float* desc, const int* pos, const int* n_in, float* desc_out) {
    const int i = blockIdx.x * 2 + (threadIdx.x >> 7), t = threadIdx.x & 127;
    if (i >= *n_in) return;
    const int p = pos[i];
    if (p >= 0) desc_out[(size_t)p * 128 + t] = desc[(size_t)i * 128 + t];
}

// ---- pair stage ---------------------------------------------------------------------------------------------------------------------
struct SPair { const float* d_i; const float2* xy_i; int n_i; const float* d_j; const float2* xy_j; int n_j; int img_i, img_j; };

// a lane per query: the train rows stream through LDS (64 rows x 128 floats), every lane reads the same row (broadcast)
__global__ __launch_bounds__(256) void surf_bf(const SPair* pairs, int* nn_idx, float* nn_dist, int stride /* queries per pair in the nn arrays */) {
    __shared__ float s_t[64 * 128];
    const SPair pd = pairs[blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= pd.n_i) return;
    float qv[128];
    const bool q_ok = q < pd.n_i;
#pragma unroll
    for (int k = 0; k < 128; k += 4) {
        float4 v = make_float4(0, 0, 0, 0);
        if (q_ok) v = *reinterpret_cast<const float4*>(pd.d_i + (size_t)q * 128 + k);
        qv[k] = v.x; qv[k + 1] = v.y; qv[k + 2] = v.z; qv[k + 3] = v.w;
    }
    float best = __builtin_inff(); int bi = -1;
    for (int t0 = 0; t0 < pd.n_j; t0 += 64) {
        const int nt = pd.n_j - t0 < 64 ? pd.n_j - t0 : 64;
        __syncthreads();
        for (int e = threadIdx.x; e < nt * 32; e += 256) reinterpret_cast<float4*>(s_t)[e] = reinterpret_cast<const float4*>(pd.d_j + (size_t)t0 * 128)[e];
        __syncthreads();
        for (int t = 0; t < nt; t++) {
            const float* tr = s_t + t * 128;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 128; k++) { const float df = qv[k] - tr[k]; acc = fmaf(df, df, acc); }
            if (acc < best) { best = acc; bi = t0 + t; }
        }
    }
    if (q_ok) { nn_idx[(size_t)blockIdx.y * stride + q] = bi; nn_dist[(size_t)blockIdx.y * stride + q] = sqrtf(best); }
}

// walk the thresholds, then sort what is left by (distance, query) (MosaicWithoutPos.cpp:5392, 5400-5424); one workgroup per pair.
// The reference sorts ALL matches and counts the ones below distT for distT = match_dist, match_dist - 0.05, ... until at most max_features
// are left; the survivors are the head of the sorted list.  Counting needs no order: every threshold is one pass of the workgroup over the
// pair's matches (a few dozen per thread), and only the <= 400 survivors are sorted -- the same list as the head of the full sort
// (which took 8 bytes of LDS per keypoint and capped the keypoints per image at 8192).
__global__ __launch_bounds__(1024) void surf_select(const SPair* pairs, const int* nn_idx, const float* nn_dist, int stride, float match_dist, int max_features,
                                                    mi355_sfpoint* sel1, mi355_sfpoint* sel2, int* nsel) {
    constexpr int NS = 512;                                           // >= MI355_MAX_SELECTED, a power of two
    static_assert(NS >= MI355_MAX_SELECTED, "survivor list");
    __shared__ unsigned long long s_key[NS];
    __shared__ int s_w[16], s_n;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SPair pd = pairs[pair];
    const int M = pd.n_j > 0 ? pd.n_i : 0;
    const size_t o = (size_t)pair * stride;
    float distT = match_dist;
    int cnt;
    for (;;) {                                                        // (the host bounds match_dist: the walk ends)
        int c = 0;
        for (int i = tid; i < M; i += 1024) c += (nn_idx[o + i] >= 0 && nn_dist[o + i] < distT) ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        __syncthreads();
        if (lane == 0) s_w[wv] = c;
        __syncthreads();
        cnt = 0;
        for (int q = 0; q < 16; q++) cnt += s_w[q];
        if (cnt <= max_features) break;
        distT = (float)((double)distT - 0.05);
    }
    if (cnt > MI355_MAX_SELECTED) cnt = MI355_MAX_SELECTED;          // (max_features <= 400: never)
    if (tid == 0) s_n = 0;
    for (int i = tid; i < NS; i += 1024) s_key[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < M; i += 1024)
        if (nn_idx[o + i] >= 0 && nn_dist[o + i] < distT) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < NS) s_key[slot] = ((unsigned long long)__float_as_uint(nn_dist[o + i]) << 32) | (unsigned)i;      // distances >= 0: bit order = value order
        }
    __syncthreads();
    for (int k = 2; k <= NS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (tid < NS) {
                const int i = tid, ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long A = s_key[i], B = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((A > B) == up) { s_key[i] = B; s_key[ixj] = A; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < cnt; i += 1024) {
        const int q = (int)(unsigned)(s_key[i] & 0xffffffffull), t = nn_idx[o + q];
        const float2 a = pd.xy_i[q], b = pd.xy_j[t];
        mi355_sfpoint p1, p2; p1.x = a.x; p1.y = a.y; p1.id = q; p2.x = b.x; p2.y = b.y; p2.id = t;
        sel1[(size_t)pair * MI355_MAX_SELECTED + i] = p1; sel2[(size_t)pair * MI355_MAX_SELECTED + i] = p2;
    }
    if (tid == 0) nsel[pair] = cnt;
}

__global__ void surf_finalize_pairs(const SPair* pairs, const int* nsel, int n_pairs, int min_inliers, mi355_pair_result* out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    out[p].i = pairs[p].img_i; out[p].j = pairs[p].img_j; out[p].n_selected = nsel[p];
    out[p].accepted = out[p].n_in > min_inliers ? 1 : 0;                 // MosaicWithoutPos.cpp:5306, 5497
}

}  // namespace

// device-resident SURF features of one image
struct SurfFeatures { int n = 0, w = 0, h = 0; DevBuf kp, desc, xy; void release() { kp.release(); desc.release(); xy.release(); } };
struct SurfState { std::unordered_map<int, SurfFeatures> feats; };

static SurfState* surf_state(mi355_ctx* ctx) {
    if (!ctx->surf) ctx->surf = new SurfState();
    return ctx->surf;
}
void mi_surf_release(mi355_ctx* ctx) {
    if (!ctx->surf) return;
    for (auto& kv : ctx->surf->feats) kv.second.release();
    delete ctx->surf;
    ctx->surf = nullptr;
}

static int surf_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int ws, float thr, int max_kp, int* n_kp) {
    if (max_kp < 1 || max_kp > SURF_MAX_KP) { ctx->set_error("surf: max_kp must be in [1, 2097152]"); return MI355_ERR_ARG; }
    if (w < 16 || h < 16 || w >= (1 << 14) || h >= (1 << 14) || ws < 3 * w) { ctx->set_error("surf: image geometry (16 <= w, h < 16384)"); return MI355_ERR_ARG; }
    const hipStream_t st = ctx->stream;
    const int sw = w + 1, sh = h + 1;
    // ---- layers ----
    SurfLayers L;
    memset(&L, 0, sizeof(L));
    size_t plane = 0;
    int max_si = 0, max_sj = 0, max_rows = 0, max_cols = 0;
    for (int o = 0; o < S_OCT; o++)
        for (int l = 0; l < S_LAY + 2; l++) {
            LayerDesc& q = L.l[o * (S_LAY + 2) + l];
            q.size = (9 + 6 * l) << o; q.step = 1 << o;
            q.rows = (sh - 1) / q.step; q.cols = (sw - 1) / q.step;
            q.off = plane; plane += (((size_t)q.rows * q.cols + 1) + 63) & ~(size_t)63;
            if (q.size > sh - 1 || q.size > sw - 1) { q.si = q.sj = 0; continue; }
            stretch(DX_P, 3, 9, q.size, q.dx); stretch(DY_P, 3, 9, q.size, q.dy); stretch(DXY_P, 4, 9, q.size, q.dxy);
            q.si = 1 + (sh - 1 - q.size) / q.step; q.sj = 1 + (sw - 1 - q.size) / q.step;
            q.margin = (q.size / 2) / q.step;
            if (q.si > max_si) max_si = q.si;
            if (q.sj > max_sj) max_sj = q.sj;
            if (q.rows > max_rows) max_rows = q.rows;
            if (q.cols > max_cols) max_cols = q.cols;
        }
    DevBuf& dgray = ctx->buf("surf_gray"); DevBuf& dS = ctx->buf("surf_integral"); DevBuf& ddet = ctx->buf("surf_det"); DevBuf& dtr = ctx->buf("surf_trace");
    DevBuf& dL = ctx->buf("surf_layers"); DevBuf& dkeys = ctx->buf("surf_keys"); DevBuf& dcnt = ctx->buf("surf_counts");
    DevBuf& dkps = ctx->buf("surf_kps"); DevBuf& ddesc = ctx->buf("surf_desc_tmp"); DevBuf& dtab = ctx->buf("surf_tables");
    MI_HIP(dgray.reserve((size_t)w * h)); MI_HIP(dS.reserve((size_t)sw * sh * 4)); MI_HIP(ddet.reserve(plane * 4)); MI_HIP(dtr.reserve(plane * 4));
    MI_HIP(dL.reserve(sizeof(SurfLayers))); MI_HIP(dkeys.reserve((size_t)CAND_CAP * 8)); MI_HIP(dcnt.reserve(64));
    // orientation disc + descriptor window weights (cv::getGaussianKernel rounding, oracle_surf.c gauss_taps)
    static OriTable h_tab; static float h_dw[PATCH * PATCH]; static bool tab_ready = false;
    static std::mutex tab_mu;
    {
        std::lock_guard<std::mutex> lk(tab_mu);
        if (!tab_ready) {
            auto taps = [](int n, double sigma, float* k) {
                double sum = 0.0, s2 = -0.5 / (sigma * sigma);
                for (int i = 0; i < n; i++) { const double x = (double)i - (double)(n - 1) * 0.5; k[i] = (float)std::exp(s2 * x * x); sum += (double)k[i]; }
                sum = 1.0 / sum;
                for (int i = 0; i < n; i++) k[i] = (float)((double)k[i] * sum);
            };
            float G[2 * ORI_R + 1], g20[PATCH];
            taps(2 * ORI_R + 1, 2.5, G); taps(PATCH, 3.3, g20);
            h_tab.n = 0;
            for (int i = -ORI_R; i <= ORI_R; i++)
                for (int j = -ORI_R; j <= ORI_R; j++)
                    if (i * i + j * j <= ORI_R * ORI_R) { h_tab.x[h_tab.n] = j; h_tab.y[h_tab.n] = i; h_tab.w[h_tab.n] = G[i + ORI_R] * G[j + ORI_R]; h_tab.n++; }
            for (int i = 0; i < PATCH; i++) for (int j = 0; j < PATCH; j++) h_dw[i * PATCH + j] = g20[i] * g20[j];
            tab_ready = true;
        }
    }
    MI_HIP(dtab.reserve(sizeof(OriTable) + sizeof(h_dw)));
    OriTable* d_tab = dtab.as<OriTable>();
    float* d_dw = reinterpret_cast<float*>(dtab.as<uint8_t>() + sizeof(OriTable));
    MI_HIP(hipMemcpyAsync(d_tab, &h_tab, sizeof(OriTable), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(d_dw, h_dw, sizeof(h_dw), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemcpyAsync(dL.p, &L, sizeof(L), hipMemcpyHostToDevice, st));
    MI_HIP(hipMemsetAsync(ddet.p, 0, plane * 4, st)); MI_HIP(hipMemsetAsync(dtr.p, 0, plane * 4, st));
    unsigned* d_count = dcnt.as<unsigned>(); int* d_nkeep = dcnt.as<int>() + 1; int* d_nout = dcnt.as<int>() + 2;
    MI_HIP(hipMemsetAsync(dcnt.p, 0, 64, st));
    {
        ProfScope ps(ctx, "surf_integral", (double)w * h * 11.0, st);
        hipLaunchKernelGGL(surf_gray_rows, dim3((h + 3) / 4), dim3(256), 0, st, d_bgr, ws, w, h, dgray.as<uint8_t>(), dS.as<uint32_t>());
        hipLaunchKernelGGL(surf_cols, dim3((sw + 255) / 256), dim3(256), 0, st, dS.as<uint32_t>(), sw, sh);
    }
    if (max_si > 0) {
        ProfScope ps(ctx, "surf_det", 0.0, st);
        // one launch per octave, sized for that octave (a common grid of the largest layer spent more time dispatching empty
        // workgroups than computing); octave 0 (step 1, boxes <= 27) from an LDS tile, the others straight from the integral image
        for (int o = 0; o < S_OCT; o++) {
            int si = 0, sj = 0, rows = 0, cols = 0;
            bool tile_ok = (o == 0);
            for (int l = 0; l < S_LAY + 2; l++) {
                const LayerDesc& q = L.l[o * (S_LAY + 2) + l];
                si = std::max(si, q.si); sj = std::max(sj, q.sj); rows = std::max(rows, q.rows); cols = std::max(cols, q.cols);
                tile_ok = tile_ok && q.step == 1 && q.size < DT_HALO;
            }
            if (si <= 0 || sj <= 0) continue;
            if (tile_ok)
                hipLaunchKernelGGL(surf_det_tile, dim3((sj + DT_W - 1) / DT_W, (si + DT_H - 1) / DT_H), dim3(256), 0, st, dS.as<uint32_t>(), sw, sh, dL.as<SurfLayers>(), ddet.as<float>(), dtr.as<float>());
            else
                hipLaunchKernelGGL(surf_det, dim3((sj + 63) / 64, (si + 3) / 4, S_LAY + 2), dim3(256), 0, st, dS.as<uint32_t>(), sw, dL.as<SurfLayers>(), ddet.as<float>(), dtr.as<float>(), o * (S_LAY + 2));
        }
        for (int o = 0; o < S_OCT; o++) {
            const LayerDesc& q = L.l[o * (S_LAY + 2) + 1];
            if (q.rows <= 0 || q.cols <= 0) continue;
            hipLaunchKernelGGL(surf_maxima, dim3((q.cols + 63) / 64, (q.rows + 3) / 4, S_LAY), dim3(256), 0, st, dL.as<SurfLayers>(), ddet.as<float>(), thr,
                               dkeys.as<unsigned long long>(), d_count, CAND_CAP, o * S_LAY);
        }
    }
    // the number of candidates decides the sort size: one small synchronous read (the extraction is synchronous anyway)
    unsigned cnt = 0;
    MI_HIP(hipMemcpyAsync(&cnt, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    if (cnt > CAND_CAP) { ctx->set_error("surf: more than 2^21 Hessian maxima above the threshold (raise hessian_threshold)"); return MI355_ERR_FAILED; }
    unsigned N = 2048;
    while (N < cnt) N <<= 1;
    {
        ProfScope ps(ctx, "surf_sort", (double)N * 8.0, st);
        hipLaunchKernelGGL(surf_pad_keys, dim3((N + 255) / 256), dim3(256), 0, st, dkeys.as<unsigned long long>(), d_count, CAND_CAP, N);
        // k = 2 .. 2048: fully inside the 2048-key blocks
        hipLaunchKernelGGL(surf_sort_local, dim3(N / 2048), dim3(1024), 0, st, dkeys.as<unsigned long long>(), N, 2u, 2048u, 1u);
        for (unsigned k = 4096; k <= N; k <<= 1) {
            for (unsigned j = k >> 1; j >= 2048; j >>= 1)
                hipLaunchKernelGGL(surf_sort_global, dim3((N / 2 + 255) / 256), dim3(256), 0, st, dkeys.as<unsigned long long>(), N, k, j);
            hipLaunchKernelGGL(surf_sort_local, dim3(N / 2048), dim3(1024), 0, st, dkeys.as<unsigned long long>(), N, k, k, 1024u);
        }
    }
    SurfFeatures& f = surf_state(ctx)->feats[img_id];
    f.w = w; f.h = h;
    const size_t keep_max = cnt < (unsigned)max_kp ? (cnt > 0 ? cnt : 1) : (size_t)max_kp;      // the image's feature storage: what this extraction can keep, not the limit
    MI_HIP(f.kp.reserve(sizeof(mi355_keypoint) * keep_max)); MI_HIP(f.desc.reserve(keep_max * 128 * 4)); MI_HIP(f.xy.reserve(sizeof(float2) * keep_max));
    // work areas and grids follow what this image really keeps (max_kp may be "all": 2^21), never less than the 32768 of rounds 1-5
    const int launch_kp = (int)(keep_max > 32768 ? keep_max : (size_t)(max_kp < 32768 ? max_kp : 32768));
    MI_HIP(dkps.reserve(sizeof(SurfKp) * (size_t)launch_kp)); MI_HIP(ddesc.reserve((size_t)launch_kp * 128 * 4));
    {
        ProfScope ps(ctx, "surf_describe", 0.0, st);
        hipLaunchKernelGGL(surf_finalize, dim3((launch_kp + 255) / 256), dim3(256), 0, st, dL.as<SurfLayers>(), ddet.as<float>(), dtr.as<float>(), dkeys.as<unsigned long long>(),
                           d_count, CAND_CAP, max_kp, dkps.as<SurfKp>(), d_nkeep);
        hipLaunchKernelGGL(surf_orient, dim3((launch_kp + 3) / 4), dim3(256), 0, st, dS.as<uint32_t>(), w, h, d_tab, dkps.as<SurfKp>(), d_nkeep);
        hipLaunchKernelGGL(surf_describe, dim3(launch_kp), dim3(256), 0, st, dgray.as<uint8_t>(), w, h, d_dw, dkps.as<SurfKp>(), d_nkeep, ddesc.as<float>());
        DevBuf& dpos = ctx->buf("surf_pos");
        MI_HIP(dpos.reserve(sizeof(int) * (size_t)launch_kp));
        hipLaunchKernelGGL(surf_compact, dim3(1), dim3(1024), 0, st, dkps.as<SurfKp>(), d_nkeep, f.kp.as<mi355_keypoint>(), f.xy.as<float2>(), dpos.as<int>(), d_nout);
        hipLaunchKernelGGL(surf_move_desc, dim3((launch_kp + 1) / 2), dim3(256), 0, st, ddesc.as<float>(), dpos.as<int>(), d_nkeep, f.desc.as<float>());
    }
    MI_HIP(hipGetLastError());
    int n = 0;
    MI_HIP(hipMemcpyAsync(&n, d_nout, sizeof(int), hipMemcpyDeviceToHost, st));
    MI_HIP(hipStreamSynchronize(st));
    f.n = n;
    if (n_kp) *n_kp = n;
    return MI355_OK;
}

// sort kernel details: surf_sort_local(k_lo, k_hi, j_start) covers, inside each 2048-key block, the comparator strides j <= 1024
extern "C" int mi355_surf_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int width_step, float hessian_threshold, int max_kp, int* n_kp) {
    LOCKED_PROLOGUE
    if (!d_bgr) return MI355_ERR_ARG;
    return surf_extract_dev(ctx, img_id, d_bgr, w, h, width_step, hessian_threshold, max_kp, n_kp);
}

// caller holds the ctx lock
static int surf_get_features_locked(mi355_ctx* ctx, int img_id, mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp) {
    auto& feats = surf_state(ctx)->feats;
    auto it = feats.find(img_id);
    if (it == feats.end()) { ctx->set_error("surf_get_features: unknown image id"); return MI355_ERR_ARG; }
    SurfFeatures& f = it->second;
    if (n_kp) *n_kp = f.n;
    const int n = f.n < max_kp ? f.n : max_kp;
    if (n <= 0) return MI355_OK;
    if (kp) MI_HIP(hipMemcpyAsync(kp, f.kp.p, sizeof(mi355_keypoint) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (desc128) MI_HIP(hipMemcpyAsync(desc128, f.desc.p, (size_t)n * 128 * 4, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

extern "C" int mi355_surf_get_features(mi355_ctx* ctx, int img_id, mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp) {
    LOCKED_PROLOGUE
    return surf_get_features_locked(ctx, img_id, kp, desc128, max_kp, n_kp);
}

extern "C" int mi355_surf_extract(mi355_ctx* ctx, int img_id, const uint8_t* bgr, int w, int h, int width_step, float hessian_threshold, int max_kp,
                                  mi355_keypoint* kp, float* desc128, int* n_kp) {
    // extraction and the copy back happen under ONE hold of the ctx lock: another thread re-extracting or dropping the same id on
    // the shared context cannot slip in between
    LOCKED_PROLOGUE
    int n = 0;
    if (!bgr || w < 16 || h < 16 || width_step < 3 * w) { ctx->set_error("surf_extract: bad image geometry"); return MI355_ERR_ARG; }
    DevBuf& dimg = ctx->buf("surf_host_img");
    MI_HIP(dimg.reserve((size_t)width_step * h + 16));
    MI_HIP(hipMemcpyAsync(dimg.p, bgr, (size_t)width_step * h, hipMemcpyHostToDevice, ctx->stream));
    int rc = surf_extract_dev(ctx, img_id, dimg.as<uint8_t>(), w, h, width_step, hessian_threshold, max_kp, &n);
    if (rc != MI355_OK) return rc;
    if (n_kp) *n_kp = n;
    if (kp || desc128) return surf_get_features_locked(ctx, img_id, kp, desc128, max_kp, nullptr);
    return MI355_OK;
}

extern "C" int mi355_surf_drop_features(mi355_ctx* ctx, int img_id) {
    LOCKED_PROLOGUE
    MI_HIP(hipStreamSynchronize(ctx->stream));
    auto& feats = surf_state(ctx)->feats;
    if (img_id < 0) { for (auto& kv : feats) kv.second.release(); feats.clear(); }
    else { auto it = feats.find(img_id); if (it != feats.end()) { it->second.release(); feats.erase(it); } }
    return MI355_OK;
}

// the ring schedule of the SURF variant: ext = min(15, n/2 - 1); j0 in (i, i + ext], wrapped modulo n (MosaicWithoutPos.cpp:5370-5377)
extern "C" int mi355_surf_pair_schedule(int n_images, int32_t* pairs_ij, int max_pairs, int* n_pairs) {
    if (n_images < 0 || !n_pairs) return MI355_ERR_ARG;
    int ext = n_images / 2 - 1; if (ext > 15) ext = 15;
    int cnt = 0;
    for (int i = 0; i < n_images; i++)
        for (int j0 = i + 1; j0 < n_images + ext; j0++) {
            if (j0 - i > ext) continue;
            const int j = j0 >= n_images ? j0 - n_images : j0;
            if (pairs_ij && cnt < max_pairs) { pairs_ij[2 * cnt] = i; pairs_ij[2 * cnt + 1] = j; }
            cnt++;
        }
    *n_pairs = cnt;
    return (pairs_ij && cnt > max_pairs) ? MI355_ERR_ARG : MI355_OK;
}

extern "C" int mi355_surf_match_pairs(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed, float match_dist, int max_features,
                                      int min_inliers, mi355_pair_result* out) {
    LOCKED_PROLOGUE
    if (n_pairs < 0 || (n_pairs > 0 && (!pairs_ij || !out))) return MI355_ERR_ARG;
    if (n_pairs == 0) return MI355_OK;
    if (max_features < 1 || max_features > MI355_MAX_SELECTED) { ctx->set_error("surf_match_pairs: max_features must be in [1, 400]"); return MI355_ERR_ARG; }
    auto& feats = surf_state(ctx)->feats;
    const int BATCH = 512;                                      // bounds the nn workspaces (512 pairs x the batch's largest query count x 8 B)
    if (!(match_dist == match_dist) || match_dist > 100.0f) { ctx->set_error("surf_match_pairs: match_dist must be a number <= 100 (descriptors are unit vectors)"); return MI355_ERR_ARG; }      // the threshold walk lowers it in steps of 0.05 until few enough matches are left
    DevBuf& dpd = ctx->buf("surf_pairs"); DevBuf& didx = ctx->buf("surf_nn_idx"); DevBuf& ddist = ctx->buf("surf_nn_dist");
    DevBuf& ds1 = ctx->buf("sel1"); DevBuf& ds2 = ctx->buf("sel2"); DevBuf& dns = ctx->buf("nsel"); DevBuf& dres = ctx->buf("pair_results");
    for (int b0 = 0; b0 < n_pairs; b0 += BATCH) {
        const int nb = n_pairs - b0 < BATCH ? n_pairs - b0 : BATCH;
        std::vector<SPair> pd(nb);
        int max_ni = 1;
        for (int p = 0; p < nb; p++) {
            const int i = pairs_ij[2 * (b0 + p)], j = pairs_ij[2 * (b0 + p) + 1];
            auto fi = feats.find(i), fj = feats.find(j);
            if (fi == feats.end() || fj == feats.end()) { ctx->set_error("surf_match_pairs: no resident SURF features for image " + std::to_string(fi == feats.end() ? i : j)); return MI355_ERR_ARG; }
            const SurfFeatures &a = fi->second, &b = fj->second;
            pd[p] = SPair{a.desc.as<float>(), a.xy.as<float2>(), a.n, b.desc.as<float>(), b.xy.as<float2>(), b.n, i, j};
            if (a.n > max_ni) max_ni = a.n;
        }
        MI_HIP(dpd.reserve(sizeof(SPair) * nb)); const int nn_stride = (max_ni + 63) & ~63;
        MI_HIP(didx.reserve((size_t)nb * nn_stride * 4)); MI_HIP(ddist.reserve((size_t)nb * nn_stride * 4));
        MI_HIP(ds1.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)nb)); MI_HIP(ds2.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)nb));
        MI_HIP(dns.reserve(sizeof(int) * nb)); MI_HIP(dres.reserve(sizeof(mi355_pair_result) * (size_t)nb));
        MI_HIP(hipMemcpyAsync(dpd.p, pd.data(), sizeof(SPair) * nb, hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
        {
            ProfScope ps(ctx, "surf_match", 0.0);
            hipLaunchKernelGGL(surf_bf, dim3((max_ni + 255) / 256, nb), dim3(256), 0, ctx->stream, dpd.as<SPair>(), didx.as<int>(), ddist.as<float>(), nn_stride);
            hipLaunchKernelGGL(surf_select, dim3(nb), dim3(1024), 0, ctx->stream, dpd.as<SPair>(), didx.as<int>(), ddist.as<float>(), nn_stride, match_dist, max_features,
                               ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>());
        }
        MI_HIP(hipGetLastError());
        int rc = mi_ransac_batch(ctx, ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>(), nullptr, nb, MI355_MAX_SELECTED, ransac_dist, ctx->p.sample_times, seed,
                                 dres.as<mi355_pair_result>(), min_inliers);
        if (rc != MI355_OK) return rc;
        hipLaunchKernelGGL(surf_finalize_pairs, dim3((nb + 255) / 256), dim3(256), 0, ctx->stream, dpd.as<SPair>(), dns.as<int>(), nb, min_inliers, dres.as<mi355_pair_result>());
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(out + b0, dres.p, sizeof(mi355_pair_result) * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MI355_OK;
}
