// csrc/blend.hip -- multiband blend of the warped chips, replacing
//   detail::MultiBandBlender blender(false, band); prepare / feed per chip / blend; convertTo(CV_8U)
//   (MosaicImage.cpp:2296-2299, 2451-2486).
// The arithmetic is OpenCV 2.4.0's (binaries only; the reference commits no blended output).  The definition implemented here --
// 16-bit Laplacian pyramids, float weight pyramids, [1 4 6 4 1] REDUCE / EXPAND in integer arithmetic, every rounding and border --
// is the one stated at the top of oracle/oracle_blend.c, which also lists what was checked against the reference's DLLs and the one
// known divergence (the binary's reassociated float REDUCE); the parity test compares the output bytes with that oracle.
// All kernels are streaming stencils over at most a few hundred MB.  Level 0 of a chip's pyramids is never stored (round 4): it IS the
// chip (u8 -> i16, extended by reflection) and mask / 255 (extended by zeros), so the first REDUCE and the level-0 Laplacian read the
// chip and the mask themselves (4 bytes per pixel instead of 10 written and 20 read).  The REDUCE chains of up to 32 chips run as five
// batched launches (blockIdx.z = chip; they are independent of one another); the accumulation into the canvas pyramids stays one chip
// after the other, in chip order, because the float weight sums make the order per pixel part of the result: per chip 5 x Laplacian +
// accumulate (a 2 x 2 fine block per thread, never stored) and the top level's accumulate.
#include "common.h"
#include <cmath>

namespace {

__device__ __forceinline__ int reflect101d(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }
__device__ __forceinline__ int reflectd(int p, int n) { while (p < 0 || p >= n) { if (p < 0) p = -p - 1; else p = 2 * n - 1 - p; } return p; }
__device__ __forceinline__ short sat16d(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// Pointers read from a structure in memory are generic to the compiler (flat loads, and no unaligned vector loads: the 24-byte reads
// below came out as 24 byte loads); in the global address space an unaligned run of bytes is one or two vector loads.
#define GLOBAL_U8(p) ((const __attribute__((address_space(1))) uint8_t*)(p))
__device__ __forceinline__ void load_run(const void* p, unsigned* out, int nbytes_const8) {      // nbytes: 8, 20 or 24
    const __attribute__((address_space(1))) uint8_t* g = GLOBAL_U8(p);
    if (nbytes_const8 == 8) __builtin_memcpy(out, (const void __attribute__((address_space(1)))*)g, 8);
    else if (nbytes_const8 == 20) __builtin_memcpy(out, (const void __attribute__((address_space(1)))*)g, 20);
    else __builtin_memcpy(out, (const void __attribute__((address_space(1)))*)g, 24);
}

// level 0 of one chip's region: chip extended by reflection (edge pixel included), weight = mask / 255 extended by zeros
__global__ __launch_bounds__(256) void blend_prep_kernel(const uint8_t* chip, int cws, const uint8_t* mask, int mws, int cw, int ch,
                                                         int left, int top, int rw, int rh, short* g0, float* w0) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= rw) return;
    const int sx = reflectd(x - left, cw), sy = reflectd(y - top, ch);
    const uint8_t* p = chip + (size_t)sy * cws + 3 * sx;
    short* g = g0 + ((size_t)y * rw + x) * 3;
    g[0] = (short)p[0]; g[1] = (short)p[1]; g[2] = (short)p[2];
    float w = 0.0f;
    if (y - top >= 0 && y - top < ch && x - left >= 0 && x - left < cw) w = (float)mask[(size_t)(y - top) * mws + (x - left)] * (float)(1.0 / 255.0);
    w0[(size_t)y * rw + x] = w;
}

// one chip of a batch: where its pixels are, how its region lies on the chip, where its pyramid levels >= 1 live
constexpr int MAX_BANDS = 16;
struct Win { int x0, y0, x1, y1; };   // inclusive
struct ChipP {
    const uint8_t* chip; const uint8_t* mask;
    int cw, ch, cws, mws;             // chip size, row pitches of chip (3 B / pixel) and mask
    int left, top, rw, rh;            // chip origin inside its region, region size (multiples of 2^bands)
    size_t tmp;                       // pixel offset of this chip's level 1 inside the batch's pyramid buffers
    int tlx, tly;                     // the region's origin on the canvas
    // Active windows (round 4).  With FindMasksByDistMap's masks a chip's weights are non-zero only over the cell of the mosaic it owns (+ the
    // reach of the REDUCE filter per level), and a pixel of weight +0 adds nothing to the canvas: only the part of the pyramids that the
    // non-zero weights can see is ever formed.  cwin[l], l = 1 .. bands: the pixels of level l (Gaussian and weight) that are computed --
    // everything outside is never written and never read; twin[l], l = 0 .. bands: the threads of the accumulation of level l (one per
    // 2 x 2 block of level l below the top level, one per pixel at the top level).  See chip_windows() for the derivation.
    Win cwin[MAX_BANDS + 1];
    Win twin[MAX_BANDS + 1];
};
// pixel offset of level l >= 1 behind tmp
__device__ __forceinline__ size_t level_off(int rw, int rh, int l) { size_t o = 0; for (int m = 1; m < l; m++) o += (size_t)(rw >> m) * (rh >> m); return o; }

// REDUCE: i16 x 3 in integers (so the 5x5 product form equals the oracle's rows-then-columns form: rows [1 4 6 4 1] . pixels, then
// columns, (sum + 128) >> 8) and f32 weights in the oracle's order (6 c + 4 (l + r) + ll + rr per row, the same over the rows, / 256).
// REDUCE of both pyramids of a chip in one launch, two horizontally adjacent outputs per thread: their 5-tap windows share three of the
// seven source columns, and away from the left / right border those seven pixels are 42 contiguous, 4-byte aligned bytes (11 32-bit
// loads per row instead of 30 16-bit ones).  The sums are the ones of pyr_down16_kernel / pyr_down_f_kernel, term for term.
__device__ __forceinline__ void pyr_down_pair_body(const short* src, const float* srcw, int w, int h, short* dst, float* dstw, const Win win) {
    const int dw = w >> 1, x0 = (win.x0 & ~1) + (blockIdx.x * 256 + threadIdx.x) * 2, y = win.y0 + blockIdx.y;
    if (x0 >= dw || y >= (h >> 1) || x0 > win.x1 || y > win.y1) return;
    const bool two = x0 + 1 < dw;
    const int wt[5] = {1, 4, 6, 4, 1};
    const bool interior = 2 * x0 - 2 >= 0 && 2 * x0 + 4 < w;
    int xs[7];
#pragma unroll
    for (int j = 0; j < 7; j++) xs[j] = reflect101d(2 * x0 - 2 + j, w);
    int acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
    float fr[2][5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int sy = reflect101d(2 * y - 2 + k, h);
        const short* s = src + (size_t)sy * w * 3;
        short px[7][3];
        if (interior) {
            const unsigned* p = reinterpret_cast<const unsigned*>(s + 3 * (2 * x0 - 2));      // 12 (x0 - 1) bytes into a row of 6 w bytes, w even
            unsigned u[11];
#pragma unroll
            for (int q = 0; q < 11; q++) u[q] = p[q];
#pragma unroll
            for (int j = 0; j < 7; j++)
#pragma unroll
                for (int c = 0; c < 3; c++) { const int e = 3 * j + c; px[j][c] = (short)((e & 1) ? (u[e >> 1] >> 16) : (u[e >> 1] & 0xffff)); }
        } else {
#pragma unroll
            for (int j = 0; j < 7; j++) { px[j][0] = s[3 * xs[j]]; px[j][1] = s[3 * xs[j] + 1]; px[j][2] = s[3 * xs[j] + 2]; }
        }
#pragma unroll
        for (int o = 0; o < 2; o++) {
            int r[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < 5; j++) { r[0] += wt[j] * px[2 * o + j][0]; r[1] += wt[j] * px[2 * o + j][1]; r[2] += wt[j] * px[2 * o + j][2]; }
            acc[o][0] += wt[k] * r[0]; acc[o][1] += wt[k] * r[1]; acc[o][2] += wt[k] * r[2];
        }
        const float* sw = srcw + (size_t)sy * w;
        float f[7];
#pragma unroll
        for (int j = 0; j < 7; j++) f[j] = sw[xs[j]];
#pragma unroll
        for (int o = 0; o < 2; o++) fr[o][k] = f[2 * o + 2] * 6.0f + (f[2 * o + 1] + f[2 * o + 3]) * 4.0f + f[2 * o] + f[2 * o + 4];
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
        if (o == 1 && !two) break;
        short* d = dst + ((size_t)y * dw + x0 + o) * 3;
        d[0] = sat16d((acc[o][0] + 128) >> 8); d[1] = sat16d((acc[o][1] + 128) >> 8); d[2] = sat16d((acc[o][2] + 128) >> 8);
        const float v = fr[o][2] * 6.0f + (fr[o][1] + fr[o][3]) * 4.0f + fr[o][0] + fr[o][4];
        dstw[(size_t)y * dw + x0 + o] = v * (1.0f / 256.0f);
    }
}
__global__ __launch_bounds__(256) void pyr_down_pair_kernel(const short* src, const float* srcw, int w, int h, short* dst, float* dstw) {
    pyr_down_pair_body(src, srcw, w, h, dst, dstw, Win{0, 0, (w >> 1) - 1, (h >> 1) - 1});
}
// level l -> l + 1 (l >= 1) of every chip of a batch; the grid covers the largest chip
__global__ __launch_bounds__(256) void pyr_down_pair_batch_kernel(const ChipP* cp, int l, short* g, float* wp) {
    const ChipP& c = cp[blockIdx.z];
    const size_t a = c.tmp + level_off(c.rw, c.rh, l), b = c.tmp + level_off(c.rw, c.rh, l + 1);
    const Win win = l + 1 <= MAX_BANDS ? c.cwin[l + 1] : Win{0, 0, (c.rw >> (l + 1)) - 1, (c.rh >> (l + 1)) - 1};
    pyr_down_pair_body(g + a * 3, wp + a, c.rw >> l, c.rh >> l, g + b * 3, wp + b, win);
}

// Level 0 -> 1 straight from the chip and its mask: the level-0 value at region pixel (x, y) is (short)chip[reflect(x - left), reflect(y - top)]
// (the chip extended by BORDER_REFLECT, edge pixel included), the weight mask / 255 inside the chip and 0 outside -- what blend_prep_kernel
// used to store.  Same sums as pyr_down_pair_body, term for term.  Away from the borders the seven pixels of a row are 21 contiguous
// bytes: six unaligned 32-bit loads (three bytes of slack inside the row), the seven mask bytes two.
__global__ __launch_bounds__(256) void pyr_down0_batch_kernel(const ChipP* cp, short* g, float* wp) {
    const ChipP& c = cp[blockIdx.z];
    const int w = c.rw, h = c.rh;
    const Win win = c.cwin[1];
    const int dw = w >> 1, x0 = (win.x0 & ~1) + (blockIdx.x * 256 + threadIdx.x) * 2, y = win.y0 + blockIdx.y;
    if (x0 >= dw || y >= (h >> 1) || x0 > win.x1 || y > win.y1) return;
    short* dst = g + c.tmp * 3;
    float* dstw = wp + c.tmp;
    const bool two = x0 + 1 < dw;
    const int wt[5] = {1, 4, 6, 4, 1};
    const int cx0 = 2 * x0 - 2 - c.left;                                   // chip column of the first of the seven
    const bool interior = 2 * x0 - 2 >= 0 && 2 * x0 + 4 < w && cx0 >= 0 && cx0 + 8 < c.cw;
    int xs[7]; bool xin[7];
    if (!interior) {
#pragma unroll
        for (int j = 0; j < 7; j++) { const int xr = reflect101d(2 * x0 - 2 + j, w) - c.left; xin[j] = xr >= 0 && xr < c.cw; xs[j] = reflectd(xr, c.cw); }
    }
    int acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
    float fr[2][5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int yr = reflect101d(2 * y - 2 + k, h) - c.top;
        const bool yin = yr >= 0 && yr < c.ch;
        const int sy = reflectd(yr, c.ch);
        const __attribute__((address_space(1))) uint8_t* s = GLOBAL_U8(c.chip) + (size_t)sy * c.cws;
        const __attribute__((address_space(1))) uint8_t* m = GLOBAL_U8(c.mask) + (size_t)sy * c.mws;
        short px[7][3];
        float f[7];
        if (interior) {
            // the horizontal [1 4 6 4 1] of both outputs and the three channels straight on the packed bytes: pixel j, channel ch is byte
            // 3 j + ch of the run, so a dword meets a constant vector of tap weights (zeros on the other channels' bytes) in one
            // v_dot4_u32_u8 -- 27 of them per row instead of 21 byte extractions and 30 multiply-adds; the same integers
            unsigned u[6];
            load_run((const void*)(s + 3 * cx0), u, 24);
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    unsigned r = 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        unsigned wq = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int e = 4 * q + b, jj = e / 3 - 2 * o;
                            if (e % 3 == ch && jj >= 0 && jj < 5) wq |= (unsigned)wt[jj] << (8 * b);
                        }
                        if (wq) r = __builtin_amdgcn_udot4(u[q], wq, r, false);
                    }
                    acc[o][ch] += wt[k] * (int)r;
                }
            unsigned mv[2];
            load_run((const void*)(m + cx0), mv, 8);
#pragma unroll
            for (int j = 0; j < 7; j++) f[j] = yin ? (float)((mv[j >> 2] >> (8 * (j & 3))) & 0xffu) * (float)(1.0 / 255.0) : 0.0f;
        } else {
#pragma unroll
            for (int j = 0; j < 7; j++) {
                px[j][0] = (short)s[3 * xs[j]]; px[j][1] = (short)s[3 * xs[j] + 1]; px[j][2] = (short)s[3 * xs[j] + 2];
                f[j] = (yin && xin[j]) ? (float)m[xs[j]] * (float)(1.0 / 255.0) : 0.0f;
            }
#pragma unroll
            for (int o = 0; o < 2; o++) {
                int r[3] = {0, 0, 0};
#pragma unroll
                for (int j = 0; j < 5; j++) { r[0] += wt[j] * px[2 * o + j][0]; r[1] += wt[j] * px[2 * o + j][1]; r[2] += wt[j] * px[2 * o + j][2]; }
                acc[o][0] += wt[k] * r[0]; acc[o][1] += wt[k] * r[1]; acc[o][2] += wt[k] * r[2];
            }
        }
#pragma unroll
        for (int o = 0; o < 2; o++) fr[o][k] = f[2 * o + 2] * 6.0f + (f[2 * o + 1] + f[2 * o + 3]) * 4.0f + f[2 * o] + f[2 * o + 4];
    }
#pragma unroll
    for (int o = 0; o < 2; o++) {
        if (o == 1 && !two) break;
        short* d = dst + ((size_t)y * dw + x0 + o) * 3;
        d[0] = sat16d((acc[o][0] + 128) >> 8); d[1] = sat16d((acc[o][1] + 128) >> 8); d[2] = sat16d((acc[o][2] + 128) >> 8);
        const float v = fr[o][2] * 6.0f + (fr[o][1] + fr[o][3]) * 4.0f + fr[o][0] + fr[o][4];
        dstw[(size_t)y * dw + x0 + o] = v * (1.0f / 256.0f);
    }
}

// horizontal EXPAND value (before the vertical combination) at fine column X of coarse row s (3 channels, channel c)
__device__ __forceinline__ int up_h(const short* s, int w, int X, int c) {
    const int x = X >> 1;
    if (w == 1) return s[c] * 8;
    if (!(X & 1)) {
        if (x == 0) return s[c] * 6 + s[3 + c] * 2;
        if (x == w - 1) return s[3 * (w - 2) + c] + s[3 * (w - 1) + c] * 7;
        return s[3 * (x - 1) + c] + s[3 * x + c] * 6 + s[3 * (x + 1) + c];
    }
    if (x == w - 1) return s[3 * (w - 1) + c] * 8;
    return (s[3 * x + c] + s[3 * (x + 1) + c]) * 4;
}

// up_h at the fine columns 2x and 2x + 1 for the three channels of one coarse row; away from the left / right border the three coarse
// pixels are 18 contiguous bytes (one 20-byte run instead of nine 2-byte loads)
__device__ __forceinline__ void up_h_pair(const short* row, int w, int x, int* he, int* ho) {
    if (x >= 1 && x + 1 < w) {
        unsigned u[5];
        load_run(row + 3 * (x - 1), u, 20);
        short v[9];
#pragma unroll
        for (int e = 0; e < 9; e++) v[e] = (short)((e & 1) ? (u[e >> 1] >> 16) : (u[e >> 1] & 0xffffu));
#pragma unroll
        for (int c = 0; c < 3; c++) { he[c] = v[c] + v[3 + c] * 6 + v[6 + c]; ho[c] = (v[3 + c] + v[6 + c]) * 4; }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) { he[c] = up_h(row, w, 2 * x, c); ho[c] = up_h(row, w, 2 * x + 1, c); }
    }
}

// fine = sat16(fine - EXPAND(coarse)) (SUB) or sat16(EXPAND(coarse) + fine); coarse is w x h, fine 2w x 2h
template <bool SUB>
__global__ __launch_bounds__(256) void pyr_up16_combine_kernel(const short* coarse, int w, int h, short* fine, int Y0 = 0) {
    const int X = blockIdx.x * 256 + threadIdx.x, Y = Y0 + blockIdx.y;      // fine rows Y0 .. Y0 + gridDim.y - 1
    if (X >= 2 * w) return;
    const int y = Y >> 1;
    const int ym = (y == 0) ? (h > 1 ? 1 : 0) : y - 1, yp = (y == h - 1) ? h - 1 : y + 1;
    const short* rm = coarse + (size_t)ym * w * 3;
    const short* r0 = coarse + (size_t)y * w * 3;
    const short* rp = coarse + (size_t)yp * w * 3;
    short* f = fine + ((size_t)Y * 2 * w + X) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int v;
        if (!(Y & 1)) v = up_h(rm, w, X, c) + up_h(r0, w, X, c) * 6 + up_h(rp, w, X, c);
        else v = (up_h(r0, w, X, c) + up_h(rp, w, X, c)) * 4;
        const int up = sat16d((v + 32) >> 6);
        f[c] = SUB ? sat16d((int)f[c] - up) : sat16d(up + (int)f[c]);
    }
}

// Laplacian level of a chip and its accumulation in one pass: lap = sat16(fine - EXPAND(coarse)) is formed in registers and added to the
// canvas, never stored (the separate in-place pyr_up16_combine<true> + blend_accumulate pair moved 12 more bytes per pixel and was a
// third of the blend's kernel time).  Both levels stay Gaussian, so the levels can be taken in any order.
// One thread per COARSE pixel = a 2 x 2 block of fine pixels: the 3 x 3 coarse neighbourhood is read once for the four of them (a thread
// per fine pixel issued 27 two-byte loads each and ran at a quarter of the bandwidth the bytes need).
__device__ __forceinline__ void blend_lap_accumulate_body(const short* coarse, int w, int h, const short* fine, const float* wgt, int ox, int oy,
                                                          short* dl, float* dw, int DW, int x, int y) {
    if (x >= w) return;
    const int FW = 2 * w;
    {
        // A pixel whose weight is +0 leaves the canvas as it is: (short)((float)lap * 0.0f) = 0 and sum + 0.0f = sum (the weight sums start
        // at +0 and only grow).  With FindMasksByDistMap's masks a chip's weights vanish outside its own cell of the mosaic (+ the reach of
        // the REDUCE filter at this level) -- 98 % of a chip's pixels when 2000 chips share a 20000^2 canvas -- so the block is judged by its
        // four weights before anything else is read.
        const float2 wa = *reinterpret_cast<const float2*>(wgt + (size_t)(2 * y) * FW + 2 * x);
        const float2 wb = *reinterpret_cast<const float2*>(wgt + (size_t)(2 * y + 1) * FW + 2 * x);
        if (wa.x == 0.0f && wa.y == 0.0f && wb.x == 0.0f && wb.y == 0.0f && !__builtin_signbit(wa.x + wa.y + wb.x + wb.y)) return;
    }
    const int ym = (y == 0) ? (h > 1 ? 1 : 0) : y - 1, yp = (y == h - 1) ? h - 1 : y + 1;
    const short* rows[3] = {coarse + (size_t)ym * w * 3, coarse + (size_t)y * w * 3, coarse + (size_t)yp * w * 3};
    int he[3][3], ho[3][3];                               // horizontal EXPAND values at fine columns 2x (even) and 2x + 1 (odd), per row and channel
#pragma unroll
    for (int r = 0; r < 3; r++) up_h_pair(rows[r], w, x, he[r], ho[r]);
    // the two fine pixels of a row are 12 contiguous bytes (4-byte aligned: the fine column 2x, the region offset ox and the row pitches are
    // even): three 32-bit loads / stores instead of six 16-bit ones, the weights as one 64-bit access
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
        const int Y = 2 * y + dy;
        const size_t fi = (size_t)Y * FW + 2 * x;
        const size_t di = (size_t)(oy + Y) * DW + (ox + 2 * x);
        const float2 wv2 = *reinterpret_cast<const float2*>(wgt + fi);
        const unsigned* fp = reinterpret_cast<const unsigned*>(fine + fi * 3);
        unsigned* dp = reinterpret_cast<unsigned*>(dl + di * 3);
        const unsigned f0 = fp[0], f1 = fp[1], f2 = fp[2];
        unsigned d0 = dp[0], d1 = dp[1], d2 = dp[2];
        const short fv[6] = {(short)(f0 & 0xffff), (short)(f0 >> 16), (short)(f1 & 0xffff), (short)(f1 >> 16), (short)(f2 & 0xffff), (short)(f2 >> 16)};
        short dv[6] = {(short)(d0 & 0xffff), (short)(d0 >> 16), (short)(d1 & 0xffff), (short)(d1 >> 16), (short)(d2 & 0xffff), (short)(d2 >> 16)};
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const float wv = dx ? wv2.y : wv2.x;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int hm = dx ? ho[0][c] : he[0][c], h0 = dx ? ho[1][c] : he[1][c], hp = dx ? ho[2][c] : he[2][c];
                const int v = dy ? (h0 + hp) * 4 : hm + h0 * 6 + hp;
                const int up = sat16d((v + 32) >> 6);
                const short lap = sat16d((int)fv[3 * dx + c] - up);
                dv[3 * dx + c] = (short)(dv[3 * dx + c] + (short)((float)lap * wv));
            }
        }
        dp[0] = (unsigned)(unsigned short)dv[0] | ((unsigned)(unsigned short)dv[1] << 16);
        dp[1] = (unsigned)(unsigned short)dv[2] | ((unsigned)(unsigned short)dv[3] << 16);
        dp[2] = (unsigned)(unsigned short)dv[4] | ((unsigned)(unsigned short)dv[5] << 16);
        float2* wp2 = reinterpret_cast<float2*>(dw + di);
        float2 a2 = *wp2; a2.x += wv2.x; a2.y += wv2.y; *wp2 = a2;
    }
}

__global__ __launch_bounds__(256) void blend_lap_accumulate_kernel(const short* coarse, int w, int h, const short* fine, const float* wgt, int ox, int oy,
                                                                   short* dl, float* dw, int DW) {
    blend_lap_accumulate_body(coarse, w, h, fine, wgt, ox, oy, dl, dw, DW, blockIdx.x * 256 + threadIdx.x, blockIdx.y);
}

// Levels 1 .. bands of one chip in ONE launch (blockIdx.y walks the rows of all of them): the Laplacian levels 1 .. bands - 1 and the top
// (Gaussian) level.  Launched one by one the small levels cost ~8 us each whatever their size: four of the six launches per chip.
struct LapLevels {
    int n;                                   // entries: n - 1 Laplacian levels, then the top level
    int row0[MAX_BANDS + 1];                 // first grid row of entry i (row0[n] = grid rows)
    int w[MAX_BANDS], h[MAX_BANDS], ox[MAX_BANDS], oy[MAX_BANDS], DW[MAX_BANDS];
    int x0[MAX_BANDS], y0[MAX_BANDS], x1[MAX_BANDS];     // first thread column / row and last thread column of the entry's active window
    size_t fine[MAX_BANDS], coarse[MAX_BANDS];      // pixel offsets into the chip pyramid (g / wp)
    long long dst[MAX_BANDS];                       // ... and into the canvas pyramids (dl / dw): may be negative for a stripe (the level's rows above the stripe are not stored)
};
__global__ __launch_bounds__(256) void blend_lap_levels_kernel(LapLevels L, const short* g, const float* wp, short* dl, float* dw) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < L.n && (int)blockIdx.y >= L.row0[i + 1]) i++;
    const int y = L.y0[i] + (blockIdx.y - L.row0[i]), x = L.x0[i] + blockIdx.x * 256 + threadIdx.x;      // the level's active window (ChipP::twin)
    if (i + 1 < L.n) {
        if (x > L.x1[i]) return;
        blend_lap_accumulate_body(g + L.coarse[i] * 3, L.w[i], L.h[i], g + L.fine[i] * 3, wp + L.fine[i], L.ox[i], L.oy[i], dl + L.dst[i] * 3, dw + L.dst[i], L.DW[i], x, y);
    } else {
        // top level: canvas Laplacian += (short)(Gaussian * weight), canvas weight += weight (blend_accumulate_kernel)
        const int lw = L.w[i];
        if (x >= lw || x > L.x1[i]) return;
        const float wv = wp[L.fine[i] + (size_t)y * lw + x];
        if (wv == 0.0f && !__builtin_signbit(wv)) return;      // adds nothing (see blend_lap_accumulate_body)
        const long long di = L.dst[i] + (long long)(L.oy[i] + y) * L.DW[i] + (L.ox[i] + x);
        const short* s = g + (L.fine[i] + (size_t)y * lw + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) dl[di * 3 + c] = (short)(dl[di * 3 + c] + (short)((float)s[c] * wv));
        dw[di] += wv;
    }
}

// The same for level 0, whose Gaussian level is the chip itself (see pyr_down0_batch_kernel): fine values from the chip extended by
// reflection, weights mask / 255 inside the chip and 0 outside.  Inside the chip the two fine pixels of a row are six contiguous bytes.
// A row's two pixels are written as three 32-bit words only when BOTH have a weight: then both belong to this chip's cell and no other chip
// has a weight there.  Otherwise each pixel with a weight is updated on its own (16-bit accesses) and a pixel without one is not touched --
// so with masks that partition the canvas (FindMasksByDistMap) the chips write disjoint bytes at level 0, in any order: the batch form
// below runs the level-0 accumulation of up to 32 chips as one launch (per chip it was ~10 us of stream time whatever the window's size).
__device__ __forceinline__ void blend_lap0_body(const ChipP& c, const short* coarse, int ox, int oy, short* dl, float* dw, int DW) {
    const int w = c.rw >> 1, h = c.rh >> 1;
    const int x = c.twin[0].x0 + blockIdx.x * 256 + threadIdx.x, y = c.twin[0].y0 + blockIdx.y;
    if (x >= w || x > c.twin[0].x1 || y > c.twin[0].y1) return;      // (the rows matter: a stripe's canvas pyramids hold the stripe's rows only)
    const int cx = 2 * x - c.left;                            // chip column of the even fine pixel
    const bool xfast = cx >= 0 && cx + 3 < c.cw;              // both columns inside the chip and the 8-byte read inside the row
    {
        // the block's four mask bytes first: all zero (or outside the chip) = four weights of +0, which leave the canvas as it is
        // (blend_lap_accumulate_body); only the chip's own cell of the mosaic goes on
        unsigned any = 0;
#pragma unroll
        for (int dy = 0; dy < 2; dy++) {
            const int cy = 2 * y + dy - c.top;
            if (cy < 0 || cy >= c.ch) continue;
            const uint8_t* mrow = c.mask + (size_t)cy * c.mws;
            if (cx >= 0 && cx < c.cw) any |= mrow[cx];
            if (cx + 1 >= 0 && cx + 1 < c.cw) any |= mrow[cx + 1];
        }
        if (!any) return;
    }
    const int ym = (y == 0) ? (h > 1 ? 1 : 0) : y - 1, yp = (y == h - 1) ? h - 1 : y + 1;
    const short* rows[3] = {coarse + (size_t)ym * w * 3, coarse + (size_t)y * w * 3, coarse + (size_t)yp * w * 3};
    int he[3][3], ho[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++) up_h_pair(rows[r], w, x, he[r], ho[r]);
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
        const int Y = 2 * y + dy;
        const int cy = Y - c.top;
        const bool yin = cy >= 0 && cy < c.ch;
        const int sy = reflectd(cy, c.ch);
        const uint8_t* srow = c.chip + (size_t)sy * c.cws;
        const uint8_t* mrow = c.mask + (size_t)sy * c.mws;
        short fv[6]; float wv2[2];
        if (xfast) {
            unsigned u[2];
            load_run(srow + 3 * cx, u, 8);
#pragma unroll
            for (int e = 0; e < 6; e++) fv[e] = (short)((u[e >> 2] >> (8 * (e & 3))) & 0xffu);
            wv2[0] = yin ? (float)mrow[cx] * (float)(1.0 / 255.0) : 0.0f;
            wv2[1] = yin ? (float)mrow[cx + 1] * (float)(1.0 / 255.0) : 0.0f;
        } else {
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int xr = cx + dx, sx = reflectd(xr, c.cw);
                fv[3 * dx] = (short)srow[3 * sx]; fv[3 * dx + 1] = (short)srow[3 * sx + 1]; fv[3 * dx + 2] = (short)srow[3 * sx + 2];
                wv2[dx] = (yin && xr >= 0 && xr < c.cw) ? (float)mrow[sx] * (float)(1.0 / 255.0) : 0.0f;
            }
        }
        const size_t di = (size_t)(oy + Y) * DW + (ox + 2 * x);
        short add[6];
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const float wv = wv2[dx];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const int hm = dx ? ho[0][ch] : he[0][ch], h0 = dx ? ho[1][ch] : he[1][ch], hp = dx ? ho[2][ch] : he[2][ch];
                const int v = dy ? (h0 + hp) * 4 : hm + h0 * 6 + hp;
                const int up = sat16d((v + 32) >> 6);
                const short lap = sat16d((int)fv[3 * dx + ch] - up);
                add[3 * dx + ch] = (short)((float)lap * wv);
            }
        }
        if (wv2[0] != 0.0f && wv2[1] != 0.0f) {
            unsigned* dp = reinterpret_cast<unsigned*>(dl + di * 3);
            const unsigned d0 = dp[0], d1 = dp[1], d2 = dp[2];
            short dv[6] = {(short)(d0 & 0xffff), (short)(d0 >> 16), (short)(d1 & 0xffff), (short)(d1 >> 16), (short)(d2 & 0xffff), (short)(d2 >> 16)};
#pragma unroll
            for (int e = 0; e < 6; e++) dv[e] = (short)(dv[e] + add[e]);
            dp[0] = (unsigned)(unsigned short)dv[0] | ((unsigned)(unsigned short)dv[1] << 16);
            dp[1] = (unsigned)(unsigned short)dv[2] | ((unsigned)(unsigned short)dv[3] << 16);
            dp[2] = (unsigned)(unsigned short)dv[4] | ((unsigned)(unsigned short)dv[5] << 16);
            float2* wp2 = reinterpret_cast<float2*>(dw + di);
            float2 a2 = *wp2; a2.x += wv2[0]; a2.y += wv2[1]; *wp2 = a2;
        } else {
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                if (wv2[dx] == 0.0f) continue;                        // adds nothing (blend_lap_accumulate_body)
                short* dp = dl + (di + dx) * 3;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) dp[ch] = (short)(dp[ch] + add[3 * dx + ch]);
                dw[di + dx] += wv2[dx];
            }
        }
    }
}
__global__ __launch_bounds__(256) void blend_lap0_accumulate_kernel(ChipP c, const short* coarse, int ox, int oy, short* dl, float* dw, int DW) {
    blend_lap0_body(c, coarse, ox, oy, dl, dw, DW);
}
// level 0 of every chip of a batch (blockIdx.z = chip; the grid covers the largest window): only for masks that partition the canvas
__global__ __launch_bounds__(256) void blend_lap0_accumulate_batch_kernel(const ChipP* cp, const short* g, short* dl, float* dw, int DW) {
    const ChipP& c = cp[blockIdx.z];
    blend_lap0_body(c, g + c.tmp * 3, c.tlx, c.tly, dl, dw, DW);
}

// canvas Laplacian += (short)(chip Laplacian * weight), canvas weight += weight, over the chip's region at this level
__global__ __launch_bounds__(256) void blend_accumulate_kernel(const short* g, const float* wgt, int lw, int lh, int ox, int oy, short* dl, float* dw, int DW) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= lw) return;
    const float wv = wgt[(size_t)y * lw + x];
    const size_t di = (size_t)(oy + y) * DW + (ox + x);
    const short* s = g + ((size_t)y * lw + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) dl[di * 3 + c] = (short)(dl[di * 3 + c] + (short)((float)s[c] * wv));
    dw[di] += wv;
}

__global__ __launch_bounds__(256) void blend_normalize_kernel(short* dl, const float* dw, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = dw[i] + 1e-5f;
#pragma unroll
    for (int c = 0; c < 3; c++) dl[i * 3 + c] = (short)((float)dl[i * 3 + c] / d);
}

__global__ __launch_bounds__(256) void blend_finalize_kernel(const short* dl, const float* dw, int Wp, int W, uint8_t* out, int ows, int row0 = 0) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = row0 + blockIdx.y;      // canvas row; `out` starts at canvas row row0
    if (x >= W) return;
    const size_t di = (size_t)y * Wp + x;
    uint8_t* o = out + (size_t)blockIdx.y * ows + 3 * x;
    if (!(dw[di] > 1e-5f)) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
#pragma unroll
    for (int c = 0; c < 3; c++) { const int v = dl[di * 3 + c]; o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}

inline dim3 grid2(int w, int h) { return dim3((unsigned)((w + 255) / 256), (unsigned)h); }

// Active windows of one chip (ChipP::cwin / twin), one axis at a time.  n_l = extent of level l, S_0 = the owned pixels' range in region
// coordinates (bb == NULL: everything).  Every set is a superset of what is needed, so a window can only cost time, never change a value:
//   S_l    where the weight of level l can be non-zero: REDUCE output q sees the inputs 2q - 2 .. 2q + 2 (reflected indices fall on inputs
//          the unreflected ones already reach), so S_l+1 = [floor((a - 2) / 2) - 1, floor((b + 2) / 2) + 1], one more on each side for slack;
//   T_l    the accumulation's threads: 2 x 2 blocks of level l below the top level ([a >> 1, b >> 1]), pixels at the top level;
//   F_l    the pixels of level l those threads read (Gaussian + weight): the blocks themselves;
//   E_l+1  the pixels of level l + 1 the EXPAND of those blocks reads: T_l widened by one;
//   C_l    what is computed of level l >= 1: F_l, E_l and the inputs of the REDUCE that forms C_l+1 ([2a - 2, 2b + 2]).
// Everything outside C_l stays unwritten in the batch's pyramid buffers and is never read (the pair kernels' second output may be formed
// from such pixels when C_l+1 starts at an odd column; it lies outside C_l+1 and is never read either).
// nlo / nhi (not NULL: a stripe of the canvas is blended): per level the canvas rows whose pyramid values the stripe's output depends on; the
// accumulation windows are cut to them, a level whose window is empty then has y1 < y0 (and adds nothing to F / E / C).  Returns false when
// no level of the chip is left: the chip adds nothing to the stripe.
bool chip_windows(ChipP& c, int nb, const int* bb, const int* nlo = nullptr, const int* nhi = nullptr) {
    const int L = nb < MAX_BANDS ? nb : MAX_BANDS;
    if (!bb || nb > MAX_BANDS) {
        for (int l = 0; l <= L; l++) {
            c.cwin[l] = Win{0, 0, (c.rw >> l) - 1, (c.rh >> l) - 1};
            c.twin[l] = l < nb ? Win{0, 0, (c.rw >> (l + 1)) - 1, (c.rh >> (l + 1)) - 1} : c.cwin[l];
        }
        return true;
    }
    bool any = false;
    for (int axis = 0; axis < 2; axis++) {
        const int dim = axis ? c.rh : c.rw, off = axis ? c.top : c.left;
        int Sa[MAX_BANDS + 1], Sb[MAX_BANDS + 1], Ta[MAX_BANDS + 1], Tb[MAX_BANDS + 1], Fa[MAX_BANDS + 1], Fb[MAX_BANDS + 1], Ea[MAX_BANDS + 2], Eb[MAX_BANDS + 2];
        int Ca[MAX_BANDS + 2], Cb[MAX_BANDS + 2];
        auto clip = [](int& a, int& b, int n) { if (a < 0) a = 0; if (b > n - 1) b = n - 1; if (a > b) { a = a < n ? a : n - 1; b = a; } };
        auto uni = [](int& a, int& b, int a2, int b2) { if (a2 > b2) return; if (a > b) { a = a2; b = b2; return; } a = a2 < a ? a2 : a; b = b2 > b ? b2 : b; };
        Sa[0] = bb[axis] + off; Sb[0] = bb[2 + axis] + off;
        clip(Sa[0], Sb[0], dim);
        for (int l = 0; l < nb; l++) {
            Sa[l + 1] = ((Sa[l] - 2) >> 1) - 1; Sb[l + 1] = ((Sb[l] + 2) >> 1) + 1;
            clip(Sa[l + 1], Sb[l + 1], dim >> (l + 1));
        }
        for (int l = 0; l <= nb + 1; l++) { Ea[l] = 0; Eb[l] = -1; }
        for (int l = 0; l <= nb; l++) {
            if (l < nb) { Ta[l] = Sa[l] >> 1; Tb[l] = Sb[l] >> 1; } else { Ta[l] = Sa[l]; Tb[l] = Sb[l]; }
            if (axis == 1 && nlo) {
                // the threads whose canvas rows meet nlo[l] .. nhi[l]: thread t covers the level's rows 2t, 2t + 1 below the top level
                const int oy = c.tly >> l, lo = nlo[l] - oy, hi = nhi[l] - oy;
                const int ta = l < nb ? (lo >= 1 ? lo >> 1 : 0) : (lo > 0 ? lo : 0), tb = l < nb ? (hi >= 0 ? hi >> 1 : -1) : hi;
                if (ta > Ta[l]) Ta[l] = ta;
                if (tb < Tb[l]) Tb[l] = tb;
            }
            if (Ta[l] > Tb[l]) { Ta[l] = 0; Tb[l] = -1; Fa[l] = 0; Fb[l] = -1; continue; }
            if (axis == 1) any = true;
            if (l < nb) {
                Fa[l] = 2 * Ta[l]; Fb[l] = 2 * Tb[l] + 1;
                Ea[l + 1] = Ta[l] - 1; Eb[l + 1] = Tb[l] + 1;
                clip(Ea[l + 1], Eb[l + 1], dim >> (l + 1));
            } else { Fa[l] = Ta[l]; Fb[l] = Tb[l]; }
        }
        Ca[nb + 1] = 0; Cb[nb + 1] = -1;
        for (int l = nb; l >= 1; l--) {
            int a = 0, b = -1;
            uni(a, b, Fa[l], Fb[l]); uni(a, b, Ea[l], Eb[l]);
            if (l < nb && Ca[l + 1] <= Cb[l + 1]) uni(a, b, 2 * Ca[l + 1] - 2, 2 * Cb[l + 1] + 2);
            if (a <= b) clip(a, b, dim >> l);
            Ca[l] = a; Cb[l] = b;
        }
        Ca[0] = 0; Cb[0] = dim - 1;
        for (int l = 0; l <= nb; l++) {
            if (axis == 0) { c.cwin[l].x0 = Ca[l]; c.cwin[l].x1 = Cb[l]; c.twin[l].x0 = Ta[l]; c.twin[l].x1 = Tb[l]; }
            else           { c.cwin[l].y0 = Ca[l]; c.cwin[l].y1 = Cb[l]; c.twin[l].y0 = Ta[l]; c.twin[l].y1 = Tb[l]; }
        }
    }
    return any;
}

}  // namespace

// The chip pixels a chip's windows read: the 2 x 2 blocks of the level-0 accumulation (twin[0]) and the inputs of the first REDUCE over
// cwin[1] (both outputs of a thread, rows and columns 2q - 2 .. 2q + 2), taken through the two reflections the kernels apply (BORDER_REFLECT_101
// at the region's border, then BORDER_REFLECT into the chip).  Chip coordinates, inclusive.
static void chip_pixel_window(const ChipP& c, int& x0, int& y0, int& x1, int& y1) {
    for (int axis = 0; axis < 2; axis++) {
        const int rdim = axis ? c.rh : c.rw, cdim = axis ? c.ch : c.cw, off = axis ? c.top : c.left;
        const int ta = axis ? c.twin[0].y0 : c.twin[0].x0, tb = axis ? c.twin[0].y1 : c.twin[0].x1;
        int ca = axis ? c.cwin[1].y0 : (c.cwin[1].x0 & ~1), cb = axis ? c.cwin[1].y1 : (c.cwin[1].x1 | 1);
        int a = 0, b = -1;                                    // (a stripe may leave either window empty: y1 < y0)
        if (ta <= tb) { a = 2 * ta; b = 2 * tb + 1; }
        if (ca <= cb) { if (a > b) { a = 2 * ca - 2; b = 2 * cb + 2; } else { a = a < 2 * ca - 2 ? a : 2 * ca - 2; b = b > 2 * cb + 2 ? b : 2 * cb + 2; } }
        if (a > b) { if (axis == 0) { x0 = 0; x1 = -1; } else { y0 = 0; y1 = -1; } continue; }
        if (a < 0) { b = b > -a ? b : -a; a = 0; }
        if (b > rdim - 1) { const int m = 2 * (rdim - 1) - b; a = a < m ? a : m; b = rdim - 1; }
        if (a < 0) a = 0;
        a -= off; b -= off;
        if (a < 0) { b = b > -a - 1 ? b : -a - 1; a = 0; }
        if (b > cdim - 1) { const int m = 2 * cdim - 1 - b; a = a < m ? a : m; b = cdim - 1; }
        if (a < 0) a = 0;
        if (b < a) b = a;
        if (axis == 0) { x0 = a; x1 = b; } else { y0 = a; y1 = b; }
    }
}

// chips / masks: host pointers (staged one chip at a time) when on_device == 0, device pointers otherwise
// The canvas rows of every pyramid level that the output rows row0 .. row0 + rows - 1 depend on (a stripe of the canvas: one rank's part of
// LaplacianPyramidBlending).  The collapse forms level l from its Laplacian and EXPAND of level l + 1: fine row Y reads the coarse rows
// (Y >> 1) - 1 .. (Y >> 1) + 1, so N_0 = the stripe, N_l+1 = [(a >> 1) - 1, (b >> 1) + 1]; below the top level the ranges are widened to whole
// 2 x 2 blocks (the accumulation's threads).  Everything a rank forms is what the whole canvas holds there: the canvas geometry (padded
// size, level count, the chips' regions) is the full canvas's, only rows are left out.
static void stripe_levels(int row0, int rows, int nb, int Hp, std::vector<int>& nlo, std::vector<int>& nhi) {
    nlo.assign(nb + 1, 0); nhi.assign(nb + 1, 0);
    nlo[0] = row0; nhi[0] = row0 + rows - 1;
    for (int l = 0; l <= nb; l++) {
        const int hl = Hp >> l;
        if (l > 0) { nlo[l] = (nlo[l - 1] >> 1) - 1; nhi[l] = (nhi[l - 1] >> 1) + 1; }
        if (l < nb) { nlo[l] &= ~1; nhi[l] |= 1; }
        if (nlo[l] < 0) nlo[l] = 0;
        if (nhi[l] > hl - 1) nhi[l] = hl - 1;
    }
}
// ... and the canvas rows whose ownership (FindMasksByDistMap) those values can depend on: a level-l value sees 2^l q -+ (2^(l+1) - 2) rows of
// level 0 through its l REDUCE steps, its Laplacian one more level; 8 * 2^l on either side covers both with room to spare
static void stripe_mask_rows(const std::vector<int>& nlo, const std::vector<int>& nhi, int H, int& r0, int& r1) {
    r0 = nlo[0]; r1 = nhi[0];
    for (size_t l = 0; l < nlo.size(); l++) {
        const long long a = ((long long)nlo[l] << l) - (8ll << l), b = (((long long)nhi[l] + 1) << l) - 1 + (8ll << l);
        if (a < r0) r0 = a < 0 ? 0 : (int)a;
        if (b > r1) r1 = b > H - 1 ? H - 1 : (int)b;
    }
    if (r1 > H - 1) r1 = H - 1;
}

// chips / masks: host pointers (staged one chip at a time) when on_device == 0, device pointers otherwise
static int blend_core(mi355_ctx* ctx, const uint8_t* const* chips, const uint8_t* const* masks, int on_device, const mi355_chip_info* info, int n,
                      int W, int H, int band, uint8_t** out, int* ow, int* oh, int* ows_out, uint8_t* d_user = nullptr, int user_ws = 0,
                      const int* owned_bbox = nullptr, int deferred_pixels = 0, int row0 = 0, int rows = -1) {
    // deferred_pixels: the chips hold no pixels yet (mi_chips_and_masks_dev(.., defer_pixels)): each chip's are made here, inside the part
    // of the chip its active windows read (chip_pixel_window)
    // owned_bbox != NULL (the masks are FindMasksByDistMap's, made on this device): per chip the box of its non-zero mask bytes -- a chip
    // that owns nothing is left out, the others work inside their active windows (chip_windows)
    // d_user != NULL: the finished canvas goes to the caller's device buffer (rows of user_ws bytes) and nothing is copied to the host
    // rows >= 0: only the canvas rows row0 .. row0 + rows - 1 are formed (d_user then starts at row row0); needs owned_bbox
    if (n < 0 || (n > 0 && (!chips || !masks || !info)) || W <= 0 || H <= 0 || band < 0 || (!out && !d_user)) { ctx->set_error("multiband_blend: bad arguments"); return MI355_ERR_ARG; }
    const hipStream_t st = ctx->stream;
    int nb = (int)std::ceil(std::log((double)(W > H ? W : H)) / std::log(2.0));
    if (nb > band) nb = band;
    if (nb < 0) nb = 0;
    const int al = 1 << nb;
    const int Wp = (W + al - 1) / al * al, Hp = (H + al - 1) / al * al;
    const bool striped = rows >= 0 && !(row0 == 0 && rows == H);
    if (striped && (row0 < 0 || rows < 1 || row0 + rows > H || !owned_bbox || !d_user || nb > MAX_BANDS || nb < 1)) { ctx->set_error("multiband_blend: bad stripe"); return MI355_ERR_ARG; }
    if (!striped) { row0 = 0; rows = H; }
    // rows of level l held by the canvas pyramids: all of them, or the stripe's (stripe_levels).  loff: level offsets in pixels inside one
    // pyramid buffer; voff: the same minus the rows left out above the stripe, so that canvas coordinates index the buffers unchanged
    std::vector<int> nlo(nb + 1, 0), nhi(nb + 1, 0);
    for (int l = 0; l <= nb; l++) nhi[l] = (Hp >> l) - 1;
    if (striped) stripe_levels(row0, rows, nb, Hp, nlo, nhi);
    std::vector<size_t> loff(nb + 2, 0);
    std::vector<long long> voff(nb + 1, 0);
    for (int l = 0; l <= nb; l++) {
        loff[l + 1] = loff[l] + (size_t)(Wp >> l) * (size_t)(nhi[l] - nlo[l] + 1);
        voff[l] = (long long)loff[l] - (long long)nlo[l] * (Wp >> l);
    }
    DevBuf& dlap = ctx->buf("blend_dst_lap");
    DevBuf& dwgt = ctx->buf("blend_dst_w");
    MI_HIP(dlap.reserve(loff[nb + 1] * 3 * sizeof(short)));
    MI_HIP(dwgt.reserve(loff[nb + 1] * sizeof(float)));
    auto vlap = [&](int l) { return dlap.as<short>() + voff[l] * 3; };      // level l of the canvas Laplacian / weight pyramid, addressed by canvas coordinates
    auto vwgt = [&](int l) { return dwgt.as<float>() + voff[l]; };
    DevBuf& dchip = ctx->buf("blend_chip");
    DevBuf& dmask = ctx->buf("blend_mask");
    DevBuf& glap = ctx->buf("blend_src_lap");
    DevBuf& gwgt = ctx->buf("blend_src_w");
    DevBuf& dpar = ctx->buf("blend_chip_params");
    // geometry of every chip (MultiBandBlender::feed: gap 3 * 2^bands, corners snapped to the level grid, region pulled back inside the canvas)
    struct Geo { int k, tlx, tly; };
    std::vector<ChipP> par; std::vector<Geo> geo;
    par.reserve(n); geo.reserve(n);
    for (int k = 0; k < n; k++) {
        const int cw = info[k].w, chh = info[k].h, x0 = info[k].x0, y0 = info[k].y0;
        if (cw <= 0 || chh <= 0) continue;
        if (owned_bbox && (owned_bbox[4 * k + 2] < owned_bbox[4 * k] || owned_bbox[4 * k + 3] < owned_bbox[4 * k + 1])) continue;      // all weights +0: adds nothing
        const int gap = 3 * al;
        int tlx = x0 - gap > 0 ? x0 - gap : 0, tly = y0 - gap > 0 ? y0 - gap : 0;
        int brx = x0 + cw + gap < Wp ? x0 + cw + gap : Wp, bry = y0 + chh + gap < Hp ? y0 + chh + gap : Hp;
        tlx = (tlx >> nb) << nb; tly = (tly >> nb) << nb;
        int rw = brx - tlx, rh = bry - tly;
        rw += (al - rw % al) % al;
        rh += (al - rh % al) % al;
        brx = tlx + rw; bry = tly + rh;
        const int dx = brx - Wp > 0 ? brx - Wp : 0, dy = bry - Hp > 0 ? bry - Hp : 0;
        tlx -= dx; tly -= dy;
        if (tlx < 0 || tly < 0 || rw <= 0 || rh <= 0) { ctx->set_error("multiband_blend: chip outside the canvas"); return MI355_ERR_ARG; }
        ChipP c; memset(&c, 0, sizeof(c));
        c.chip = chips[k]; c.mask = masks[k];
        c.cw = cw; c.ch = chh; c.cws = (cw * 3 + 3) & ~3; c.mws = (cw + 3) & ~3;
        c.left = x0 - tlx; c.top = y0 - tly; c.rw = rw; c.rh = rh; c.tlx = tlx; c.tly = tly;
        if (!chip_windows(c, nb, owned_bbox ? owned_bbox + 4 * k : nullptr, striped ? nlo.data() : nullptr, striped ? nhi.data() : nullptr)) continue;      // nothing of it reaches the stripe
        par.push_back(c); geo.push_back({k, tlx, tly});
    }
    const int nc = (int)par.size();
    // batches of up to 32 chips whose levels >= 1 (10 bytes per pixel, a third of the region) fit 2 GB; host chips are staged per batch
    constexpr int MAXB = 32;
    const size_t tmp_budget_px = ((size_t)2 << 30) / 10;
    auto levels_px = [&](const ChipP& c) { size_t px = 0; for (int l = 1; l <= nb; l++) px += (size_t)(c.rw >> l) * (c.rh >> l); return px; };
    struct Batch { int b0, b1, maxw, maxh; size_t px, cbytes, mbytes; };
    std::vector<Batch> batches;
    size_t max_px = 0, max_cb = 0, max_mb = 0;
    for (int b0 = 0; b0 < nc;) {
        Batch bt = {b0, b0, 0, 0, 0, 0, 0};
        while (bt.b1 < nc && bt.b1 - b0 < MAXB) {
            ChipP& c = par[bt.b1];
            const size_t p = nb > 0 ? levels_px(c) : (size_t)c.rw * c.rh;
            if (bt.b1 > b0 && bt.px + p > tmp_budget_px) break;
            c.tmp = bt.px; bt.px += p;
            if (!on_device) { bt.cbytes += ((size_t)c.cws * c.ch + 15) & ~(size_t)15; bt.mbytes += ((size_t)c.mws * c.ch + 15) & ~(size_t)15; }
            bt.maxw = c.rw > bt.maxw ? c.rw : bt.maxw; bt.maxh = c.rh > bt.maxh ? c.rh : bt.maxh;
            bt.b1++;
        }
        max_px = bt.px > max_px ? bt.px : max_px; max_cb = bt.cbytes > max_cb ? bt.cbytes : max_cb; max_mb = bt.mbytes > max_mb ? bt.mbytes : max_mb;
        batches.push_back(bt);
        b0 = bt.b1;
    }
    // every buffer is sized once, up front: the work below is then pure stream-ordered copies and launches (no allocation, no
    // synchronisation between chips or batches)
    MI_HIP(glap.reserve(max_px * 3 * sizeof(short) + 16));
    MI_HIP(gwgt.reserve(max_px * sizeof(float) + 16));
    MI_HIP(dpar.reserve((size_t)(nc > 0 ? nc : 1) * sizeof(ChipP)));
    if (!on_device) { MI_HIP(dchip.reserve(max_cb + 16)); MI_HIP(dmask.reserve(max_mb + 16)); }
    if (deferred_pixels) {
        std::vector<int> ids((size_t)(nc > 0 ? nc : 1)), wins((size_t)4 * (nc > 0 ? nc : 1));
        for (int i = 0; i < nc; i++) {
            int x0 = 0, y0 = 0, x1 = par[i].cw - 1, y1 = par[i].ch - 1;
            if (nb >= 1 && nb <= MAX_BANDS) chip_pixel_window(par[i], x0, y0, x1, y1);
            ids[i] = geo[i].k; wins[4 * i] = x0; wins[4 * i + 1] = y0; wins[4 * i + 2] = x1; wins[4 * i + 3] = y1;
        }
        const int rc = mi_chip_pixels_prepare(ctx, nc, ids.data(), wins.data());
        if (rc != MI355_OK) return rc;
    }
    if (on_device && nb > 0 && nc > 0) {
        MI_HIP(hipMemcpyAsync(dpar.as<ChipP>(), par.data(), (size_t)nc * sizeof(ChipP), hipMemcpyHostToDevice, st));   // one copy, not one per batch
        MI_HIP(hipStreamSynchronize(st));      // `par` is a local and the device-canvas path returns without another wait: the copy must have read it (the stream holds little here: the stage before ended with a wait)
    }
    // the canvas pyramids start from zero (after the wait above, so that the host does not sit through them)
    MI_HIP(hipMemsetAsync(dlap.p, 0, loff[nb + 1] * 3 * sizeof(short), st));
    MI_HIP(hipMemsetAsync(dwgt.p, 0, loff[nb + 1] * sizeof(float), st));
    for (const Batch& bt : batches) {
        const int b0 = bt.b0, b1 = bt.b1, B = b1 - b0, maxw = bt.maxw, maxh = bt.maxh;
        if (!on_device) {
            size_t co = 0, mo = 0;
            for (int i = b0; i < b1; i++) {
                const size_t cb = (size_t)par[i].cws * par[i].ch, mb = (size_t)par[i].mws * par[i].ch;
                MI_HIP(hipMemcpyAsync(dchip.as<uint8_t>() + co, par[i].chip, cb, hipMemcpyHostToDevice, st));
                MI_HIP(hipMemcpyAsync(dmask.as<uint8_t>() + mo, par[i].mask, mb, hipMemcpyHostToDevice, st));
                par[i].chip = dchip.as<uint8_t>() + co; par[i].mask = dmask.as<uint8_t>() + mo;
                co += (cb + 15) & ~(size_t)15; mo += (mb + 15) & ~(size_t)15;
            }
        }
        if (deferred_pixels) { const int rc = mi_chip_pixels_launch(ctx, b0, B); if (rc != MI355_OK) return rc; }      // the batch's chip pixels, one launch
        short* g = glap.as<short>();
        float* wp = gwgt.as<float>();
        if (nb == 0) {
            // no pyramid: level 0 is the only level; it is materialised and accumulated (the path of bands = 0)
            for (int i = b0; i < b1; i++) {
                const ChipP& c = par[i];
                hipLaunchKernelGGL(blend_prep_kernel, grid2(c.rw, c.rh), dim3(256), 0, st, c.chip, c.cws, c.mask, c.mws, c.cw, c.ch, c.left, c.top, c.rw, c.rh, g + c.tmp * 3, wp + c.tmp);
                hipLaunchKernelGGL(blend_accumulate_kernel, grid2(c.rw, c.rh), dim3(256), 0, st, g + c.tmp * 3, wp + c.tmp, c.rw, c.rh, geo[i].tlx, geo[i].tly,
                                   vlap(0), vwgt(0), Wp);
            }
            MI_HIP(hipGetLastError());
            continue;
        }
        // the chips' parameters of this batch (the slot of the previous batch may still be read: one slot per batch, sized once)
        const ChipP* d_par = dpar.as<ChipP>() + b0;
        if (!on_device) MI_HIP(hipMemcpyAsync(dpar.as<ChipP>() + b0, par.data() + b0, (size_t)B * sizeof(ChipP), hipMemcpyHostToDevice, st));      // (device chips: all of them at once, above)
        // REDUCE chains of the whole batch: independent of one another and of the canvas
        // (grids: the largest active window of the batch at that level, in threads of two outputs)
        auto red_grid = [&](int l1) {
            int tw = 1, th = 1;
            for (int i = b0; i < b1; i++) {
                const Win wn = l1 <= MAX_BANDS ? par[i].cwin[l1] : Win{0, 0, (par[i].rw >> l1) - 1, (par[i].rh >> l1) - 1};
                const int t = (wn.x1 - (wn.x0 & ~1)) / 2 + 1, hh = wn.y1 - wn.y0 + 1;
                tw = t > tw ? t : tw; th = hh > th ? hh : th;
            }
            return dim3((unsigned)((tw + 255) / 256), (unsigned)th, (unsigned)B);
        };
        (void)maxw; (void)maxh;
        hipLaunchKernelGGL(pyr_down0_batch_kernel, red_grid(1), dim3(256), 0, st, d_par, g, wp);
        for (int l = 1; l < nb; l++)
            hipLaunchKernelGGL(pyr_down_pair_batch_kernel, red_grid(l + 1), dim3(256), 0, st, d_par, l, g, wp);
        // level 0 of the whole batch at once when the masks partition the canvas (owned_bbox: they are this device's FindMasksByDistMap masks)
        const bool lap0_batched = owned_bbox != nullptr && nb <= MAX_BANDS;
        if (lap0_batched) {
            int tw = 1, th = 1;
            for (int i = b0; i < b1; i++) { const Win t = par[i].twin[0]; tw = t.x1 - t.x0 + 1 > tw ? t.x1 - t.x0 + 1 : tw; th = t.y1 - t.y0 + 1 > th ? t.y1 - t.y0 + 1 : th; }
            hipLaunchKernelGGL(blend_lap0_accumulate_batch_kernel, dim3((unsigned)((tw + 255) / 256), (unsigned)th, (unsigned)B), dim3(256), 0, st, d_par, g, vlap(0), vwgt(0), Wp);
        }
        // accumulation, chip after chip in chip order: Laplacian level l = Gaussian l - EXPAND(Gaussian l + 1), accumulated as it is formed
        for (int i = b0; i < b1; i++) {
            const ChipP& c = par[i];
            const int tlx = geo[i].tlx, tly = geo[i].tly, rw = c.rw, rh = c.rh;
            std::vector<size_t> roff(nb + 2, 0);                      // levels >= 1 behind c.tmp
            roff[1] = c.tmp;
            for (int l = 1; l < nb; l++) roff[l + 1] = roff[l] + (size_t)(rw >> l) * (rh >> l);
            if (!lap0_batched && c.twin[0].y1 >= c.twin[0].y0)
                hipLaunchKernelGGL(blend_lap0_accumulate_kernel, grid2(c.twin[0].x1 - c.twin[0].x0 + 1, c.twin[0].y1 - c.twin[0].y0 + 1), dim3(256), 0, st, c, g + roff[1] * 3, tlx, tly,
                                   vlap(0), vwgt(0), Wp);
            if (nb > MAX_BANDS) {                                             // (band > 16 on a canvas that allows it:) level by level
                for (int l = 1; l < nb; l++)
                    hipLaunchKernelGGL(blend_lap_accumulate_kernel, grid2(rw >> (l + 1), rh >> (l + 1)), dim3(256), 0, st, g + roff[l + 1] * 3, rw >> (l + 1), rh >> (l + 1),
                                       g + roff[l] * 3, wp + roff[l], tlx >> l, tly >> l, vlap(l), vwgt(l), Wp >> l);
                hipLaunchKernelGGL(blend_accumulate_kernel, grid2(rw >> nb, rh >> nb), dim3(256), 0, st, g + roff[nb] * 3, wp + roff[nb], rw >> nb, rh >> nb, tlx >> nb, tly >> nb,
                                   vlap(nb), vwgt(nb), Wp >> nb);
                continue;
            }
            LapLevels L; memset(&L, 0, sizeof(L));
            int grows = 0, maxw = 0;                                           // (an empty window -- a stripe -- has y1 = y0 - 1: no rows)
            for (int l = 1; l < nb; l++) {                                     // Laplacian level l: one thread per pixel of level l + 1
                const int i = L.n++;
                L.row0[i] = grows; L.w[i] = rw >> (l + 1); L.h[i] = rh >> (l + 1); L.ox[i] = tlx >> l; L.oy[i] = tly >> l; L.DW[i] = Wp >> l;
                L.fine[i] = roff[l]; L.coarse[i] = roff[l + 1]; L.dst[i] = voff[l];
                const Win t = c.twin[l];
                L.x0[i] = t.x0; L.y0[i] = t.y0; L.x1[i] = t.x1;
                grows += t.y1 - t.y0 + 1; maxw = t.x1 - t.x0 + 1 > maxw ? t.x1 - t.x0 + 1 : maxw;
            }
            {
                const int i = L.n++;
                L.row0[i] = grows; L.w[i] = rw >> nb; L.h[i] = rh >> nb; L.ox[i] = tlx >> nb; L.oy[i] = tly >> nb; L.DW[i] = Wp >> nb;
                L.fine[i] = roff[nb]; L.coarse[i] = 0; L.dst[i] = voff[nb];
                const Win t = c.twin[nb];
                L.x0[i] = t.x0; L.y0[i] = t.y0; L.x1[i] = t.x1;
                grows += t.y1 - t.y0 + 1; maxw = t.x1 - t.x0 + 1 > maxw ? t.x1 - t.x0 + 1 : maxw;
            }
            L.row0[L.n] = grows;
            if (grows > 0 && maxw > 0) hipLaunchKernelGGL(blend_lap_levels_kernel, grid2(maxw, grows), dim3(256), 0, st, L, g, wp, dlap.as<short>(), dwgt.as<float>());
        }
        MI_HIP(hipGetLastError());
    }
    for (int l = 0; l <= nb; l++) {
        const size_t cnt = (size_t)(Wp >> l) * (size_t)(nhi[l] - nlo[l] + 1);      // the level's stored rows are contiguous from loff[l]
        hipLaunchKernelGGL(blend_normalize_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, dlap.as<short>() + loff[l] * 3, dwgt.as<float>() + loff[l], cnt);
    }
    for (int l = nb - 1; l >= 0; l--)       // collapse: rows nlo[l] .. nhi[l] of level l from the rows of level l + 1 they read (all stored: stripe_levels)
        hipLaunchKernelGGL((pyr_up16_combine_kernel<false>), grid2(Wp >> l, nhi[l] - nlo[l] + 1), dim3(256), 0, st, vlap(l + 1), Wp >> (l + 1), Hp >> (l + 1), vlap(l), nlo[l]);
    const int ows = d_user ? user_ws : (W * 3 + 3) & ~3;
    if (d_user) {
        MI_HIP(hipMemsetAsync(d_user, 0, (size_t)ows * rows, st));
        hipLaunchKernelGGL(blend_finalize_kernel, grid2(W, rows), dim3(256), 0, st, vlap(0), vwgt(0), Wp, W, d_user, ows, row0);
        MI_HIP(hipGetLastError());
        if (ow) *ow = W;
        if (oh) *oh = rows;
        if (ows_out) *ows_out = ows;
        return MI355_OK;
    }
    DevBuf& dout = ctx->buf("blend_out");
    MI_HIP(dout.reserve((size_t)ows * H));
    MI_HIP(hipMemsetAsync(dout.p, 0, (size_t)ows * H, st));
    hipLaunchKernelGGL(blend_finalize_kernel, grid2(W, H), dim3(256), 0, st, dlap.as<short>(), dwgt.as<float>(), Wp, W, dout.as<uint8_t>(), ows);
    MI_HIP(hipGetLastError());
    uint8_t* host = (uint8_t*)malloc((size_t)ows * H);
    if (!host) return MI355_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(host, dout.p, (size_t)ows * H, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(host); ctx->set_error(std::string("multiband_blend: ") + hipGetErrorString(e)); return MI355_ERR_DEVICE; }
    *out = host;
    if (ow) *ow = W;
    if (oh) *oh = H;
    if (ows_out) *ows_out = ows;
    return MI355_OK;
}

int mi_multiband_blend(mi355_ctx* ctx, const uint8_t* const* chips, const uint8_t* const* masks, const mi355_chip_info* info, int n,
                       int W, int H, int band, uint8_t** out, int* ow, int* oh, int* ows_out) {
    return blend_core(ctx, chips, masks, 0, info, n, W, H, band, out, ow, oh, ows_out);
}

// The whole of LaplacianPyramidBlending (MosaicImage.cpp:2205-2510) without leaving the device between its stages: chips
// and masks (mi_chips_and_masks_dev) feed the blender straight from HBM; only the finished canvas goes back to the host.
int mi_mosaic_blended(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                      const uint8_t* keep, int band, uint8_t** out, int* ow, int* oh, int* ows_out) {
    std::vector<size_t> chip_off, mask_off;
    int nv = 0, cw = 0, ch = 0;
    mi355_chip_info* ci = nullptr;
    std::vector<int> bbox;
    int rc = mi_chips_and_masks_dev(ctx, imgs, w, h, ws, n, h9s, keep, 1, &nv, &ci, chip_off, mask_off, &cw, &ch, 0, &bbox, 1);
    if (rc != MI355_OK) { free(ci); return rc; }
    std::vector<const uint8_t*> dc(nv > 0 ? nv : 1), dm(nv > 0 ? nv : 1);
    for (int v = 0; v < nv; v++) { dc[v] = ctx->buf("chip_imgs").as<uint8_t>() + chip_off[v]; dm[v] = ctx->buf("chip_masks").as<uint8_t>() + mask_off[v]; }
    rc = blend_core(ctx, dc.data(), dm.data(), 1, ci, nv, cw, ch, band, out, ow, oh, ows_out, nullptr, 0, (int)bbox.size() == 4 * nv && nv > 0 ? bbox.data() : nullptr, 1);
    free(ci);
    return rc;
}

// The same with the survey resident in HBM: device frames in, device canvas out (C5: frames + chips + masks + distance maps + both
// pyramid sets co-resident).  Enqueues on the ctx stream.
// row0, rows: one STRIPE of the canvas (rows < 0: all of it) -- a rank's part of the blended mosaic, the counterpart of mi_mosaic_refined_dev's
// stripes.  d_canvas then holds the rows row0 .. row0 + rows - 1 only.  The canvas geometry stays the whole canvas's; the rank forms the chips
// that reach its rows (+ the pyramids' reach: stripe_levels / stripe_mask_rows), their ownership there, and the rows of every canvas pyramid level
// its output rows depend on -- the same values the whole canvas holds there, so stripes put side by side are the whole canvas byte for byte.
int mi_mosaic_blended_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                          const uint8_t* keep, int band, uint8_t* d_canvas, int cw, int ch, int cws, int row0, int rows, uint8_t* cover_only) {
    if (!d_canvas && !cover_only) return MI355_ERR_ARG;
    int lw = 0, lh = 0;
    { int rc = mi_blend_layout(w, h, n, h9s, keep, &lw, &lh); if (rc != MI355_OK) return rc; }
    if (lw != cw || lh != ch || cws < 3 * cw || (cws & 3)) { ctx->set_error("mosaic_blended_dev: canvas geometry does not match mi355_blend_layout"); return MI355_ERR_ARG; }
    if (rows < 0) { row0 = 0; rows = ch; }
    if (row0 < 0 || rows < 1 || row0 + rows > ch) { ctx->set_error("mosaic_blended_dev: bad stripe"); return MI355_ERR_ARG; }
    int nb = (int)std::ceil(std::log((double)(cw > ch ? cw : ch)) / std::log(2.0));
    if (nb > band) nb = band;
    if (nb < 0) nb = 0;
    bool striped = !(row0 == 0 && rows == ch);
    if (striped && (nb < 1 || nb > MAX_BANDS)) striped = false;      // (no pyramid, or more levels than the windows hold:) the whole canvas is formed and the stripe copied out
    int mr0 = 0, mr1 = ch - 1;
    if (striped) {
        const int al = 1 << nb, Hp = (ch + al - 1) / al * al;
        std::vector<int> nlo, nhi;
        stripe_levels(row0, rows, nb, Hp, nlo, nhi);
        stripe_mask_rows(nlo, nhi, ch, mr0, mr1);
    }
    std::vector<size_t> chip_off, mask_off;
    int nv = 0, gw = 0, gh = 0;
    mi355_chip_info* ci = nullptr;
    std::vector<int> bbox;
    int rc = mi_chips_and_masks_dev(ctx, d_imgs, w, h, ws, n, h9s, keep, 1, &nv, &ci, chip_off, mask_off, &gw, &gh, 1, &bbox, 1, mr0, mr1, cover_only);
    if (rc != MI355_OK) { free(ci); return rc; }
    if (cover_only) return MI355_OK;                     // mi355_mosaic_stripe_cover: the frames whose chips reach the stripe (+ the pyramids' reach) are marked, nothing was enqueued
    std::vector<const uint8_t*> dc(nv > 0 ? nv : 1), dm(nv > 0 ? nv : 1);
    for (int v = 0; v < nv; v++) { dc[v] = ctx->buf("chip_imgs").as<uint8_t>() + chip_off[v]; dm[v] = ctx->buf("chip_masks").as<uint8_t>() + mask_off[v]; }
    const int* bb = (int)bbox.size() == 4 * nv && nv > 0 ? bbox.data() : nullptr;
    if (striped || (row0 == 0 && rows == ch)) {
        rc = blend_core(ctx, dc.data(), dm.data(), 1, ci, nv, gw, gh, band, nullptr, nullptr, nullptr, nullptr, d_canvas, cws, bb, 1, row0, striped ? rows : -1);
    } else {
        DevBuf& full = ctx->buf("blend_full_canvas");
        hipError_t e = full.reserve((size_t)cws * ch);
        if (e != hipSuccess) { free(ci); ctx->set_error(std::string("mosaic_blended_dev: ") + hipGetErrorString(e)); return MI355_ERR_NOMEM; }
        rc = blend_core(ctx, dc.data(), dm.data(), 1, ci, nv, gw, gh, band, nullptr, nullptr, nullptr, nullptr, full.as<uint8_t>(), cws, bb, 1);
        if (rc == MI355_OK && hipMemcpyAsync(d_canvas, full.as<uint8_t>() + (size_t)row0 * cws, (size_t)rows * cws, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = MI355_ERR_DEVICE;
    }
    free(ci);
    return rc;
}
