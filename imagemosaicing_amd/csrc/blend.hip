// csrc/blend.hip -- multiband blend of the warped chips, replacing
//   detail::MultiBandBlender blender(false, band); prepare / feed per chip / blend; convertTo(CV_8U)
//   (MosaicImage.cpp:2296-2299, 2451-2486).
// The arithmetic is OpenCV 2.4.0's (binaries only; the reference commits no blended output).  The definition implemented here --
// 16-bit Laplacian pyramids, float weight pyramids, [1 4 6 4 1] REDUCE / EXPAND in integer arithmetic, every rounding and border --
// is the one stated at the top of oracle/oracle_blend.c, which also lists what was checked against the reference's DLLs and the one
// known divergence (the binary's reassociated float REDUCE); the parity test compares the output bytes with that oracle.
// All kernels are streaming stencils over at most a few hundred MB.  Per chip: prep, 5 x paired REDUCE (both pyramids, two outputs per
// thread from 32-bit loads), 5 x Laplacian + accumulate (a 2 x 2 fine block per thread, never stored), the top level's accumulate.
#include "common.h"
#include <cmath>

namespace {

__device__ __forceinline__ int reflect101d(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }
__device__ __forceinline__ int reflectd(int p, int n) { while (p < 0 || p >= n) { if (p < 0) p = -p - 1; else p = 2 * n - 1 - p; } return p; }
__device__ __forceinline__ short sat16d(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

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

// REDUCE: i16 x 3 in integers (so the 5x5 product form equals the oracle's rows-then-columns form: rows [1 4 6 4 1] . pixels, then
// columns, (sum + 128) >> 8) and f32 weights in the oracle's order (6 c + 4 (l + r) + ll + rr per row, the same over the rows, / 256).
// REDUCE of both pyramids of a chip in one launch, two horizontally adjacent outputs per thread: their 5-tap windows share three of the
// seven source columns, and away from the left / right border those seven pixels are 42 contiguous, 4-byte aligned bytes (11 32-bit
// loads per row instead of 30 16-bit ones).  The sums are the ones of pyr_down16_kernel / pyr_down_f_kernel, term for term.
__global__ __launch_bounds__(256) void pyr_down_pair_kernel(const short* src, const float* srcw, int w, int h, short* dst, float* dstw) {
    const int dw = w >> 1, x0 = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
    if (x0 >= dw) return;
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

// fine = sat16(fine - EXPAND(coarse)) (SUB) or sat16(EXPAND(coarse) + fine); coarse is w x h, fine 2w x 2h
template <bool SUB>
__global__ __launch_bounds__(256) void pyr_up16_combine_kernel(const short* coarse, int w, int h, short* fine) {
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y;
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
__global__ __launch_bounds__(256) void blend_lap_accumulate_kernel(const short* coarse, int w, int h, const short* fine, const float* wgt, int ox, int oy,
                                                                   short* dl, float* dw, int DW) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int ym = (y == 0) ? (h > 1 ? 1 : 0) : y - 1, yp = (y == h - 1) ? h - 1 : y + 1;
    const short* rows[3] = {coarse + (size_t)ym * w * 3, coarse + (size_t)y * w * 3, coarse + (size_t)yp * w * 3};
    int he[3][3], ho[3][3];                               // horizontal EXPAND values at fine columns 2x (even) and 2x + 1 (odd), per row and channel
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) { he[r][c] = up_h(rows[r], w, 2 * x, c); ho[r][c] = up_h(rows[r], w, 2 * x + 1, c); }
    const int FW = 2 * w;
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

__global__ __launch_bounds__(256) void blend_finalize_kernel(const short* dl, const float* dw, int Wp, int W, uint8_t* out, int ows) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t di = (size_t)y * Wp + x;
    uint8_t* o = out + (size_t)y * ows + 3 * x;
    if (!(dw[di] > 1e-5f)) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
#pragma unroll
    for (int c = 0; c < 3; c++) { const int v = dl[di * 3 + c]; o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}

inline dim3 grid2(int w, int h) { return dim3((unsigned)((w + 255) / 256), (unsigned)h); }

}  // namespace

// chips / masks: host pointers (staged one chip at a time) when on_device == 0, device pointers otherwise
static int blend_core(mi355_ctx* ctx, const uint8_t* const* chips, const uint8_t* const* masks, int on_device, const mi355_chip_info* info, int n,
                      int W, int H, int band, uint8_t** out, int* ow, int* oh, int* ows_out, uint8_t* d_user = nullptr, int user_ws = 0) {
    // d_user != NULL: the finished canvas goes to the caller's device buffer (rows of user_ws bytes) and nothing is copied to the host
    if (n < 0 || (n > 0 && (!chips || !masks || !info)) || W <= 0 || H <= 0 || band < 0 || (!out && !d_user)) { ctx->set_error("multiband_blend: bad arguments"); return MI355_ERR_ARG; }
    const hipStream_t st = ctx->stream;
    int nb = (int)std::ceil(std::log((double)(W > H ? W : H)) / std::log(2.0));
    if (nb > band) nb = band;
    if (nb < 0) nb = 0;
    const int al = 1 << nb;
    const int Wp = (W + al - 1) / al * al, Hp = (H + al - 1) / al * al;
    std::vector<size_t> loff(nb + 2, 0);                         // level offsets in pixels inside one pyramid buffer
    for (int l = 0; l <= nb; l++) loff[l + 1] = loff[l] + (size_t)(Wp >> l) * (Hp >> l);
    DevBuf& dlap = ctx->buf("blend_dst_lap");
    DevBuf& dwgt = ctx->buf("blend_dst_w");
    MI_HIP(dlap.reserve(loff[nb + 1] * 3 * sizeof(short)));
    MI_HIP(dwgt.reserve(loff[nb + 1] * sizeof(float)));
    MI_HIP(hipMemsetAsync(dlap.p, 0, loff[nb + 1] * 3 * sizeof(short), st));
    MI_HIP(hipMemsetAsync(dwgt.p, 0, loff[nb + 1] * sizeof(float), st));
    DevBuf& dchip = ctx->buf("blend_chip");
    DevBuf& dmask = ctx->buf("blend_mask");
    DevBuf& glap = ctx->buf("blend_src_lap");
    DevBuf& gwgt = ctx->buf("blend_src_w");
    // every staging buffer is sized for the largest chip up front: the per-chip work below is then pure stream-ordered launches
    // (no allocation, no synchronisation between chips; the next chip's upload / pyramid simply queues behind this chip's kernels)
    {
        size_t max_px = 0, max_chip = 0, max_mask = 0;
        for (int k = 0; k < n; k++) {
            if (info[k].w <= 0 || info[k].h <= 0) continue;
            const size_t rw = (size_t)info[k].w + 8 * (size_t)al, rh = (size_t)info[k].h + 8 * (size_t)al;     // region <= chip + 2 x (3 al gap + al rounding)
            size_t px = 0;
            for (int l = 0; l <= nb; l++) px += (rw >> l) * (rh >> l);
            if (px > max_px) max_px = px;
            const size_t cb = (size_t)((info[k].w * 3 + 3) & ~3) * info[k].h, mb = (size_t)((info[k].w + 3) & ~3) * info[k].h;
            if (cb > max_chip) max_chip = cb;
            if (mb > max_mask) max_mask = mb;
        }
        MI_HIP(glap.reserve(max_px * 3 * sizeof(short)));
        MI_HIP(gwgt.reserve(max_px * sizeof(float)));
        if (!on_device) { MI_HIP(dchip.reserve(max_chip + 16)); MI_HIP(dmask.reserve(max_mask + 16)); }
    }
    for (int k = 0; k < n; k++) {
        const int cw = info[k].w, chh = info[k].h, x0 = info[k].x0, y0 = info[k].y0;
        if (cw <= 0 || chh <= 0) continue;
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
        const int left = x0 - tlx, top = y0 - tly;
        const int cws = (cw * 3 + 3) & ~3, mws = (cw + 3) & ~3;
        std::vector<size_t> roff(nb + 2, 0);
        for (int l = 0; l <= nb; l++) roff[l + 1] = roff[l] + (size_t)(rw >> l) * (rh >> l);
        MI_HIP(glap.reserve(roff[nb + 1] * 3 * sizeof(short)));
        MI_HIP(gwgt.reserve(roff[nb + 1] * sizeof(float)));
        const uint8_t* d_chip = chips[k];
        const uint8_t* d_mask = masks[k];
        if (!on_device) {
            MI_HIP(dchip.reserve((size_t)cws * chh));
            MI_HIP(dmask.reserve((size_t)mws * chh));
            MI_HIP(hipMemcpyAsync(dchip.p, chips[k], (size_t)cws * chh, hipMemcpyHostToDevice, st));
            MI_HIP(hipMemcpyAsync(dmask.p, masks[k], (size_t)mws * chh, hipMemcpyHostToDevice, st));
            d_chip = dchip.as<uint8_t>(); d_mask = dmask.as<uint8_t>();
        }
        short* g = glap.as<short>();
        float* wp = gwgt.as<float>();
        hipLaunchKernelGGL(blend_prep_kernel, grid2(rw, rh), dim3(256), 0, st, d_chip, cws, d_mask, mws, cw, chh, left, top, rw, rh, g, wp);
        for (int l = 0; l < nb; l++)
            hipLaunchKernelGGL(pyr_down_pair_kernel, grid2(((rw >> (l + 1)) + 1) / 2, rh >> (l + 1)), dim3(256), 0, st, g + roff[l] * 3, wp + roff[l], rw >> l, rh >> l,
                               g + roff[l + 1] * 3, wp + roff[l + 1]);
        for (int l = 0; l < nb; l++)                                // Laplacian level l = Gaussian l - EXPAND(Gaussian l + 1), accumulated as it is formed
            hipLaunchKernelGGL(blend_lap_accumulate_kernel, grid2(rw >> (l + 1), rh >> (l + 1)), dim3(256), 0, st, g + roff[l + 1] * 3, rw >> (l + 1), rh >> (l + 1),
                               g + roff[l] * 3, wp + roff[l], tlx >> l, tly >> l, dlap.as<short>() + loff[l] * 3, dwgt.as<float>() + loff[l], Wp >> l);
        hipLaunchKernelGGL(blend_accumulate_kernel, grid2(rw >> nb, rh >> nb), dim3(256), 0, st, g + roff[nb] * 3, wp + roff[nb], rw >> nb, rh >> nb, tlx >> nb, tly >> nb,
                           dlap.as<short>() + loff[nb] * 3, dwgt.as<float>() + loff[nb], Wp >> nb);
        MI_HIP(hipGetLastError());
    }
    for (int l = 0; l <= nb; l++) {
        const size_t cnt = (size_t)(Wp >> l) * (Hp >> l);
        hipLaunchKernelGGL(blend_normalize_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, dlap.as<short>() + loff[l] * 3, dwgt.as<float>() + loff[l], cnt);
    }
    for (int l = nb - 1; l >= 0; l--)
        hipLaunchKernelGGL((pyr_up16_combine_kernel<false>), grid2(Wp >> l, Hp >> l), dim3(256), 0, st, dlap.as<short>() + loff[l + 1] * 3, Wp >> (l + 1), Hp >> (l + 1), dlap.as<short>() + loff[l] * 3);
    const int ows = d_user ? user_ws : (W * 3 + 3) & ~3;
    if (d_user) {
        MI_HIP(hipMemsetAsync(d_user, 0, (size_t)ows * H, st));
        hipLaunchKernelGGL(blend_finalize_kernel, grid2(W, H), dim3(256), 0, st, dlap.as<short>(), dwgt.as<float>(), Wp, W, d_user, ows);
        MI_HIP(hipGetLastError());
        if (ow) *ow = W;
        if (oh) *oh = H;
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
    int rc = mi_chips_and_masks_dev(ctx, imgs, w, h, ws, n, h9s, keep, 1, &nv, &ci, chip_off, mask_off, &cw, &ch);
    if (rc != MI355_OK) { free(ci); return rc; }
    std::vector<const uint8_t*> dc(nv > 0 ? nv : 1), dm(nv > 0 ? nv : 1);
    for (int v = 0; v < nv; v++) { dc[v] = ctx->buf("chip_imgs").as<uint8_t>() + chip_off[v]; dm[v] = ctx->buf("chip_masks").as<uint8_t>() + mask_off[v]; }
    rc = blend_core(ctx, dc.data(), dm.data(), 1, ci, nv, cw, ch, band, out, ow, oh, ows_out);
    free(ci);
    return rc;
}

// The same with the survey resident in HBM: device frames in, device canvas out (C5: frames + chips + masks + distance maps + both
// pyramid sets co-resident).  Enqueues on the ctx stream.
int mi_mosaic_blended_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                          const uint8_t* keep, int band, uint8_t* d_canvas, int cw, int ch, int cws) {
    if (!d_canvas) return MI355_ERR_ARG;
    int lw = 0, lh = 0;
    { int rc = mi_blend_layout(w, h, n, h9s, keep, &lw, &lh); if (rc != MI355_OK) return rc; }
    if (lw != cw || lh != ch || cws < 3 * cw || (cws & 3)) { ctx->set_error("mosaic_blended_dev: canvas geometry does not match mi355_blend_layout"); return MI355_ERR_ARG; }
    std::vector<size_t> chip_off, mask_off;
    int nv = 0, gw = 0, gh = 0;
    mi355_chip_info* ci = nullptr;
    int rc = mi_chips_and_masks_dev(ctx, d_imgs, w, h, ws, n, h9s, keep, 1, &nv, &ci, chip_off, mask_off, &gw, &gh, 1);
    if (rc != MI355_OK) { free(ci); return rc; }
    std::vector<const uint8_t*> dc(nv > 0 ? nv : 1), dm(nv > 0 ? nv : 1);
    for (int v = 0; v < nv; v++) { dc[v] = ctx->buf("chip_imgs").as<uint8_t>() + chip_off[v]; dm[v] = ctx->buf("chip_masks").as<uint8_t>() + mask_off[v]; }
    rc = blend_core(ctx, dc.data(), dm.data(), 1, ci, nv, gw, gh, band, nullptr, nullptr, nullptr, nullptr, d_canvas, cws);
    free(ci);
    return rc;
}
