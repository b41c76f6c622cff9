// csrc/warp.hip -- K9 inverse-mapping bilinear warps and K10 distance-map masks (gfx950).
//
// Replaces the reference's hand-written warps (paths relative to code/MosaicingCode/mosaicing/):
//   ImageProjectionTransform             MosaicImage.cpp:1613-1758       -> mi_warp_image
//   CMosaicByPose::MosaicImagesRefined   MosaicWithoutPos.cpp:2194-2352  -> mi_mosaic_refined_dev
//   LaplacianPyramidBlending warp stage  MosaicImage.cpp:2233-2460       -> mi_chips_and_masks (chips)
//   FindMasksByDistMap                   MosaicImage.cpp:1761-1881       -> mi_chips_and_masks (masks)
//
// Parity contract: bit-exact pixels.  The source coordinate is two true float divisions by the same
// denominator, truncated with (int); the pixel is the 4-term float expression of hm::bilin.  This TU is
// compiled with -ffp-contract=off; HIP's fp32 division is correctly rounded by default.
//
// Kernel shape: HBM-bound gather.  One thread produces 4 horizontally adjacent destination pixels
// (12 bytes = three aligned dwords for BGR) so that interior groups are written with dword stores and
// never read the canvas; only partially covered groups (image borders) fall back to byte stores, which is
// what preserves the reference's "later image overwrites earlier only where it has a valid sample".
// A wave covers 256 consecutive destination pixels of one row: its source footprint is a short, nearly
// contiguous run of the source row pair, so the 6-byte neighbourhood loads hit the same cache lines.
#include "common.h"
#include "hmath.h"
#include <memory>

namespace {

struct WarpArgs {
    const uint8_t* src; int w, h, ws;          // source image
    uint8_t* dst; int dws;                      // destination (canvas / chip / tight image), row stride
    uint8_t* mask; int mws;                     // chip validity mask (CHIP mode) or nullptr
    int x_beg, x_end, y_beg, y_end;             // inclusive destination range to visit
    float inv[9];                               // inverse homography (destination -> source)
    float dx, dy;                               // mode 0: xs = (float)xD - dx
    float sx, sy; int x0, y0;                   // mode 1 (chips): ((float)xD - dx) - sx + (float)x0
};

template <int CH>
__device__ __forceinline__ void load_pair(const uint8_t* s, float& a, float& b, float& c, float& a1, float& b1, float& c1);

// 6 bytes (two BGR pixels) with one dword + one ushort load; the device handles unaligned addresses.
template <>
__device__ __forceinline__ void load_pair<3>(const uint8_t* s, float& b0, float& g0, float& r0, float& b1, float& g1, float& r1) {
    uint32_t lo; uint16_t hi;
    __builtin_memcpy(&lo, s, 4);
    __builtin_memcpy(&hi, s + 4, 2);
    b0 = (float)(lo & 0xff); g0 = (float)((lo >> 8) & 0xff); r0 = (float)((lo >> 16) & 0xff);
    b1 = (float)(lo >> 24);  g1 = (float)(hi & 0xff);        r1 = (float)(hi >> 8);
}

// the two BGR pixel pairs of a 2 x 2 neighbourhood (rows s and s + ws): one 8-byte load per row where 8 bytes from s still lie inside the row's
// pitch (6 are used), else the 4 + 2 byte form -- half the load instructions of a sample (the canvas kernel is bound by their number)
__device__ __forceinline__ void load_quad3(const uint8_t* s, int ws, bool wide, float& b00, float& g00, float& r00, float& b01, float& g01, float& r01,
                                           float& b10, float& g10, float& r10, float& b11, float& g11, float& r11) {
    if (wide) {
        uint64_t q0, q1;
        __builtin_memcpy(&q0, s, 8);
        __builtin_memcpy(&q1, s + ws, 8);
        b00 = (float)(q0 & 0xff); g00 = (float)((q0 >> 8) & 0xff); r00 = (float)((q0 >> 16) & 0xff);
        b01 = (float)((q0 >> 24) & 0xff); g01 = (float)((q0 >> 32) & 0xff); r01 = (float)((q0 >> 40) & 0xff);
        b10 = (float)(q1 & 0xff); g10 = (float)((q1 >> 8) & 0xff); r10 = (float)((q1 >> 16) & 0xff);
        b11 = (float)((q1 >> 24) & 0xff); g11 = (float)((q1 >> 32) & 0xff); r11 = (float)((q1 >> 40) & 0xff);
    } else {
        load_pair<3>(s, b00, g00, r00, b01, g01, r01);
        load_pair<3>(s + ws, b10, g10, r10, b11, g11, r11);
    }
}

// PART (chips only): 0 = pixels and validity mask, 1 = the validity mask alone (no source read), 2 = the pixels alone (the mask bytes
// hold the ownership by then and stay as they are)
template <int CH, bool CHIP, int PART>
__device__ __forceinline__ void warp_body(const WarpArgs& a) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;          // group of 4 pixels
    const int yD = a.y_beg + blockIdx.y * blockDim.y + threadIdx.y;
    const int xg = (a.x_beg & ~3) + gx * 4;
    if (yD > a.y_end || xg > a.x_end) return;
    const float w1 = (float)(a.w - 1), h1 = (float)(a.h - 1);
    uint8_t out[4 * CH];
    bool valid[4];
    int nvalid = 0;
    // affine transform with m8 = 1: the denominator m6 x + m7 y + m8 is exactly 1.0f for every finite (x, y), and t / 1.0f = t,
    // so the two divisions can be skipped without changing a bit (uniform test on the kernel arguments)
    const bool unit_den = a.inv[6] == 0.0f && a.inv[7] == 0.0f && a.inv[8] == 1.0f;
    // no per-pixel branches: a pixel without a sample still computes from a clamped address and is masked at the store
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xD = xg + k;
        const bool inside = (xD >= a.x_beg) && (xD <= a.x_end);
        float xf, yf;
        if (CHIP) {
            xf = (((float)xD - a.dx) - a.sx) + (float)a.x0;
            yf = (((float)yD - a.dy) - a.sy) + (float)a.y0;
        } else {
            xf = (float)xD - a.dx;
            yf = (float)yD - a.dy;
        }
        float xs, ys;
        if (unit_den) { xs = a.inv[0] * xf + a.inv[1] * yf + a.inv[2]; ys = a.inv[3] * xf + a.inv[4] * yf + a.inv[5]; }
        else hm::apply_div9(a.inv, xf, yf, xs, ys);
        const bool ok = inside && (xs >= 0.0f && xs < w1 && ys >= 0.0f && ys < h1);      // also rejects NaN
        const float xc = ok ? xs : 0.0f, yc = ok ? ys : 0.0f;
        const int xi = (int)xc, yi = (int)yc;
        const float p = yc - (float)yi, q = xc - (float)xi;
        const uint8_t* s = a.src + (size_t)yi * a.ws + (size_t)CH * xi;
        if (PART == 1) {
        } else if (CH == 3) {
            float b00, g00, r00, b01, g01, r01, b10, g10, r10, b11, g11, r11;
            load_quad3(s, a.ws, 3 * xi + 8 <= a.ws, b00, g00, r00, b01, g01, r01, b10, g10, r10, b11, g11, r11);
            out[3 * k + 0] = hm::bilin(b00, b01, b10, b11, p, q);
            out[3 * k + 1] = hm::bilin(g00, g01, g10, g11, p, q);
            out[3 * k + 2] = hm::bilin(r00, r01, r10, r11, p, q);
        } else {
            uint16_t t0, t1;
            __builtin_memcpy(&t0, s, 2);
            __builtin_memcpy(&t1, s + a.ws, 2);
            out[k] = hm::bilin((float)(t0 & 0xff), (float)(t0 >> 8), (float)(t1 & 0xff), (float)(t1 >> 8), p, q);
        }
        valid[k] = ok;
        nvalid += ok ? 1 : 0;
    }
    uint8_t* drow = a.dst + (size_t)yD * a.dws + (size_t)CH * xg;
    if (PART == 1) {
        uint8_t* mrow = a.mask + (size_t)yD * a.mws + xg;
        if (xg >= a.x_beg && xg + 3 <= a.x_end) {
            *reinterpret_cast<uint32_t*>(mrow) = (valid[0] ? 0xffu : 0u) | (valid[1] ? 0xff00u : 0u) | (valid[2] ? 0xff0000u : 0u) | (valid[3] ? 0xff000000u : 0u);     // xg and mws are multiples of 4
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (xg + k >= a.x_beg && xg + k <= a.x_end) mrow[k] = valid[k] ? 255 : 0;
        }
        return;
    }
    if (nvalid == 4) {
        // whole group valid: aligned dword stores (xg % 4 == 0 and dws % 4 == 0)
        uint32_t* d32 = reinterpret_cast<uint32_t*>(drow);
        const uint32_t* o32 = reinterpret_cast<const uint32_t*>(out);
#pragma unroll
        for (int i = 0; i < CH; i++) d32[i] = o32[i];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (valid[k])
                for (int c = 0; c < CH; c++) drow[CH * k + c] = out[CH * k + c];
    }
    if (CHIP) {
        // mask = 255 where a sample exists, 0 elsewhere (MosaicImage.cpp:2345, 2444-2447); chip pixels
        // without a sample are zeroed (the reference leaves them uninitialised)
        uint8_t* mrow = a.mask + (size_t)yD * a.mws + xg;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xD = xg + k;
            if (xD < a.x_beg || xD > a.x_end) continue;
            if (PART == 0) mrow[k] = valid[k] ? 255 : 0;
            if (!valid[k]) for (int c = 0; c < CH; c++) drow[CH * k + c] = 0;
        }
    }
}

template <int CH, bool CHIP, int PART = 0>
__global__ __launch_bounds__(256) void warp_kernel(WarpArgs a) { warp_body<CH, CHIP, PART>(a); }
// the chips of a survey in one launch (blockIdx.z = chip; the grid covers the largest range): one launch per chip was ~10 us of stream time
// each whatever its size -- 2000 chips, two passes
template <int PART>
__global__ __launch_bounds__(256) void warp_chips_kernel(const WarpArgs* arr) { warp_body<3, true, PART>(arr[blockIdx.z]); }

template <int CH, bool CHIP, int PART = 0>
void launch_warp(mi355_ctx* ctx, const WarpArgs& a) {
    if (a.x_end < a.x_beg || a.y_end < a.y_beg) return;
    const int groups = (a.x_end - (a.x_beg & ~3)) / 4 + 1;
    const int rows = a.y_end - a.y_beg + 1;
    dim3 block(64, 4);
    dim3 grid((groups + 63) / 64, (rows + 3) / 4);
    const double bytes = (double)CH * ((double)(a.x_end - a.x_beg + 1) * rows) * 2.0;   // read ~P + write P
    ProfScope ps(ctx, "warp", bytes);
    hipLaunchKernelGGL((warp_kernel<CH, CHIP, PART>), grid, block, 0, ctx->stream, a);
}

inline float big() { return (float)(1 << 29); }

// corners (0,0) (w-1,0) (w-1,h-1) (0,h-1) through the two-division form
void corner_bbox(const float* m, int w, int h, float& minX, float& minY, float& maxX, float& maxY, float* quad = nullptr) {
    const float cx[4] = {0.0f, (float)(w - 1), (float)(w - 1), 0.0f};
    const float cy[4] = {0.0f, 0.0f, (float)(h - 1), (float)(h - 1)};
    for (int i = 0; i < 4; i++) {
        float X, Y;
        hm::apply_div9(m, cx[i], cy[i], X, Y);
        if (quad) { quad[2 * i] = X; quad[2 * i + 1] = Y; }
        if (X < minX) minX = X;
        if (X > maxX) maxX = X;
        if (Y < minY) minY = Y;
        if (Y > maxY) maxY = Y;
    }
}

}  // namespace

int mi_inverse_matrix_host(const float* src, int order, float* dst, float eps) {
    float t[2 * 13 * 13];
    return hm::inverse_matrix<13>(src, order, dst, eps, t);
}

// MosaicWithoutPos.cpp:2199-2249
extern "C" int mi355_mosaic_layout(const int* w, const int* h, int n, const float* h9s, int* cw, int* ch, int* cws, float* dGxy) {
    if (!w || !h || !h9s || n <= 0) return MI355_ERR_ARG;
    float minX = big(), minY = big(), maxX = -big(), maxY = -big();
    bool any = false;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if (m[8] == 0.0f) continue;                 // "invalid image" convention, MosaicWithoutPos.cpp:2205, 4646-4652
        corner_bbox(m, w[k], h[k], minX, minY, maxX, maxY);
        any = true;
    }
    if (!any) return MI355_ERR_FAILED;
    const int mw = (int)(maxX - minX + 1.5f), mh = (int)(maxY - minY + 1.5f);
    if (mw <= 0 || mh <= 0) return MI355_ERR_FAILED;
    if (cw) *cw = mw;
    if (ch) *ch = mh;
    if (cws) *cws = (mw * 3 + 3) & ~3;              // IplImage rows are padded to 4 bytes
    if (dGxy) { dGxy[0] = -minX; dGxy[1] = -minY; }
    return MI355_OK;
}

// ---- MosaicImagesRefined as ONE launch: every canvas tile is produced once -------------------------------------------------------
// The reference composites image after image, each overwriting the canvas wherever it has a valid sample
// (MosaicWithoutPos.cpp:2254-2348), so a canvas pixel ends up with the sample of the HIGHEST-index image that covers it.  A
// workgroup here owns a 128 x 16 tile of the canvas, walks the images whose canvas bounding box meets the tile in DESCENDING
// index, and takes for every pixel the first valid sample it meets: the same bytes, with one read of the winning image's
// 2 x 2 neighbourhoods and one write per canvas pixel instead of one read + one write per covering image (3.1 covering images per
// pixel in the C3 survey, ~60 at C5) and no clearing pass (pixels nobody covers are stored as zeros).  All images go through
// one launch; the result does not depend on any execution order.
struct FrameDev {
    const uint8_t* src; int w, h, ws;
    int begX, endX, begY, endY;                 // clipped canvas bounding box the reference visits for this image (:2276-2306)
    float inv[9];
    int unit_den;                               // affine with m8 = 1: the two divisions are by exactly 1.0f
};
#ifndef MT_RPL_V
#define MT_RPL_V 2
#endif
constexpr int MT_W = 128, MT_RPL = MT_RPL_V, MT_H = 8 * MT_RPL;   // canvas tile of one workgroup: 256 threads x 4 pixels x MT_RPL rows
constexpr int MT_COARSE = 256;                  // candidate lists are kept per 256 x 256 block of the canvas

// one thread per coarse block: the images whose box meets the block, highest index first
__global__ __launch_bounds__(256) void mosaic_lists_kernel(const FrameDev* fr, int n, int bx_n, int by_n, int row0, uint16_t* lists, int* counts) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= bx_n * by_n) return;
    const int by = b / bx_n, bx = b - by * bx_n;
    const int x0 = bx * MT_COARSE, x1 = x0 + MT_COARSE - 1, y0 = row0 + by * MT_COARSE, y1 = y0 + MT_COARSE - 1;
    uint16_t* l = lists + (size_t)b * n;
    int cnt = 0;
    for (int k = n - 1; k >= 0; k--) {
        const FrameDev& f = fr[k];
        if (f.begX <= x1 && f.endX >= x0 && f.begY <= y1 && f.endY >= y0) l[cnt++] = (uint16_t)k;
    }
    counts[b] = cnt;
}

// COVER: no pixel is loaded or stored -- the walk only records which frames GIVE a sample to at least one pixel of these rows (used[list
// entry] = 1).  A frame that covers a pixel but lies under a later one is never read by the real pass either, so `used` is exactly the set of
// frames the real pass dereferences: what a rank must hold to render the stripe (mi355_mosaic_stripe_cover, mode 2).
template <bool COVER>
__global__ __launch_bounds__(256) void mosaic_tile_kernel(const FrameDev* fr, int n, const uint16_t* lists, const int* counts, int bx_n,
                                                          uint8_t* canvas, int cw, int cws, int row0, int row_end, float dGx, float dGy, int* used) {
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * MT_W, ty0 = row0 + blockIdx.y * MT_H;
    // a lane owns 4 adjacent pixels in each of MT_RPL rows (rows ty0 + (tid >> 5) + 8 j): the tile's list / image loads are paid
    // once per 4 MT_RPL pixels of a lane.  No workgroup-level coupling: a wave leaves as soon as its own pixels are resolved.
    const int xg = tx0 + 4 * (tid & 31), yB = ty0 + (tid >> 5);
    const int cb = ((ty0 - row0) / MT_COARSE) * bx_n + tx0 / MT_COARSE;
    const uint16_t* list = lists + (size_t)cb * n;
    const int cnt = counts[cb];
    unsigned open = 0;                                   // bit 4 j + k: pixel k of row j still without a sample
#pragma unroll
    for (int j = 0; j < MT_RPL; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) if (yB + 8 * j < row_end && xg + k < cw) open |= 1u << (4 * j + k);
    uint32_t out[MT_RPL][3];                             // 12 bytes per row: B G R of the 4 pixels
#pragma unroll
    for (int j = 0; j < MT_RPL; j++) { out[j][0] = 0; out[j][1] = 0; out[j][2] = 0; }
    const int tx1 = tx0 + MT_W - 1 < cw - 1 ? tx0 + MT_W - 1 : cw - 1;
    const int ty1 = ty0 + MT_H - 1 < row_end - 1 ? ty0 + MT_H - 1 : row_end - 1;
    for (int e = 0; e < cnt; e++) {
        if (__builtin_amdgcn_ballot_w64(open != 0) == 0) break;          // every pixel of this wave has its sample
        const FrameDev& f = fr[list[e]];                 // uniform over the workgroup: scalar loads
        if (f.begX > tx1 || f.endX < tx0 || f.begY > ty1 || f.endY < ty0) continue;
        if (!open) continue;
        const float w1 = (float)(f.w - 1), h1 = (float)(f.h - 1);
#pragma unroll
        for (int j = 0; j < MT_RPL; j++) {
            const int yD = yB + 8 * j;
            const bool yin = yD >= f.begY && yD <= f.endY;
            const float yf = (float)yD - dGy;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int xD = xg + k;
                const bool want = ((open >> (4 * j + k)) & 1u) && yin && xD >= f.begX && xD <= f.endX;
                const float xf = (float)xD - dGx;
                float xs, ys;
                if (f.unit_den) { xs = f.inv[0] * xf + f.inv[1] * yf + f.inv[2]; ys = f.inv[3] * xf + f.inv[4] * yf + f.inv[5]; }
                else hm::apply_div9(f.inv, xf, yf, xs, ys);
                const bool ok = want && (xs >= 0.0f && xs < w1 && ys >= 0.0f && ys < h1);      // also rejects NaN
                if (!ok) continue;
                if constexpr (COVER) { used[list[e]] = 1; open &= ~(1u << (4 * j + k)); continue; }
                const int xi = (int)xs, yi = (int)ys;
                const float p = ys - (float)yi, q = xs - (float)xi;
                float b00, g00, r00, b01, g01, r01, b10, g10, r10, b11, g11, r11;
                // the 2 x 2 neighbourhood straight from the image: the lanes of a wave walk two nearly contiguous runs of the two
                // source rows, so the 6-byte loads share their cache lines (staging the tile's footprint in LDS first was measured
                // slower: 11.4 ms against 8.5 ms for the C3 canvas -- its barriers and 24 KB per workgroup cost more than the
                // L1 / L2 hits they replace)
                const uint8_t* g0 = f.src + (size_t)yi * f.ws + 3 * (size_t)xi;
                load_quad3(g0, f.ws, 3 * xi + 8 <= f.ws, b00, g00, r00, b01, g01, r01, b10, g10, r10, b11, g11, r11);
                const unsigned vb = hm::bilin(b00, b01, b10, b11, p, q), vg = hm::bilin(g00, g01, g10, g11, p, q), vr = hm::bilin(r00, r01, r10, r11, p, q);
                // bytes 3k, 3k+1, 3k+2 of the row's 12: static positions
                out[j][(3 * k) >> 2] |= vb << (8 * ((3 * k) & 3));
                out[j][(3 * k + 1) >> 2] |= vg << (8 * ((3 * k + 1) & 3));
                out[j][(3 * k + 2) >> 2] |= vr << (8 * ((3 * k + 2) & 3));
                open &= ~(1u << (4 * j + k));
            }
        }
    }
    if (COVER || xg >= cw) return;
#pragma unroll
    for (int j = 0; j < MT_RPL; j++) {
        const int yD = yB + 8 * j;
        if (yD >= row_end) continue;
        uint8_t* drow = canvas + (size_t)yD * cws + 3 * (size_t)xg;
        if (xg + 3 < cw) {
            uint32_t* d32 = reinterpret_cast<uint32_t*>(drow);
            d32[0] = out[j][0]; d32[1] = out[j][1]; d32[2] = out[j][2];
        } else {
#pragma unroll
            for (int b = 0; b < 9; b++)                  // at most 3 pixels; static indices keep out[] in registers
                if (xg + b / 3 < cw) drow[b] = (uint8_t)(out[j][b >> 2] >> (8 * (b & 3)));
        }
        // row padding (cvZero'd in the reference, :2248): the lane that owns the row's last pixel group clears [3 cw, cws), also when the
        // group is a full one and the caller chose a wider row than the layout's
        if (xg + 4 >= cw)
            for (int b = 3 * cw; b < cws; b++) canvas[(size_t)yD * cws + b] = 0;
    }
}

// cover_only != NULL: cover_only[k] = 1 for the frames this call would read (the stripe's cover list, mi355_mosaic_stripe_cover), nothing is
// rendered.  cover_exact == 0: every frame whose clipped canvas box meets the rows (host geometry alone: a superset); != 0: the frames that
// give at least one pixel its sample -- the tile kernel's walk without its loads (what the rendering pass really dereferences).
int mi_mosaic_refined_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n,
                          const float* h9s, uint8_t* d_canvas, int cw, int ch, int cws, int row0, int rows, uint8_t* cover_only, int cover_exact) {
    int lw, lh, lws; float dG[2];
    int rc = mi355_mosaic_layout(w, h, n, h9s, &lw, &lh, &lws, dG);
    if (rc != MI355_OK) { ctx->set_error("mosaic_refined: no image with h[8] != 0 / empty canvas"); return rc; }
    if (lw != cw || lh != ch || cws < cw * 3 || (cws & 3)) { ctx->set_error("mosaic_refined: canvas geometry does not match mi355_mosaic_layout"); return MI355_ERR_ARG; }
    if (row0 < 0) row0 = 0;
    if (rows < 0 || row0 + rows > ch) rows = ch - row0;
    if (rows <= 0) return MI355_OK;
    if (n > 65535) { ctx->set_error("mosaic_refined: at most 65535 images"); return MI355_ERR_ARG; }
    std::vector<FrameDev> fr;
    std::vector<int> frame_of, withheld;           // image index of fr[q]; images whose box meets the rows and that came without a pointer
    fr.reserve(n);
    for (int k = 0; k < n; k++) {                  // ascending image order = overwrite order (MosaicWithoutPos.cpp:2254)
        const float* m = h9s + 9 * k;
        if (m[8] == 0.0f) continue;
        FrameDev f;
        memset(&f, 0, sizeof(f));
        if (mi_inverse_matrix_host(m, 3, f.inv, 1e-12f) != 1) continue;   // MosaicWithoutPos.cpp:2275 (reference: garbage invH)
        float bminX = big(), bminY = big(), bmaxX = -big(), bmaxY = -big();
        {
            const float cx[4] = {0.0f, (float)(w[k] - 1), (float)(w[k] - 1), 0.0f};
            const float cy[4] = {0.0f, 0.0f, (float)(h[k] - 1), (float)(h[k] - 1)};
            for (int i = 0; i < 4; i++) {
                float X, Y;
                hm::apply_div9(m, cx[i], cy[i], X, Y);
                X = X + (0.0f + dG[0]); Y = Y + (0.0f + dG[1]);
                if (X < bminX) bminX = X;
                if (X > bmaxX) bmaxX = X;
                if (Y < bminY) bminY = Y;
                if (Y > bmaxY) bmaxY = Y;
            }
        }
        int begY = (int)(bminY - 0.5f), endY = (int)(bmaxY + 0.5f);       // :2305-2306
        int begX = (int)(bminX - 0.5f), endX = (int)(bmaxX + 0.5f);
        if (begX < 0) begX = 0;
        if (begY < 0) begY = 0;
        if (endX > cw - 1) endX = cw - 1;
        if (endY > ch - 1) endY = ch - 1;
        if (begY < row0) begY = row0;                                   // canvas stripe
        if (endY > row0 + rows - 1) endY = row0 + rows - 1;
        if (endX < begX || endY < begY) continue;
        if (cover_only && !cover_exact) { cover_only[k] = 1; continue; }
        if (w[k] < 2 || h[k] < 2 || ws[k] < 3 * w[k]) { ctx->set_error("mosaic_refined: bad image geometry"); return MI355_ERR_ARG; }
        // d_imgs[k] == NULL: the caller holds no copy of this image (owner-only frames, mi355_exchange_frames): it says the rows do not read
        // it -- which is true of a frame that lies under later frames wherever its box meets the rows (MI355_COVER_REFINED_EXACT lists what
        // IS read).  Such a frame is left out of the walk; with option "strict_frames" the statement is checked first (one cover pass).
        if (!cover_only && !d_imgs[k]) { withheld.push_back(k); continue; }
        f.src = cover_only ? nullptr : d_imgs[k]; f.w = w[k]; f.h = h[k]; f.ws = ws[k];
        frame_of.push_back(k);
        f.begX = begX; f.endX = endX; f.begY = begY; f.endY = endY;
        f.unit_den = (f.inv[6] == 0.0f && f.inv[7] == 0.0f && f.inv[8] == 1.0f) ? 1 : 0;
        fr.push_back(f);
    }
    if (cover_only && !cover_exact) return MI355_OK;
    if (!withheld.empty() && ctx->strict_frames) {       // the caller's statement "these rows do not read the images I withhold", checked: one cover pass
        std::vector<uint8_t> read((size_t)n, 0);
        const int rc2 = mi_mosaic_refined_dev(ctx, d_imgs, w, h, ws, n, h9s, nullptr, cw, ch, cws, row0, rows, read.data(), 1);
        if (rc2 != MI355_OK) return rc2;
        for (int k : withheld) if (read[k]) { ctx->set_error("mosaic_refined: these canvas rows read image " + std::to_string(k) + " but no pointer to it was given"); return MI355_ERR_ARG; }
    }
    const int nf = (int)fr.size();
    if (cover_only && nf == 0) return MI355_OK;
    const int bx_n = (cw + MT_COARSE - 1) / MT_COARSE, by_n = (rows + MT_COARSE - 1) / MT_COARSE;
    DevBuf& dfr = ctx->buf("mosaic_frames");
    DevBuf& dl = ctx->buf("mosaic_lists");
    DevBuf& dc = ctx->buf("mosaic_counts");
    MI_HIP(dfr.reserve(sizeof(FrameDev) * (size_t)(nf > 0 ? nf : 1)));
    MI_HIP(dl.reserve(sizeof(uint16_t) * (size_t)bx_n * by_n * (size_t)(nf > 0 ? nf : 1)));
    MI_HIP(dc.reserve(sizeof(int) * (size_t)bx_n * by_n));
    if (nf > 0) MI_HIP(hipMemcpyAsync(dfr.p, fr.data(), sizeof(FrameDev) * (size_t)nf, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mosaic_lists_kernel, dim3((bx_n * by_n + 255) / 256), dim3(256), 0, ctx->stream, dfr.as<FrameDev>(), nf, bx_n, by_n, row0, dl.as<uint16_t>(), dc.as<int>());
    if (cover_only) {
        DevBuf& du = ctx->buf("mosaic_used");
        MI_HIP(du.reserve(sizeof(int) * (size_t)nf));
        MI_HIP(hipMemsetAsync(du.p, 0, sizeof(int) * (size_t)nf, ctx->stream));
        hipLaunchKernelGGL(mosaic_tile_kernel<true>, dim3((cw + MT_W - 1) / MT_W, (rows + MT_H - 1) / MT_H), dim3(256), 0, ctx->stream,
                           dfr.as<FrameDev>(), nf, dl.as<uint16_t>(), dc.as<int>(), bx_n, (uint8_t*)nullptr, cw, cws, row0, row0 + rows, dG[0], dG[1], du.as<int>());
        MI_HIP(hipGetLastError());
        std::vector<int> used((size_t)nf);
        MI_HIP(hipMemcpyAsync(used.data(), du.p, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
        for (int q = 0; q < nf; q++) if (used[q]) cover_only[frame_of[q]] = 1;
        return MI355_OK;
    }
    {
        // SURVEY 8(d) algorithmic figure: every image read once and written once (6 P per image)
        double bytes = 0.0;
        for (const FrameDev& f : fr) bytes += 6.0 * (double)f.w * f.h;
        ProfScope ps(ctx, "warp", bytes);
        hipLaunchKernelGGL(mosaic_tile_kernel<false>, dim3((cw + MT_W - 1) / MT_W, (rows + MT_H - 1) / MT_H), dim3(256), 0, ctx->stream,
                           dfr.as<FrameDev>(), nf, dl.as<uint16_t>(), dc.as<int>(), bx_n, d_canvas, cw, cws, row0, row0 + rows, dG[0], dG[1], (int*)nullptr);
    }
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(ctx->stream));           // `fr` goes out of scope
    return MI355_OK;
}

int mi_warp_image(mi355_ctx* ctx, const uint8_t* src, int w, int h, int ws, int ch, const float* h9,
                  uint8_t** dst, int* dw, int* dh, int* dws) {
    if (!src || !h9 || !dst || w < 2 || h < 2 || (ch != 1 && ch != 3) || ws < w * ch) { ctx->set_error("warp_image: bad arguments"); return MI355_ERR_ARG; }
    float minX = big(), minY = big(), maxX = -big(), maxY = -big();
    corner_bbox(h9, w, h, minX, minY, maxX, maxY);
    const int nw = (int)(maxX - minX + 1.5f), nh = (int)(maxY - minY + 1.5f);       // MosaicImage.cpp:1653-1654
    if (nw <= 0 || nh <= 0) { ctx->set_error("warp_image: empty result"); return MI355_ERR_ARG; }     // CreateBitmap8U -> NULL
    const int nws = (nw * ch + 3) / 4 * 4;                                               // ImageIO.cpp:70
    WarpArgs a;
    memset(&a, 0, sizeof(a));
    if (mi_inverse_matrix_host(h9, 3, a.inv, 1e-6f) != 1) { ctx->set_error("warp_image: homography not invertible"); return MI355_ERR_SINGULAR; }
    DevBuf& dsrc = ctx->buf("warp_src");
    DevBuf& ddst = ctx->buf("warp_dst");
    const size_t src_bytes = (size_t)ws * h, dst_bytes = (size_t)nws * nh;
    MI_HIP(dsrc.reserve(src_bytes + 16));
    MI_HIP(ddst.reserve(dst_bytes));
    MI_HIP(hipMemcpyAsync(dsrc.p, src, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemsetAsync(ddst.p, 0, dst_bytes, ctx->stream));                            // ZeroImage, MosaicImage.cpp:1659
    a.src = dsrc.as<uint8_t>(); a.w = w; a.h = h; a.ws = ws;
    a.dst = ddst.as<uint8_t>(); a.dws = nws;
    a.x_beg = 0; a.x_end = nw - 1; a.y_beg = 0; a.y_end = nh - 1;
    a.dx = -minX; a.dy = -minY;
    if (ch == 3) launch_warp<3, false>(ctx, a); else launch_warp<1, false>(ctx, a);
    uint8_t* out = (uint8_t*)malloc(dst_bytes);
    if (!out) return MI355_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(out, ddst.p, dst_bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(out); ctx->set_error(hipGetErrorString(e)); return MI355_ERR_DEVICE; }
    *dst = out; *dw = nw; *dh = nh; *dws = nws;
    return MI355_OK;
}

// ---------------------------------------------------------------------------------------------------------
// chips + masks
// ---------------------------------------------------------------------------------------------------------
namespace {

struct ChipDev { int x0, y0, w, h, mws; size_t map_off; };

// ImageMath.cpp:88-103 LineOf2Points1
void line_of_2_points(float& a, float& b, float& c, float x1, float y1, float x2, float y2) {
    if (fabs((double)(x1 - x2)) < 0.000001) { a = 1.0f; b = 0.0f; c = -x1; }
    else { a = (y1 - y2) / (x1 - x2); b = -1.0f; c = y1 - a * x1; }
}

struct LineSet { float A[4], B[4], C[4], inv[4]; };

// distance of chip pixel (c, r) to the nearest of the 4 quad edges (MosaicImage.cpp:1790-1826); ONE expression for both kernels below:
// the ownership kernel recomputes what the maximum kernel saw, so the values must be the same bits
__device__ __forceinline__ float quad_min_dist(const LineSet& L, int c, int r) {
    float minDist = (float)(1 << 29);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float d = fabsf(L.A[i] * (float)c + L.B[i] * (float)r + L.C[i]) * L.inv[i];
        if (d < minDist) minDist = d;
    }
    return minDist;
}

// per chip: the chip-wide maximum of that distance over the valid pixels (float bits, all >= 0).  The distances themselves are NOT
// stored (round 2 kept a float map per chip pixel: 4 of the 8 bytes per chip pixel, 101 GB for the 2000 chips of C5)
// One launch for all chips (blockIdx.z = chip): a workgroup walks a 256 x 64 pixel block (4 pixels per lane from one 32-bit mask
// load, 16 row steps) and ends with ONE atomic -- a thread per pixel with an atomic per wave took 290 us per 12 MP chip, 25 % of a blend.
__global__ __launch_bounds__(256) void distmax_kernel(const ChipDev* chips, uint8_t* const* masks, const LineSet* lines, unsigned* maxbits) {
    const ChipDev cd = chips[blockIdx.z];
    const int c0 = (blockIdx.x * 64 + threadIdx.x) * 4, r0 = blockIdx.y * 64;
    if (blockIdx.x * 256 >= cd.w || r0 >= cd.h) return;
    const LineSet L = lines[blockIdx.z];
    const uint8_t* mask = masks[blockIdx.z];
    unsigned bits = 0;                                                  // all distances >= 0: float order = bit order (a NaN would win, as it did per pixel)
    if (c0 < cd.w) {
        for (int r = r0 + threadIdx.y; r < r0 + 64 && r < cd.h; r += 4) {
            const uint8_t* m = mask + (size_t)r * cd.mws + c0;          // mws is a multiple of 4 and c0 too: the 4 bytes are inside the row's storage
            const unsigned mm = *reinterpret_cast<const unsigned*>(m);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (c0 + k < cd.w && ((mm >> (8 * k)) & 0xffu) != 0) { const unsigned b = __float_as_uint(quad_min_dist(L, c0 + k, r)); bits = b > bits ? b : bits; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) { unsigned o = __shfl_xor(bits, off); bits = o > bits ? o : bits; }
    __shared__ unsigned s_m[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) s_m[tid >> 6] = bits;
    __syncthreads();
    if (tid == 0) {
        unsigned m = s_m[0]; m = s_m[1] > m ? s_m[1] : m; m = s_m[2] > m ? s_m[2] : m; m = s_m[3] > m ? s_m[3] : m;
        if (m > *reinterpret_cast<volatile unsigned*>(maxbits + blockIdx.z)) atomicMax(maxbits + blockIdx.z, m);
    }
}

// per canvas pixel: the chip with the strictly largest normalised distance (first wins, start 0) owns it (MosaicImage.cpp:1842-1872).
// The candidates of a pixel come from the chip list of its 256 x 256 canvas block (ascending chip index = the reference's order);
// validity is read from the masks (255 = sample exists), the distance is recomputed and divided by the chip's maximum (:1828), then
// every covering chip's mask byte is rewritten: 255 for the owner, 0 for the others.  A mask byte belongs to exactly one canvas
// pixel, so the thread of that pixel is the only one that touches it.
constexpr int OWN_BLK = 256;
constexpr int OWN_ROWS = 4;
// The chips of the block (uniform over the workgroup: 64 x 4 threads inside one block) are staged in LDS, 32 at a time: with the
// descriptors read per thread and per chip from global memory every candidate was a chain of four dependent loads (list -> chip ->
// mask pointer -> mask byte) and the kernel ran at 0.2 TB/s of its own byte traffic.
struct OwnEntry { int x0, y0, w, h, mws, k; float maxv; int pad; uint8_t* mask; LineSet L; };
__global__ __launch_bounds__(256) void owner_kernel(const ChipDev* chips, const LineSet* lines, const unsigned* maxbits, const int* list_off, const int* list,
                                                    int bx_n, int rectW, int rectH, uint8_t* const* masks, int row_beg) {
    constexpr int NE = 32;
    __shared__ OwnEntry s_e[NE];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    // row_beg (a multiple of the workgroup's 16 rows) .. rectH - 1: the canvas rows this launch decides (a stripe of the canvas, or all of it)
    const int r = row_beg + (blockIdx.y * blockDim.y + threadIdx.y) * OWN_ROWS;                        // first of the thread's rows
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const bool inside = c < rectW && r < rectH;
    const int rb = (row_beg + blockIdx.y * blockDim.y * OWN_ROWS) / OWN_BLK, cb = (blockIdx.x * blockDim.x) / OWN_BLK;      // the workgroup (64 x 16 pixels) lies inside one block
    const int blk = rb * bx_n + cb;
    const int l0 = list_off[blk], l1 = list_off[blk + 1];
    // One walk over the candidates (round 4; it was two: find the owner, then rewrite every covering chip's byte).  Afterwards the owner's
    // byte is 255 and every other covering chip's is 0.  A byte is 255 or 0 before (sample / no sample) and the owner's is 255 necessarily
    // (its distance is > 0, so it has a sample): it is enough to clear the byte of every candidate that has a sample and loses -- at once when
    // it does not beat the leader (d > bd is false: smaller, equal = the earlier chip wins, or NaN), or when a later chip takes the lead.
    // Four consecutive rows per thread (a wave = 64 columns x 4 rows, the workgroup 64 x 16): the candidate's descriptor (26 dwords of LDS),
    // the column test and the products A[i] * column are shared by the four pixels; each row keeps its own leader.
    float bd[OWN_ROWS];
    uint8_t* lead[OWN_ROWS];                                      // the leader's mask byte
#pragma unroll
    for (int j = 0; j < OWN_ROWS; j++) { bd[j] = 0.0f; lead[j] = nullptr; }
    for (int base = l0; base < l1; base += NE) {
        const int ne = l1 - base < NE ? l1 - base : NE;
        __syncthreads();
        if (tid < ne) {
            const int k = list[base + tid];
            const ChipDev cd = chips[k];
            OwnEntry e;
            e.x0 = cd.x0; e.y0 = cd.y0; e.w = cd.w; e.h = cd.h; e.mws = cd.mws; e.k = k; e.maxv = __uint_as_float(maxbits[k]); e.pad = 0;
            e.mask = masks[k]; e.L = lines[k];
            s_e[tid] = e;
        }
        __syncthreads();
        if (!inside) continue;
        // The mask bytes of candidate q + 1 are fetched before candidate q is judged: a candidate's stores go to its own mask or to an earlier
        // leader's, never to the next candidate's, but the compiler cannot know that and would start each load after the previous stores.
        uint8_t nxt[OWN_ROWS];
        auto fetch = [&](int q, uint8_t* out) {
            const OwnEntry& e = s_e[q];
            const int xC = c - e.x0, yC0 = r - e.y0;
            const bool xin = xC >= 0 && xC < e.w;
#pragma unroll
            for (int j = 0; j < OWN_ROWS; j++) {
                const int yC = yC0 + j;
                out[j] = (xin && yC >= 0 && yC < e.h && r + j < rectH) ? e.mask[(size_t)yC * e.mws + xC] : (uint8_t)0;
            }
        };
        fetch(0, nxt);
        for (int q = 0; q < ne; q++) {
            const OwnEntry& e = s_e[q];
            uint8_t cur[OWN_ROWS];
#pragma unroll
            for (int j = 0; j < OWN_ROWS; j++) cur[j] = nxt[j];
            if (q + 1 < ne) fetch(q + 1, nxt);
            const int xC = c - e.x0, yC0 = r - e.y0;
            if (xC < 0 || xC >= e.w || yC0 + OWN_ROWS - 1 < 0 || yC0 >= e.h) continue;
            float ac[4];
#pragma unroll
            for (int i = 0; i < 4; i++) ac[i] = e.L.A[i] * (float)xC;      // the first product of quad_min_dist's expression
#pragma unroll
            for (int j = 0; j < OWN_ROWS; j++) {
                const int yC = yC0 + j;
                if (yC < 0 || yC >= e.h || r + j >= rectH) continue;
                uint8_t* m = e.mask + (size_t)yC * e.mws + xC;
                if (cur[j] != 0) {
                    float minDist = (float)(1 << 29);                       // quad_min_dist(e.L, xC, yC), same operations
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const float dd = fabsf(ac[i] + e.L.B[i] * (float)yC + e.L.C[i]) * e.L.inv[i];
                        if (dd < minDist) minDist = dd;
                    }
                    const float d = minDist / e.maxv;
                    if (d > bd[j]) { bd[j] = d; if (lead[j]) *lead[j] = 0; lead[j] = m; }
                    else *m = 0;
                }
            }
        }
    }
}

// Per chip the box of its mask's non-zero bytes, i.e. of the pixels the chip OWNS after owner_kernel: {min column, min row} in
// bbox_min[2k..] (start 0x7f7f7f7f), {max column, max row} in bbox_max[2k..] (start -1).  The blender works only where a chip's weights can
// be non-zero (blend.hip chip_windows).  One launch for all chips, a workgroup per 256 x 64 mask block (one 32-bit load per lane and row);
// only the workgroups that meet a non-zero byte -- a chip owns a few percent of its area in a dense survey -- end with atomics.
// (Tracked inside owner_kernel instead, every wave of a cell moved the maximum row: 12 000 atomics per address, 7 -> 18 ms per canvas.)
__global__ __launch_bounds__(256) void mask_bbox_kernel(const ChipDev* chips, uint8_t* const* masks, int* bbox_min, int* bbox_max, int row_lo, int row_hi) {
    const ChipDev cd = chips[blockIdx.z];
    const int c0 = (blockIdx.x * 64 + threadIdx.x) * 4, r0 = blockIdx.y * 64;
    if (blockIdx.x * 256 >= cd.w || r0 >= cd.h) return;
    // only the canvas rows row_lo .. row_hi were decided by owner_kernel (a stripe; the whole canvas otherwise): the rest of the mask still
    // holds validity and is not part of the box
    if (cd.y0 + r0 + 63 < row_lo || cd.y0 + r0 > row_hi) return;
    const uint8_t* mask = masks[blockIdx.z];
    int xa = 0x7fffffff, xb = -1, ya = 0x7fffffff, yb = -1;
    if (c0 < cd.w) {
#pragma unroll 4
        for (int r = r0 + threadIdx.y; r < r0 + 64 && r < cd.h; r += 4) {
            if (cd.y0 + r < row_lo || cd.y0 + r > row_hi) continue;
            unsigned mm = *reinterpret_cast<const unsigned*>(mask + (size_t)r * cd.mws + c0);      // mws and c0 are multiples of 4
            if (c0 + 3 >= cd.w) mm &= 0xffffffffu >> (8 * (c0 + 4 - cd.w));                        // row padding is not part of the chip
            if (mm) {
                const int lo = c0 + (__builtin_ctz(mm) >> 3), hi = c0 + 3 - (__builtin_clz(mm) >> 3);
                xa = lo < xa ? lo : xa; xb = hi > xb ? hi : xb; ya = r < ya ? r : ya; yb = r > yb ? r : yb;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        int o;
        o = __shfl_xor(xa, off); xa = o < xa ? o : xa;
        o = __shfl_xor(ya, off); ya = o < ya ? o : ya;
        o = __shfl_xor(xb, off); xb = o > xb ? o : xb;
        o = __shfl_xor(yb, off); yb = o > yb ? o : yb;
    }
    __shared__ int s_b[4][4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if ((tid & 63) == 0) { s_b[tid >> 6][0] = xa; s_b[tid >> 6][1] = ya; s_b[tid >> 6][2] = xb; s_b[tid >> 6][3] = yb; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; i++) {
            xa = s_b[i][0] < xa ? s_b[i][0] : xa; ya = s_b[i][1] < ya ? s_b[i][1] : ya;
            xb = s_b[i][2] > xb ? s_b[i][2] : xb; yb = s_b[i][3] > yb ? s_b[i][3] : yb;
        }
        if (xb >= 0) {
            int* mn = bbox_min + 2 * blockIdx.z; int* mx = bbox_max + 2 * blockIdx.z;
            if (xa < *reinterpret_cast<volatile int*>(mn)) atomicMin(mn, xa);
            if (ya < *reinterpret_cast<volatile int*>(mn + 1)) atomicMin(mn + 1, ya);
            if (xb > *reinterpret_cast<volatile int*>(mx)) atomicMax(mx, xb);
            if (yb > *reinterpret_cast<volatile int*>(mx + 1)) atomicMax(mx + 1, yb);
        }
    }
}

}  // namespace

// Device stage of the chips: layout on the host, warps / distance maps / ownership on the device.  The chips and masks
// stay in ctx buffers "chip_imgs" / "chip_masks" at chip_off[v] / mask_off[v]; *chips_out is malloc'd.
// row_lo .. row_hi (canvas rows, inclusive; the default is the whole canvas): a STRIPE of the canvas -- only the chips that reach these rows
// (widened to whole groups of 16) get storage, validity masks, a maximum distance, and ownership is decided for these rows only; the owned
// boxes are the boxes inside the stripe, and the chips outside it report an empty box.  What is decided is what the whole canvas gives
// there: ownership is per canvas pixel among the chips that cover it, and a chip's maximum distance is taken over the whole chip.
int mi_chips_and_masks_dev(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                           const float* h9s, const uint8_t* keep, int find_masks, int* n_chips, mi355_chip_info** chips_out,
                           std::vector<size_t>& chip_off, std::vector<size_t>& mask_off, int* cw_out, int* ch_out, int imgs_on_device,
                           std::vector<int>* owned_bbox, int defer_pixels, int row_lo, int row_hi, uint8_t* cover_only) {
    if (!imgs || !w || !h || !ws || !h9s || n <= 0 || !n_chips || !chips_out) return MI355_ERR_ARG;
    // ---- layout, MosaicImage.cpp:2233-2343 (host, same float ops) ----
    float maxX = 0.0f, maxY = 0.0f, minX = 0.0f, minY = 0.0f;                     // :2234 (canvas always contains the origin)
    std::vector<float> bx0(n), by0(n), bx1(n), by1(n);
    std::vector<int> kept;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if ((keep && !keep[k]) || m[8] == 0.0f) continue;
        float bMinX = big(), bMinY = big(), bMaxX = -big(), bMaxY = -big();
        corner_bbox(m, w[k], h[k], bMinX, bMinY, bMaxX, bMaxY);
        if (bMaxX > maxX) maxX = bMaxX;
        if (bMinX < minX) minX = bMinX;
        if (bMaxY > maxY) maxY = bMaxY;
        if (bMinY < minY) minY = bMinY;
        bx0[k] = bMinX; by0[k] = bMinY; bx1[k] = bMaxX; by1[k] = bMaxY;
        kept.push_back(k);
    }
    const float dGx = -minX, dGy = -minY;
    const int newW = (int)(maxX - minX + 1.5f), newH = (int)(maxY - minY + 1.5f);
    const int nv = (int)kept.size();
    // the stripe in whole groups of owner_kernel's 16 rows
    const bool striped = row_lo > 0 || row_hi < newH - 1;
    if (striped && (!find_masks || !defer_pixels || !owned_bbox)) { ctx->set_error("chips: a row window needs the one-call blend's form"); return MI355_ERR_ARG; }
    const int row_beg = (row_lo < 0 ? 0 : row_lo) & ~(4 * OWN_ROWS - 1);
    int row_end = row_hi >= newH - 1 ? newH : ((row_hi + 4 * OWN_ROWS) & ~(4 * OWN_ROWS - 1));      // exclusive
    if (row_end > newH) row_end = newH;
    // released on every failure path below; handed to the caller only on success
    std::unique_ptr<mi355_chip_info, void (*)(void*)> ci_hold((mi355_chip_info*)calloc((size_t)(nv > 0 ? nv : 1), sizeof(mi355_chip_info)), free);
    mi355_chip_info* ci = ci_hold.get();
    if (!ci) return MI355_ERR_NOMEM;
    std::vector<ChipDev> cd(nv);
    std::vector<int> av;                                     // the chips that reach the stripe, ascending (= the reference's order)
    size_t map_total = 0, chip_total = 0, mask_total = 0;
    chip_off.assign(nv, 0); mask_off.assign(nv, 0);
    for (int v = 0; v < nv; v++) {
        const int k = kept[v];
        const float* m = h9s + 9 * k;
        const float bX = bx0[k] + dGx, bY = by0[k] + dGy, eX = bx1[k] + dGx, eY = by1[k] + dGy;     // :2314-2317
        const int begX = (int)bX, begY = (int)bY, endX = (int)(eX + 0.5f), endY = (int)(eY + 0.5f);
        const float sx = (float)begX - bX, sy = (float)begY - bY;
        mi355_chip_info& c = ci[v];
        c.x0 = begX; c.y0 = begY; c.w = endX - begX + 1; c.h = endY - begY + 1; c.img = k; c.sx = sx; c.sy = sy;
        const float ox[4] = {0.0f, (float)(w[k] - 1), (float)(w[k] - 1), 0.0f};
        const float oy[4] = {0.0f, 0.0f, (float)(h[k] - 1), (float)(h[k] - 1)};
        for (int i = 0; i < 4; i++) {
            float tx, ty;
            hm::apply_recip9(m, ox[i], oy[i], tx, ty);                                               // :2334
            c.quad[2 * i] = ((tx + dGx) + sx) - (float)begX;
            c.quad[2 * i + 1] = ((ty + dGy) + sy) - (float)begY;
        }
        if (c.w <= 0 || c.h <= 0) { ctx->set_error("chips: empty chip"); return MI355_ERR_FAILED; }
        const int cws = (c.w * 3 + 3) & ~3, mws = (c.w + 3) & ~3;
        cd[v] = ChipDev{c.x0, c.y0, c.w, c.h, mws, 0};
        if (striped && (c.y0 + c.h - 1 < row_beg || c.y0 >= row_end)) continue;                      // does not reach the stripe
        chip_off[v] = chip_total; mask_off[v] = mask_total;
        chip_total += (size_t)cws * c.h; mask_total += (size_t)mws * c.h;
        cd[v].map_off = map_total;
        map_total += (size_t)mws * c.h;
        av.push_back(v);
    }
    if (cover_only) {                                        // the images whose chips this call would form: nothing else is done
        for (int v : av) cover_only[kept[v]] = 1;
        *n_chips = nv; *chips_out = nullptr;
        if (cw_out) *cw_out = newW;
        if (ch_out) *ch_out = newH;
        return MI355_OK;
    }
    const int na = (int)av.size();
    DevBuf& dchips = ctx->buf("chip_imgs");
    DevBuf& dmasks = ctx->buf("chip_masks");
    DevBuf& dsrc = ctx->buf("warp_src");
    DevBuf& dmeta = ctx->buf("chip_meta");
    MI_HIP(dchips.reserve(chip_total + 16));
    MI_HIP(dmasks.reserve(mask_total + 16));
    // defer_pixels (the one-call blend): every mask byte of a chip's columns is written by the validity pass, the pixels later and only
    // inside the chip's active window (mi_chip_pixels_prepare / _launch); row padding is never read there, so nothing is cleared
    if (!defer_pixels) {
        MI_HIP(hipMemsetAsync(dchips.p, 0, chip_total, ctx->stream));
        MI_HIP(hipMemsetAsync(dmasks.p, 0, mask_total, ctx->stream));
    }
    if (defer_pixels) ctx->deferred_warps.assign(sizeof(WarpArgs) * (size_t)nv, 0);
    // every kept source is staged in HBM up front (frames stay resident: 288 GB), so uploads and warps of consecutive chips
    // overlap on the stream instead of synchronising per chip
    std::vector<size_t> src_off(nv, 0);
    size_t src_total = 0;
    for (int v : av) {
        const int k = kept[v];
        if (!imgs[k] || w[k] < 2 || h[k] < 2 || ws[k] < 3 * w[k]) { ctx->set_error("chips: bad image geometry"); return MI355_ERR_ARG; }
        src_off[v] = src_total; src_total += ((size_t)ws[k] * h[k] + 255) & ~(size_t)255;
    }
    if (!imgs_on_device) MI_HIP(dsrc.reserve(src_total + 16));
    std::vector<WarpArgs> wargs((size_t)(na > 0 ? na : 1));
    for (int q = 0; q < na; q++) {
        const int v = av[q], k = kept[v];
        const mi355_chip_info& c = ci[v];
        WarpArgs a;
        memset(&a, 0, sizeof(a));
        if (mi_inverse_matrix_host(h9s + 9 * k, 3, a.inv, 1e-12f) != 1) { ctx->set_error("chips: homography not invertible"); return MI355_ERR_SINGULAR; }  // :2348
        const size_t src_bytes = (size_t)ws[k] * h[k];
        if (!imgs_on_device) MI_HIP(hipMemcpyAsync(dsrc.as<uint8_t>() + src_off[v], imgs[k], src_bytes, hipMemcpyHostToDevice, ctx->stream));
        a.src = imgs_on_device ? imgs[k] : dsrc.as<uint8_t>() + src_off[v]; a.w = w[k]; a.h = h[k]; a.ws = ws[k];
        a.dst = dchips.as<uint8_t>() + chip_off[v]; a.dws = (c.w * 3 + 3) & ~3;
        a.mask = dmasks.as<uint8_t>() + mask_off[v]; a.mws = cd[v].mws;
        a.x_beg = 0; a.x_end = c.w - 1; a.y_beg = 0; a.y_end = c.h - 1;
        a.dx = dGx; a.dy = dGy; a.sx = c.sx; a.sy = c.sy; a.x0 = c.x0; a.y0 = c.y0;
        if (defer_pixels) { memcpy(ctx->deferred_warps.data() + sizeof(WarpArgs) * (size_t)v, &a, sizeof(a)); wargs[q] = a; }
        else launch_warp<3, true>(ctx, a);
    }
    if (defer_pixels && na > 0) {                       // validity masks of all chips: one launch
        DevBuf& dwa = ctx->buf("chip_warp_args");
        MI_HIP(dwa.reserve(sizeof(WarpArgs) * (size_t)na));
        MI_HIP(hipMemcpyAsync(dwa.p, wargs.data(), sizeof(WarpArgs) * (size_t)na, hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));      // `wargs` is a local: the copy must have read it before it goes
        int mw = 1, mh = 1;
        for (int v : av) { if (ci[v].w > mw) mw = ci[v].w; if (ci[v].h > mh) mh = ci[v].h; }
        ProfScope ps(ctx, "warp", (double)mask_total);
        for (int v0 = 0; v0 < na; v0 += 65535)
            hipLaunchKernelGGL((warp_chips_kernel<1>), dim3(((mw + 3) / 4 + 63) / 64, (mh + 3) / 4, na - v0 < 65535 ? na - v0 : 65535), dim3(64, 4), 0, ctx->stream,
                               dwa.as<WarpArgs>() + v0);
    }
    // validity masks are final here unless the distance-map ownership is requested
    if (find_masks && na > 0) {
        // chip lists per 256 x 256 canvas block (host: the chips' rectangles are known), ascending chip index inside a block; entries are
        // positions in `av`
        const int bx_n = (newW + OWN_BLK - 1) / OWN_BLK, by_n = (newH + OWN_BLK - 1) / OWN_BLK;
        std::vector<int> loff((size_t)bx_n * by_n + 1, 0);
        for (int q = 0; q < na; q++) {
            const int v = av[q];
            const int bx0 = std::max(0, ci[v].x0 / OWN_BLK), bx1 = std::min(bx_n - 1, (ci[v].x0 + ci[v].w - 1) / OWN_BLK);
            const int by0 = std::max(0, ci[v].y0 / OWN_BLK), by1 = std::min(by_n - 1, (ci[v].y0 + ci[v].h - 1) / OWN_BLK);
            for (int by = by0; by <= by1; by++) for (int bx = bx0; bx <= bx1; bx++) loff[(size_t)by * bx_n + bx + 1]++;
        }
        for (size_t q = 1; q < loff.size(); q++) loff[q] += loff[q - 1];
        std::vector<int> lst((size_t)loff.back() > 0 ? loff.back() : 1), fill(loff.begin(), loff.end() - 1);
        for (int q = 0; q < na; q++) {
            const int v = av[q];
            const int bx0 = std::max(0, ci[v].x0 / OWN_BLK), bx1 = std::min(bx_n - 1, (ci[v].x0 + ci[v].w - 1) / OWN_BLK);
            const int by0 = std::max(0, ci[v].y0 / OWN_BLK), by1 = std::min(by_n - 1, (ci[v].y0 + ci[v].h - 1) / OWN_BLK);
            for (int by = by0; by <= by1; by++) for (int bx = bx0; bx <= bx1; bx++) lst[fill[(size_t)by * bx_n + bx]++] = q;
        }
        std::vector<LineSet> lines(na);
        std::vector<ChipDev> cda(na);
        for (int q = 0; q < na; q++) {
            const int v = av[q];
            cda[q] = cd[v];
            const float* qd = ci[v].quad;
            LineSet& L = lines[q];
            line_of_2_points(L.A[0], L.B[0], L.C[0], qd[0], qd[1], qd[2], qd[3]);
            line_of_2_points(L.A[1], L.B[1], L.C[1], qd[2], qd[3], qd[4], qd[5]);
            line_of_2_points(L.A[2], L.B[2], L.C[2], qd[4], qd[5], qd[6], qd[7]);
            line_of_2_points(L.A[3], L.B[3], L.C[3], qd[6], qd[7], qd[0], qd[1]);
            for (int i = 0; i < 4; i++) L.inv[i] = 1.0f / sqrtf(L.A[i] * L.A[i] + L.B[i] * L.B[i]);       // :1786
        }
        auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t o_cd = 0, o_mp = up16(o_cd + sizeof(ChipDev) * na), o_mx = up16(o_mp + sizeof(uint8_t*) * na), o_ln = up16(o_mx + sizeof(unsigned) * na),
                     o_lo = up16(o_ln + sizeof(LineSet) * na), o_ls = up16(o_lo + sizeof(int) * loff.size()), o_bb = up16(o_ls + sizeof(int) * lst.size()),
                     meta_bytes = o_bb + sizeof(int) * 4 * na;
        MI_HIP(dmeta.reserve(meta_bytes + 64));
        uint8_t* mb = dmeta.as<uint8_t>();
        ChipDev* d_cd = reinterpret_cast<ChipDev*>(mb + o_cd);
        uint8_t** d_mptr = reinterpret_cast<uint8_t**>(mb + o_mp);
        unsigned* d_max = reinterpret_cast<unsigned*>(mb + o_mx);
        LineSet* d_lines = reinterpret_cast<LineSet*>(mb + o_ln);
        int* d_loff = reinterpret_cast<int*>(mb + o_lo);
        int* d_list = reinterpret_cast<int*>(mb + o_ls);
        // one host image of the whole meta area, one copy (six small copies from pageable memory cost ~0.35 ms each on the stream): chip
        // descriptors, mask pointers, the maxima (zero), the edge lines, the block lists, the owned boxes' start values
        std::vector<uint8_t> blob(meta_bytes, 0);
        memcpy(blob.data() + o_cd, cda.data(), sizeof(ChipDev) * na);
        for (int q = 0; q < na; q++) { uint8_t* mp = dmasks.as<uint8_t>() + mask_off[av[q]]; memcpy(blob.data() + o_mp + sizeof(uint8_t*) * q, &mp, sizeof(mp)); }
        memcpy(blob.data() + o_ln, lines.data(), sizeof(LineSet) * na);
        memcpy(blob.data() + o_lo, loff.data(), sizeof(int) * loff.size());
        memcpy(blob.data() + o_ls, lst.data(), sizeof(int) * lst.size());
        memset(blob.data() + o_bb, 0x7f, sizeof(int) * 2 * na);
        memset(blob.data() + o_bb + sizeof(int) * 2 * na, 0xff, sizeof(int) * 2 * na);
        MI_HIP(hipMemcpyAsync(mb, blob.data(), meta_bytes, hipMemcpyHostToDevice, ctx->stream));
        int* d_bbmin = owned_bbox ? reinterpret_cast<int*>(mb + o_bb) : nullptr;      // [2 na] minima, then [2 na] maxima
        int* d_bbmax = owned_bbox ? d_bbmin + 2 * na : nullptr;
        dim3 block(64, 4);
        int mw = 1, mh = 1;
        for (int v : av) { if (ci[v].w > mw) mw = ci[v].w; if (ci[v].h > mh) mh = ci[v].h; }
        {
            ProfScope ps(ctx, "distmap", (double)chip_total / 3.0);
            for (int v0 = 0; v0 < na; v0 += 65535)      // gridDim.z limit
                hipLaunchKernelGGL(distmax_kernel, dim3((mw + 255) / 256, (mh + 63) / 64, na - v0 < 65535 ? na - v0 : 65535), block, 0, ctx->stream,
                                   d_cd + v0, d_mptr + v0, d_lines + v0, d_max + v0);
        }
        dim3 grid((newW + 63) / 64, (row_end - row_beg + 4 * OWN_ROWS - 1) / (4 * OWN_ROWS));
        {
            ProfScope ps(ctx, "owner", (double)newW * (row_end - row_beg));
            hipLaunchKernelGGL(owner_kernel, grid, block, 0, ctx->stream, d_cd, d_lines, d_max, d_loff, d_list, bx_n, newW, row_end, d_mptr, row_beg);
        }
        if (owned_bbox) {
            ProfScope ps(ctx, "distmap", (double)chip_total / 3.0);
            for (int v0 = 0; v0 < na; v0 += 65535)
                hipLaunchKernelGGL(mask_bbox_kernel, dim3((mw + 255) / 256, (mh + 63) / 64, na - v0 < 65535 ? na - v0 : 65535), block, 0, ctx->stream,
                                   d_cd + v0, d_mptr + v0, d_bbmin + 2 * v0, d_bbmax + 2 * v0, row_beg, row_end - 1);
        }
        std::vector<int> bb;
        if (owned_bbox) { bb.resize((size_t)4 * na); MI_HIP(hipMemcpyAsync(bb.data(), d_bbmin, sizeof(int) * 4 * na, hipMemcpyDeviceToHost, ctx->stream)); }
        MI_HIP(hipStreamSynchronize(ctx->stream));        // the host vectors above were sources of asynchronous copies
        if (owned_bbox) {
            owned_bbox->assign((size_t)4 * nv, 0);
            for (int v = 0; v < nv; v++) { int* o = owned_bbox->data() + 4 * v; o[0] = 0; o[1] = 0; o[2] = -1; o[3] = -1; }      // chips outside the stripe: nothing owned
            for (int q = 0; q < na; q++) { int* o = owned_bbox->data() + 4 * av[q]; o[0] = bb[2 * q]; o[1] = bb[2 * q + 1]; o[2] = bb[2 * na + 2 * q]; o[3] = bb[2 * na + 2 * q + 1]; }
        }
    } else if (owned_bbox && find_masks) {
        owned_bbox->assign((size_t)4 * nv, 0);
        for (int v = 0; v < nv; v++) { int* o = owned_bbox->data() + 4 * v; o[2] = -1; o[3] = -1; }
    }
    MI_HIP(hipGetLastError());
    *n_chips = nv; *chips_out = ci_hold.release();
    if (cw_out) *cw_out = newW;
    if (ch_out) *ch_out = newH;
    return MI355_OK;
}

// deferred chips: entry e of the list = chip chips[e] restricted to columns / rows win[4e .. 4e+3] (inclusive, clipped to the chip).  The
// arguments of all entries go to the device in one copy; mi_chip_pixels_launch then renders a run of entries with one launch.
int mi_chip_pixels_prepare(mi355_ctx* ctx, int n, const int* chips, const int* win) {
    std::vector<WarpArgs> arr((size_t)(n > 0 ? n : 1));
    ctx->deferred_dims.assign((size_t)2 * (n > 0 ? n : 0), 0);
    for (int e = 0; e < n; e++) {
        const int chip = chips[e];
        if (chip < 0 || sizeof(WarpArgs) * ((size_t)chip + 1) > ctx->deferred_warps.size()) { ctx->set_error("chip_pixels: no such deferred chip"); return MI355_ERR_ARG; }
        WarpArgs a;
        memcpy(&a, ctx->deferred_warps.data() + sizeof(WarpArgs) * (size_t)chip, sizeof(a));
        const int* w4 = win + 4 * e;
        a.x_beg = w4[0] > a.x_beg ? w4[0] : a.x_beg; a.y_beg = w4[1] > a.y_beg ? w4[1] : a.y_beg;
        a.x_end = w4[2] < a.x_end ? w4[2] : a.x_end; a.y_end = w4[3] < a.y_end ? w4[3] : a.y_end;
        arr[e] = a;
        ctx->deferred_dims[2 * e] = a.x_end < a.x_beg ? 0 : (a.x_end - (a.x_beg & ~3)) / 4 + 1;      // groups of 4 columns
        ctx->deferred_dims[2 * e + 1] = a.y_end < a.y_beg ? 0 : a.y_end - a.y_beg + 1;
    }
    if (n <= 0) return MI355_OK;
    DevBuf& d = ctx->buf("chip_warp_windows");
    MI_HIP(d.reserve(sizeof(WarpArgs) * (size_t)n));
    MI_HIP(hipMemcpy(d.p, arr.data(), sizeof(WarpArgs) * (size_t)n, hipMemcpyHostToDevice));      // `arr` is a local: the copy must have read it on return
    return MI355_OK;
}
int mi_chip_pixels_launch(mi355_ctx* ctx, int first, int count) {
    if (count <= 0) return MI355_OK;
    if (first < 0 || (size_t)2 * (first + count) > ctx->deferred_dims.size()) { ctx->set_error("chip_pixels: bad range"); return MI355_ERR_ARG; }
    int gw = 0, gh = 0;
    for (int e = first; e < first + count; e++) { gw = ctx->deferred_dims[2 * e] > gw ? ctx->deferred_dims[2 * e] : gw; gh = ctx->deferred_dims[2 * e + 1] > gh ? ctx->deferred_dims[2 * e + 1] : gh; }
    if (gw == 0 || gh == 0) return MI355_OK;
    ProfScope ps(ctx, "warp", 0.0);
    hipLaunchKernelGGL((warp_chips_kernel<2>), dim3((gw + 63) / 64, (gh + 3) / 4, count), dim3(64, 4), 0, ctx->stream, ctx->buf("chip_warp_windows").as<WarpArgs>() + first);
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

// canvas size of the chips' layout alone (MosaicImage.cpp:2233-2292): the same float operations as the head of mi_chips_and_masks_dev
int mi_blend_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int* cw, int* ch) {
    if (!w || !h || !h9s || n <= 0 || !cw || !ch) return MI355_ERR_ARG;
    float maxX = 0.0f, maxY = 0.0f, minX = 0.0f, minY = 0.0f;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if ((keep && !keep[k]) || m[8] == 0.0f) continue;
        float bMinX = big(), bMinY = big(), bMaxX = -big(), bMaxY = -big();
        corner_bbox(m, w[k], h[k], bMinX, bMinY, bMaxX, bMaxY);
        if (bMaxX > maxX) maxX = bMaxX;
        if (bMinX < minX) minX = bMinX;
        if (bMaxY > maxY) maxY = bMaxY;
        if (bMinY < minY) minY = bMinY;
    }
    *cw = (int)(maxX - minX + 1.5f); *ch = (int)(maxY - minY + 1.5f);
    return MI355_OK;
}

int mi_chips_and_masks(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                       const float* h9s, const uint8_t* keep, int find_masks, int* n_chips, mi355_chip_info** chips_out,
                       uint8_t*** chip_imgs, uint8_t*** masks_out, int* cw_out, int* ch_out) {
    if (!chip_imgs || !masks_out) return MI355_ERR_ARG;
    std::vector<size_t> chip_off, mask_off;
    int nv = 0;
    mi355_chip_info* ci = nullptr;
    int rc = mi_chips_and_masks_dev(ctx, imgs, w, h, ws, n, h9s, keep, find_masks, &nv, &ci, chip_off, mask_off, cw_out, ch_out);
    if (rc != MI355_OK) return rc;                     // nothing was handed out
    // host results: owned here until the last copy has landed, then handed to the caller
    struct HostArrays {
        mi355_chip_info* ci; uint8_t** a = nullptr; uint8_t** b = nullptr; int n;
        ~HostArrays() {
            if (a) for (int v = 0; v < n; v++) free(a[v]);
            if (b) for (int v = 0; v < n; v++) free(b[v]);
            free(a); free(b); free(ci);
        }
    } hold{ci, nullptr, nullptr, nv};
    hold.a = (uint8_t**)calloc((size_t)(nv > 0 ? nv : 1), sizeof(uint8_t*));
    hold.b = (uint8_t**)calloc((size_t)(nv > 0 ? nv : 1), sizeof(uint8_t*));
    if (!hold.a || !hold.b) return MI355_ERR_NOMEM;
    DevBuf& dchips = ctx->buf("chip_imgs");
    DevBuf& dmasks = ctx->buf("chip_masks");
    for (int v = 0; v < nv; v++) {
        const size_t cb = (size_t)((ci[v].w * 3 + 3) & ~3) * ci[v].h, mb = (size_t)((ci[v].w + 3) & ~3) * ci[v].h;
        hold.a[v] = (uint8_t*)malloc(cb);
        hold.b[v] = (uint8_t*)malloc(mb);
        if (!hold.a[v] || !hold.b[v]) return MI355_ERR_NOMEM;
        MI_HIP(hipMemcpyAsync(hold.a[v], dchips.as<uint8_t>() + chip_off[v], cb, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipMemcpyAsync(hold.b[v], dmasks.as<uint8_t>() + mask_off[v], mb, hipMemcpyDeviceToHost, ctx->stream));
    }
    MI_HIP(hipStreamSynchronize(ctx->stream));
    *n_chips = nv; *chips_out = hold.ci; *chip_imgs = hold.a; *masks_out = hold.b;
    hold.ci = nullptr; hold.a = nullptr; hold.b = nullptr;
    return MI355_OK;
}
