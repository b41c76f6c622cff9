/* oracle/oracle_blend.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Multiband blend of the warped chips: SURVEY 8(f) row f3, replacing
 *   detail::MultiBandBlender blender(false, band);  blender.prepare(Rect(0,0,W,H));
 *   blender.feed(chip converted to CV_16S, mask, corner) per chip;  blender.blend(result_s, result_mask);
 *   result_s.convertTo(result, CV_8U)                       (MosaicImage.cpp:2296-2299, 2451-2486)
 *
 * PARITY UNPINNED AT THE OUTPUT LEVEL: the arithmetic is OpenCV 2.4.0 `stitching` (blenders.cpp) + `imgproc` (pyramids.cpp), vendored
 * in the reference as headers + Win32 binaries only, and the reference commits no blended output to compare with.  The STRUCTURE
 * below was checked against the reference's own binary (Release/opencv_stitching240.dll, llvm-objdump):
 *   MultiBandBlender::feed (1000b190): gap = 3 << num_bands (1000b3c0-1000b3cb), corners snapped by >> / << num_bands (1000b486-1000b48c),
 *     copyMakeBorder(img, .., BORDER_REFLECT = 2) (1000b61d-1000b657), createLaplacePyr (1000b6c0), weight = mask.convertTo(CV_32F, 1/255)
 *     (1000b776-1000b782; the CV_16S weight branch exists and is not taken: the reference constructs MultiBandBlender(false, band)),
 *     copyMakeBorder of the weight with a constant border, pyrDown(.., BORDER_DEFAULT = 4) per level (1000b9bb), accumulation through
 *     sign-extended 16-bit loads and truncating float -> int conversions;
 *   createLaplacePyr (1000a440): pyrDown chain, pyrUp to the finer size, cv::subtract(.., dtype CV_16S = 3) (1000a59f-1000a797);
 *   normalizeUsingWeightMap (10006710): (short)(value / (weight + 1e-5f)) with cvttss2si = truncation toward zero (1000684d-100068ce).
 * The per-pixel arithmetic of cv::pyrDown / cv::pyrUp is restated from Burt & Adelson's REDUCE / EXPAND in the shape OpenCV gives it and
 * was then checked against the reference's opencv_imgproc240.dll (round 3, llvm-objdump; the template instantiations are identified by
 * their load / store widths):
 *   pyrDown, 16-bit outputs (two instantiations, short and ushort, vertical pass at 100f793c-100f7969 and 100f818c-100f81b9):
 *     (6 r2 + 4 (r1 + r3) + r0 + r4 + 128) >> 8 on int rows, stored with a plain 16-bit move (the weights sum to 256: saturation
 *     cannot trigger); the 8-bit instantiation at 100f70f0-100f7121 has the same shape;
 *   pyrDown, float (the weight pyramid): scalar SSE single precision (mulss / addss, no x87 extended intermediates) -- but the binary
 *     was compiled with reassociating float optimisation, and the association of the sum DIFFERS BY LOOP SLOT.  Horizontal pass, with
 *     a4 = (s[2x-1] + s[2x+1]) * 4, c6 = s[2x] * 6: the 4x unrolled interior loop (100f88d0-100f897f) computes ((a4 + c6) + s[2x+2]) + s[2x-2]
 *     in three slots and ((a4 + s[2x+2]) + c6) + s[2x-2] in the fourth, its remainder loop (100f89a0-100f89cf) ((a4 + c6) + s[2x-2]) + s[2x+2],
 *     the border loop over the index table (100f8822-100f888a) ((a4 + s[2x-2]) + c6) + s[2x+2].  Vertical pass: an SSE routine
 *     (call at 100f8e51 -> 100f6460, OpenCV's PyrDownVec_32f) for the bulk of a row and a scalar tail (100f8eaf-100f8ef6) with the
 *     1/256 distributed into the constants: (r0 + r4) * (1/256) + (r1 + r3) * (4/256) + r2 * (6/256).
 *     THE ORACLE DOES NOT FOLLOW THESE SLOT-DEPENDENT ORDERS: it keeps the single association of OpenCV's published source
 *     (c6 + a4 + s[2x-2] + s[2x+2], the same over the rows, times 1/256).  The weights are the same real numbers; their float values
 *     can differ from the reference's in the last bit at some pixels, which can move a (short)(laplacian * weight) truncation by one.
 *     The 16-bit Laplacian paths are integer arithmetic and do not depend on association.
 *   pyrUp, short: horizontal pass at 100fa54b-100fa5b3: left border 6 s0 + 2 s1 | 4 (s0 + s1), right border s[w-2] + 7 s[w-1] | 8 s[w-1];
 *     vertical pass at 100fa6dc-100fa708: (r0 + 6 r1 + r2 + 32) >> 6 and (4 (r1 + r2) + 32) >> 6, 16-bit moves.
 * So the choices below are the binary's:
 *   REDUCE  (pyrDown) i16: v = sum over the 5x5 taps (rows then columns, int), out = (v + 128) >> 8; reflect-101 border
 *           f32: row = s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2] (left to right), same vertically, times 1/256
 *   EXPAND  (pyrUp) i16 to exactly twice the size: horizontally even = s[x-1] + 6 s[x] + s[x+1], odd = 4 (s[x] + s[x+1]),
 *           first pair 6 s0 + 2 s1 | 4 (s0 + s1), last pair s[w-2] + 7 s[w-1] | 8 s[w-1]; vertically the same
 *           even / odd combination of rows with row(-1) := row(1) and row(h) := row(h-1); out = (v + 32) >> 6
 *   Laplacian level = Gaussian level - EXPAND(next level), saturated to i16; the last level is the Gaussian itself
 *   feed:   region of interest = chip rectangle grown by 3*2^bands, clipped to the padded canvas, snapped to multiples
 *           of 2^bands; chip extended into it by reflection including the edge pixel (fedcba|abcdefgh|hgfedcb);
 *           weight = mask * (1/255) as float, extended by zeros, REDUCEd per level;
 *           canvas Laplacian += (short)(chip Laplacian * weight) (truncation toward zero, 16-bit wrap-around add),
 *           canvas weight += weight
 *   blend:  canvas Laplacian = (short)(value / (weight + 1e-5f)); collapse by level = sat16(EXPAND(level+1) + level);
 *           pixels whose level-0 weight is <= 1e-5 become 0; result = clamp to 0..255
 * bands = min(band, ceil(log2(max(W, H)))); the canvas is padded to multiples of 2^bands, so every level halves exactly.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

static inline int reflect101i(int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; } return p; }
static inline int reflecti(int p, int n) { while (p < 0 || p >= n) { if (p < 0) p = -p - 1; else p = 2 * n - 1 - p; } return p; }
static inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* REDUCE, 3-channel i16: src w x h -> dst (w/2) x (h/2), w and h even */
void orc_pyr_down16(const int16_t* src, int w, int h, int16_t* dst)
{
    const int dw = w / 2, dh = h / 2;
    int* rows = (int*)malloc(sizeof(int) * 5 * dw * 3);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101i(2 * y - 2 + k, h);
            const int16_t* s = src + (size_t)sy * w * 3;
            int* r = rows + (size_t)k * dw * 3;
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < 3; c++) {
                    const int x0 = reflect101i(2 * x - 2, w), x1 = reflect101i(2 * x - 1, w), x2 = 2 * x, x3 = reflect101i(2 * x + 1, w), x4 = reflect101i(2 * x + 2, w);
                    r[3 * x + c] = s[3 * x2 + c] * 6 + (s[3 * x1 + c] + s[3 * x3 + c]) * 4 + s[3 * x0 + c] + s[3 * x4 + c];
                }
        }
        for (int i = 0; i < dw * 3; i++) {
            const int v = rows[2 * dw * 3 + i] * 6 + (rows[1 * dw * 3 + i] + rows[3 * dw * 3 + i]) * 4 + rows[i] + rows[4 * dw * 3 + i];
            dst[(size_t)y * dw * 3 + i] = sat16((v + 128) >> 8);
        }
    }
    free(rows);
}

/* REDUCE, 1-channel f32 */
/* REDUCE, 1-channel f32 (the weight pyramid): cv::pyrDown<float> of opencv_imgproc240.dll, association by association.
 * The binary was compiled with reassociating float optimisation and the order of the four additions of a row value depends on the
 * LOOP that produces it (a4 = (s[2x-1] + s[2x+1]) * 4, c6 = s[2x] * 6):
 *   x = 0 and x >= width0 (border loop over the index table, 100f8822-100f888a)   ((a4 + s[2x-2]) + c6) + s[2x+2]
 *   1 <= x < 1 + 4 n4, n4 = (width0 - 1) / 4 (4 x unrolled loop, 100f88d0-100f897f)
 *         (x - 1) mod 4 in {0, 1, 2}                                                ((a4 + c6) + s[2x+2]) + s[2x-2]
 *         (x - 1) mod 4 == 3                                                        ((a4 + s[2x+2]) + c6) + s[2x-2]
 *   1 + 4 n4 <= x < width0 (remainder loop, 100f89a0-100f89cf)                      ((a4 + c6) + s[2x-2]) + s[2x+2]
 * with width0 = min((w - 3) / 2 + 1, dw) (= dw - 1 for even w).  Vertical pass over the five row values r0 .. r4:
 *   x < 8 (dw / 8) (PyrDownVec_32f, 100f6540-100f65c6)   (((r1 + r3) + r2) * 4 + ((r0 + r4) + (r2 + r2))) * (1 / 256)
 *   the other columns (scalar tail, 100f8eaf-100f8ef6)    ((r0 + r4) * (1 / 256) + (r1 + r3) * (4 / 256)) + r2 * (6 / 256)
 * orc_float_reduce_mode = 0 (the default, and what csrc/blend.hip computes) keeps the single association of the published source
 * instead (c6 + a4 + s[2x-2] + s[2x+2], the same over the rows, times 1 / 256).  The binary's orders are implemented here to MEASURE
 * what the difference does to a blended mosaic (tests/test_blend.py: weights of levels 1-3 are exact multiples of 2^-24 and do not
 * depend on the order at all; levels 4-5 differ in the last bit near mask seams).  They are not the product's definition because they
 * tie a pixel's weight to its column's position inside the chip's region modulo 4 and 8: a window of the canvas could then no longer
 * be checked on a crop (the check tests/test_gpu_full_size.py runs at the C5 size), for an effect of a few output bytes by one level. */
int orc_float_reduce_mode = 0;
void orc_set_float_reduce_mode(int binary_order) { orc_float_reduce_mode = binary_order; }

void orc_pyr_down_f(const float* src, int w, int h, float* dst)
{
    const int dw = w / 2, dh = h / 2;
    float* rows = (float*)malloc(sizeof(float) * 5 * dw);
    int width0 = (w - 3) / 2 + 1;
    if (width0 > dw) width0 = dw;
    const int n4 = width0 - 1 >= 4 ? (width0 - 1) / 4 : 0;
    const int vec_end = dw >= 8 ? (dw / 8) * 8 : 0;
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101i(2 * y - 2 + k, h);
            const float* s = src + (size_t)sy * w;
            float* r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                const int x0 = reflect101i(2 * x - 2, w), x1 = reflect101i(2 * x - 1, w), x2 = 2 * x, x3 = reflect101i(2 * x + 1, w), x4 = reflect101i(2 * x + 2, w);
                if (!orc_float_reduce_mode) { r[x] = s[x2] * 6.0f + (s[x1] + s[x3]) * 4.0f + s[x0] + s[x4]; continue; }
                const float a4 = (s[x1] + s[x3]) * 4.0f, c6 = s[x2] * 6.0f, m2 = s[x0], p2 = s[x4];
                float v;
                if (x == 0 || x >= width0) { v = a4 + m2; v = v + c6; v = v + p2; }
                else if (x < 1 + 4 * n4) {
                    if (((x - 1) & 3) == 3) { v = a4 + p2; v = v + c6; v = v + m2; }
                    else { v = a4 + c6; v = v + p2; v = v + m2; }
                } else { v = a4 + c6; v = v + m2; v = v + p2; }
                r[x] = v;
            }
        }
        const float *r0 = rows, *r1 = rows + dw, *r2 = rows + 2 * dw, *r3 = rows + 3 * dw, *r4 = rows + 4 * dw;
        for (int x = 0; x < dw; x++) {
            float v;
            if (!orc_float_reduce_mode) { v = r2[x] * 6.0f + (r1[x] + r3[x]) * 4.0f + r0[x] + r4[x]; v = v * (1.0f / 256.0f); }
            else if (x < vec_end) {
                float a = r1[x] + r3[x]; a = a + r2[x]; a = a * 4.0f;
                float b = r0[x] + r4[x]; const float c = r2[x] + r2[x]; b = b + c;
                v = a + b; v = v * (1.0f / 256.0f);
            } else {
                float a = r0[x] + r4[x]; a = a * (1.0f / 256.0f);
                float b = r1[x] + r3[x]; b = b * (4.0f / 256.0f);
                const float c = r2[x] * (6.0f / 256.0f);
                v = a + b; v = v + c;
            }
            dst[(size_t)y * dw + x] = v;
        }
    }
    free(rows);
}

/* horizontal EXPAND of one source row (3 channels) into 2w int values per channel */
static void up_row(const int16_t* s, int w, int* r)
{
    for (int c = 0; c < 3; c++) {
        if (w == 1) { r[c] = s[c] * 8; r[3 + c] = s[c] * 8; continue; }
        r[c] = s[c] * 6 + s[3 + c] * 2;
        r[3 + c] = (s[c] + s[3 + c]) * 4;
        for (int x = 1; x < w - 1; x++) {
            r[3 * (2 * x) + c] = s[3 * (x - 1) + c] + s[3 * x + c] * 6 + s[3 * (x + 1) + c];
            r[3 * (2 * x + 1) + c] = (s[3 * x + c] + s[3 * (x + 1) + c]) * 4;
        }
        r[3 * (2 * (w - 1)) + c] = s[3 * (w - 2) + c] + s[3 * (w - 1) + c] * 7;
        r[3 * (2 * (w - 1) + 1) + c] = s[3 * (w - 1) + c] * 8;
    }
}

/* EXPAND, 3-channel i16: src w x h -> dst 2w x 2h */
void orc_pyr_up16(const int16_t* src, int w, int h, int16_t* dst)
{
    const int dw = 2 * w;
    int* r0 = (int*)malloc(sizeof(int) * dw * 3);
    int* r1 = (int*)malloc(sizeof(int) * dw * 3);
    int* r2 = (int*)malloc(sizeof(int) * dw * 3);
    for (int y = 0; y < h; y++) {
        const int ym = (y == 0) ? (h > 1 ? 1 : 0) : y - 1;           /* row(-1) := row(1) */
        const int yp = (y == h - 1) ? h - 1 : y + 1;                 /* row(h)  := row(h-1) */
        up_row(src + (size_t)ym * w * 3, w, r0);
        up_row(src + (size_t)y * w * 3, w, r1);
        up_row(src + (size_t)yp * w * 3, w, r2);
        for (int i = 0; i < dw * 3; i++) {
            dst[(size_t)(2 * y) * dw * 3 + i] = sat16((r0[i] + r1[i] * 6 + r2[i] + 32) >> 6);
            dst[(size_t)(2 * y + 1) * dw * 3 + i] = sat16(((r1[i] + r2[i]) * 4 + 32) >> 6);
        }
    }
    free(r0); free(r1); free(r2);
}

/* chips: BGR u8, row stride (w*3+3)&~3; masks: u8, row stride (w+3)&~3; x0/y0/w/h per chip; canvas W x H.
 * out: BGR u8, row stride (W*3+3)&~3 (caller-allocated).  Returns the number of bands used. */
int orc_multiband_blend(const uint8_t* const* chips, const uint8_t* const* masks, const int* cx0, const int* cy0, const int* cw, const int* chh,
                        int n, int W, int H, int band, uint8_t* out)
{
    int maxlen = W > H ? W : H;
    int nb = (int)ceil(log((double)maxlen) / log(2.0));
    if (nb > band) nb = band;
    if (nb < 0) nb = 0;
    const int al = 1 << nb;
    const int Wp = (W + al - 1) / al * al, Hp = (H + al - 1) / al * al;
    int16_t** dl = (int16_t**)calloc((size_t)nb + 1, sizeof(int16_t*));
    float** dwt = (float**)calloc((size_t)nb + 1, sizeof(float*));
    for (int l = 0; l <= nb; l++) {
        dl[l] = (int16_t*)calloc((size_t)(Wp >> l) * (Hp >> l) * 3, sizeof(int16_t));
        dwt[l] = (float*)calloc((size_t)(Wp >> l) * (Hp >> l), sizeof(float));
    }
    for (int k = 0; k < n; k++) {
        const int gap = 3 * al;
        int tlx = cx0[k] - gap > 0 ? cx0[k] - gap : 0, tly = cy0[k] - gap > 0 ? cy0[k] - gap : 0;
        int brx = cx0[k] + cw[k] + gap < Wp ? cx0[k] + cw[k] + gap : Wp, bry = cy0[k] + chh[k] + gap < Hp ? cy0[k] + chh[k] + gap : Hp;
        tlx = (tlx >> nb) << nb; tly = (tly >> nb) << nb;
        int rw = brx - tlx, rh = bry - tly;
        rw += ((1 << nb) - rw % (1 << nb)) % (1 << nb);
        rh += ((1 << nb) - rh % (1 << nb)) % (1 << nb);
        brx = tlx + rw; bry = tly + rh;
        int dx = brx - Wp > 0 ? brx - Wp : 0, dy = bry - Hp > 0 ? bry - Hp : 0;
        tlx -= dx; brx -= dx; tly -= dy; bry -= dy;
        const int left = cx0[k] - tlx, top = cy0[k] - tly;
        const int cws = (cw[k] * 3 + 3) & ~3, mws = (cw[k] + 3) & ~3;
        /* level-0 Gaussian (reflect border) and weight (zero border) of the region */
        int16_t** g = (int16_t**)calloc((size_t)nb + 1, sizeof(int16_t*));
        float** wp = (float**)calloc((size_t)nb + 1, sizeof(float*));
        g[0] = (int16_t*)malloc(sizeof(int16_t) * (size_t)rw * rh * 3);
        wp[0] = (float*)calloc((size_t)rw * rh, sizeof(float));
        for (int y = 0; y < rh; y++) {
            const int sy = reflecti(y - top, chh[k]);
            for (int x = 0; x < rw; x++) {
                const int sx = reflecti(x - left, cw[k]);
                for (int c = 0; c < 3; c++) g[0][((size_t)y * rw + x) * 3 + c] = (int16_t)chips[k][(size_t)sy * cws + 3 * sx + c];
                if (y - top >= 0 && y - top < chh[k] && x - left >= 0 && x - left < cw[k])
                    wp[0][(size_t)y * rw + x] = (float)masks[k][(size_t)(y - top) * mws + (x - left)] * (float)(1.0 / 255.0);
            }
        }
        for (int l = 0; l < nb; l++) {
            g[l + 1] = (int16_t*)malloc(sizeof(int16_t) * (size_t)(rw >> (l + 1)) * (rh >> (l + 1)) * 3);
            orc_pyr_down16(g[l], rw >> l, rh >> l, g[l + 1]);
            wp[l + 1] = (float*)malloc(sizeof(float) * (size_t)(rw >> (l + 1)) * (rh >> (l + 1)));
            orc_pyr_down_f(wp[l], rw >> l, rh >> l, wp[l + 1]);
        }
        for (int l = 0; l < nb; l++) {                               /* Gaussian -> Laplacian in place */
            const size_t cnt = (size_t)(rw >> l) * (rh >> l) * 3;
            int16_t* up = (int16_t*)malloc(sizeof(int16_t) * cnt);
            orc_pyr_up16(g[l + 1], rw >> (l + 1), rh >> (l + 1), up);
            for (size_t i = 0; i < cnt; i++) g[l][i] = sat16((int)g[l][i] - (int)up[i]);
            free(up);
        }
        for (int l = 0; l <= nb; l++) {
            const int lw = rw >> l, lh = rh >> l, ox = tlx >> l, oy = tly >> l, DW = Wp >> l;
            for (int y = 0; y < lh; y++)
                for (int x = 0; x < lw; x++) {
                    const float wgt = wp[l][(size_t)y * lw + x];
                    const size_t di = (size_t)(oy + y) * DW + (ox + x);
                    for (int c = 0; c < 3; c++) {
                        const int16_t add = (int16_t)((float)g[l][((size_t)y * lw + x) * 3 + c] * wgt);
                        dl[l][di * 3 + c] = (int16_t)(dl[l][di * 3 + c] + add);
                    }
                    dwt[l][di] += wgt;
                }
            free(g[l]); free(wp[l]);
        }
        free(g); free(wp);
    }
    for (int l = 0; l <= nb; l++) {
        const size_t cnt = (size_t)(Wp >> l) * (Hp >> l);
        for (size_t i = 0; i < cnt; i++)
            for (int c = 0; c < 3; c++) dl[l][i * 3 + c] = (int16_t)((float)dl[l][i * 3 + c] / (dwt[l][i] + 1e-5f));
    }
    for (int l = nb - 1; l >= 0; l--) {
        const size_t cnt = (size_t)(Wp >> l) * (Hp >> l) * 3;
        int16_t* up = (int16_t*)malloc(sizeof(int16_t) * cnt);
        orc_pyr_up16(dl[l + 1], Wp >> (l + 1), Hp >> (l + 1), up);
        for (size_t i = 0; i < cnt; i++) dl[l][i] = sat16((int)up[i] + (int)dl[l][i]);
        free(up);
    }
    const int ows = (W * 3 + 3) & ~3;
    memset(out, 0, (size_t)ows * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t di = (size_t)y * Wp + x;
            if (!(dwt[0][di] > 1e-5f)) continue;
            for (int c = 0; c < 3; c++) { const int v = dl[0][di * 3 + c]; out[(size_t)y * ows + 3 * x + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
        }
    for (int l = 0; l <= nb; l++) { free(dl[l]); free(dwt[l]); }
    free(dl); free(dwt);
    return nb;
}
