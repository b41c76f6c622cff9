/* oracle/oracle_surf.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).   PARITY UNPINNED.
 *
 * CPU restatement of the SURF detect+describe step of the reference's SURF variant of the path
 * (GetMatchedPairsOneToAllSurf, MosaicWithoutPos.cpp:5300-5533; all its call sites are commented out):
 *     SurfFeatureDetector detector(minHessian);   detector.detect(img, kp);          (:5313, :5333; minHessian = 50, MosaicWithoutPos.h:71)
 *     SurfDescriptorExtractor extractor;          extractor.compute(img, kp, desc);  (:5314, :5335)
 * cv::SURF's arithmetic lives in OpenCV 2.4.0's nonfree module (headers + Win32 binaries only in the reference tree): it can
 * neither be compiled nor run here and the reference holds no test or golden vector at this boundary.  This file restates the
 * PUBLISHED algorithm (H. Bay, A. Ess, T. Tuytelaars, L. Van Gool, "Speeded-Up Robust Features (SURF)", CVIU 2008) in the
 * structure and with the parameters of the cv::SURF class the reference instantiates (declaration: nonfree/features2d.hpp:107-145 of
 * the vendored headers: SURF(hessianThreshold, nOctaves = 4, nOctaveLayers = 2, extended = true, upright = false)):
 *   1. gray = (1868 B + 9617 G + 4899 R + 8192) >> 14 (8-bit BGR2GRAY fixed point, as in oracle_sift.c); integral image S of
 *      (h+1) x (w+1) 32-bit sums, wrapping modulo 2^32 (box sums are differences and stay exact)
 *   2. fast Hessian: per octave o (0..3) the box-filter sizes (9 + 6 l) << o for l = 0..3 sampled every 1 << o pixels;
 *      Dxx / Dyy (three boxes, weights 1 -2 1) and Dxy (four boxes, 1 -1 -1 1) from the 9 x 9 prototype stretched to the filter
 *      size, every box weighted by 1 / area; det = Dxx Dyy - 0.81 Dxy^2, trace = Dxx + Dyy
 *   3. maxima of det over the 26 neighbours in (x, y, layer) for the two middle layers of every octave, strictly greater than all
 *      and than hessianThreshold; quadratic interpolation in (x, y, size) (3 x 3 solve), accepted when every offset is within one
 *      sample; size = round(size + ds * offset); class_id = sign of the trace (the Laplacian)
 *   4. orientation: Haar responses (size 4 s, s = size * 1.2 / 9) at the 113 points of a radius-6 disc spaced s apart, Gaussian
 *      weighted (sigma 2.5), their angles rounded to degrees; the 60-degree window (72 positions, 5 degrees apart) with the
 *      largest summed response vector gives the direction
 *   5. descriptor: a (21 s)-wide window sampled along the direction (nearest pixel), reduced to 21 x 21 by area averaging, 20 x 20
 *      Haar differences weighted by a Gaussian (sigma 3.3), 4 x 4 cells of 5 x 5 samples, per cell the extended 8 sums
 *      (dx and |dx| split by the sign of dy, dy and |dy| split by the sign of dx) = 128 floats, normalised to unit length
 *   6. keypoints leave ordered by (response descending, octave, layer, row, column); the strongest max_kp are kept
 *
 * READ FROM THE REFERENCE'S BINARY (Release/opencv_nonfree240.dll, llvm-objdump; a spot check, not a pin): the constants 0.81
 * (1001cb7f, calcLayerDetAndTrace), 1.2 / 9 (100208d5), the Gaussian sigmas 2.5 and 3.3 as doubles (100228e9, 10022aae) are the ones
 * used below.  The same function shows what a bit-level pin would have to follow and this file does NOT: the build evaluates floating
 * point on the x87 unit -- det = (float)(dx * dy - (dxy * dxy) * 0.81) with dx, dy rounded to float and dxy kept at the double
 * precision calcHaarPattern accumulated it in (1001cb55-1001cb87), trace = (float)(dy + dx) -- where this restatement rounds every
 * operation to float.  Determinants therefore differ from the reference's in their last bits, which can reorder keypoints of nearly
 * equal response; SURF stays PARITY UNPINNED.
 *
 * DEFINED HERE (where the publication leaves freedom) so that the HIP implementation can be compared bit for bit:
 *   - box responses accumulate (double)boxsum * (double)weight over the boxes in prototype order, then round to float once;
 *   - rounding to integers is round-half-to-even (rint); atan2 / sin / cos are the fixed polynomials of oracle_sift.c;
 *   - the orientation window sums run over the samples in disc order (row-major over the disc), in float;
 *   - the 21 x 21 patch is the exact area average of the window over each cell, fractional border pixels weighted by their
 *     coverage, accumulated in float row-major over the covered pixels, rounded to the nearest integer 0..255;
 *   - cell sums accumulate in float row-major over the 5 x 5 samples; the squared magnitude accumulates in double over the 128
 *     values in index order; scale = (float)(1 / (sqrt(sum) + DBL_EPSILON)).
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SURF_OCTAVES 4
#define SURF_LAYERS 2                      /* nOctaveLayers: middle layers per octave; 4 filter sizes per octave */
#define SURF_HAAR_SIZE0 9
#define SURF_HAAR_SIZE_INC 6
#define ORI_RADIUS 6
#define ORI_WIN 60
#define ORI_SEARCH_INC 5
#define PATCH_SZ 20

/* ---- fixed transcendental approximations: the definitions of oracle_sift.c ------------------------------------------------ */
static inline float surf_atan2deg(float y, float x)
{
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + 2.220446e-16f); c2 = c * c; a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c; }
    else { c = ax / (ay + 2.220446e-16f); c2 = c * c; a = 90.0f - fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c; }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}
static inline void surf_sincosdeg(float deg, float* sn, float* cs)
{
    float q = rintf(deg * (1.0f / 90.0f));
    float r = fmaf(-90.0f, q, deg);
    float t = r * 0.017453292519943295f;
    float t2 = t * t;
    float sp = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, t2, 8.3333333e-3f);
    sp = fmaf(sp, t2, -1.6666667e-1f);
    float s = fmaf(sp * t2, t, t);
    float cp = fmaf(t2, 2.4801587e-5f, -1.3888889e-3f);
    cp = fmaf(cp, t2, 4.1666667e-2f);
    cp = fmaf(cp, t2, -0.5f);
    float c = fmaf(cp, t2, 1.0f);
    int k = ((int)q) & 3;
    if (k == 0) { *sn = s; *cs = c; }
    else if (k == 1) { *sn = c; *cs = -s; }
    else if (k == 2) { *sn = -s; *cs = -c; }
    else { *sn = -c; *cs = s; }
}

/* cv::getGaussianKernel(n, sigma, CV_32F) rounding (see oracle_sift.c gauss_kernel) */
static void gauss_taps(int n, double sigma, float* k)
{
    double sum = 0.0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { double x = (double)i - (double)(n - 1) * 0.5; k[i] = (float)exp(s2 * x * x); sum += (double)k[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) k[i] = (float)((double)k[i] * sum);
}

/* ---- box filters on the integral image -------------------------------------------------------------------------------------- */
typedef struct { int x1, y1, x2, y2; float w; } surf_box;          /* box [x1,x2) x [y1,y2) relative to the sample origin */

/* the 9 x 9 (or 4 x 4) prototype {x1, y1, x2, y2, weight} stretched to `size`: corners rint(ratio * c), weight / area */
static void stretch(const int proto[][5], int n, int old_size, int size, surf_box* out)
{
    const float ratio = (float)size / (float)old_size;
    for (int k = 0; k < n; k++) {
        out[k].x1 = (int)rintf(ratio * (float)proto[k][0]); out[k].y1 = (int)rintf(ratio * (float)proto[k][1]);
        out[k].x2 = (int)rintf(ratio * (float)proto[k][2]); out[k].y2 = (int)rintf(ratio * (float)proto[k][3]);
        out[k].w = (float)proto[k][4] / ((float)(out[k].x2 - out[k].x1) * (float)(out[k].y2 - out[k].y1));
    }
}
/* orc_surf_set_mode(1): det / trace as the reference's binary forms them on the x87 unit (see the header: dx, dy rounded to float, dxy
 * kept at the double precision it was accumulated in, the whole expression in double, rounded to float once) -- a measuring instrument
 * for tests/test_surf.py, not the definition the product implements (mode 0: every operation rounded to float) */
static int g_surf_mode = 0;
void orc_surf_set_mode(int mode) { g_surf_mode = mode; }
static inline double haar_d(const uint32_t* S, int sw, int x, int y, const surf_box* f, int n)
{
    double d = 0.0;
    for (int k = 0; k < n; k++) {
        const uint32_t a = S[(size_t)(y + f[k].y1) * sw + x + f[k].x1], b = S[(size_t)(y + f[k].y1) * sw + x + f[k].x2];
        const uint32_t c = S[(size_t)(y + f[k].y2) * sw + x + f[k].x1], e = S[(size_t)(y + f[k].y2) * sw + x + f[k].x2];
        const int32_t box = (int32_t)(a + e - b - c);
        d += (double)box * (double)f[k].w;
    }
    return d;
}
static inline float haar(const uint32_t* S, int sw, int x, int y, const surf_box* f, int n)
{
    double d = 0.0;
    for (int k = 0; k < n; k++) {
        const uint32_t a = S[(size_t)(y + f[k].y1) * sw + x + f[k].x1], b = S[(size_t)(y + f[k].y1) * sw + x + f[k].x2];
        const uint32_t c = S[(size_t)(y + f[k].y2) * sw + x + f[k].x1], e = S[(size_t)(y + f[k].y2) * sw + x + f[k].x2];
        const int32_t box = (int32_t)(a + e - b - c);              /* modulo 2^32: exact while the box sum is below 2^31 */
        d += (double)box * (double)f[k].w;
    }
    return (float)d;
}
static const int DX_P[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
static const int DY_P[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
static const int DXY_P[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};
static const int OX_P[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};
static const int OY_P[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};

typedef struct { int size, step, rows, cols; float* det; float* trace; } surf_layer;

typedef struct { uint32_t resp_bits; int octave, layer, i, j; float x, y, size, response; int lap; } surf_cand;
static int surf_cmp(const void* a, const void* b)
{
    const surf_cand* x = (const surf_cand*)a; const surf_cand* y = (const surf_cand*)b;
    if (x->resp_bits != y->resp_bits) return x->resp_bits > y->resp_bits ? -1 : 1;
    if (x->octave != y->octave) return x->octave < y->octave ? -1 : 1;
    if (x->layer != y->layer) return x->layer < y->layer ? -1 : 1;
    if (x->i != y->i) return x->i < y->i ? -1 : 1;
    if (x->j != y->j) return x->j < y->j ? -1 : 1;
    return 0;
}

/* x = A^-1 b, Gaussian elimination with partial pivoting (first largest pivot); singular -> 0 (oracle_sift.c solve3) */
static void surf_solve3(float A[3][3], float b[3], float x[3])
{
    int p[3] = {0, 1, 2};
    for (int k = 0; k < 3; k++) {
        int m = k; float best = fabsf(A[p[k]][k]);
        for (int r = k + 1; r < 3; r++) { float v = fabsf(A[p[r]][k]); if (v > best) { best = v; m = r; } }
        if (!(best > 1e-30f)) { x[0] = x[1] = x[2] = 0.0f; return; }
        int t = p[k]; p[k] = p[m]; p[m] = t;
        for (int r = k + 1; r < 3; r++) {
            float f = A[p[r]][k] / A[p[k]][k];
            for (int c = k + 1; c < 3; c++) A[p[r]][c] = A[p[r]][c] - f * A[p[k]][c];
            b[p[r]] = b[p[r]] - f * b[p[k]];
        }
    }
    x[2] = b[p[2]] / A[p[2]][2];
    x[1] = (b[p[1]] - A[p[1]][2] * x[2]) / A[p[1]][1];
    x[0] = ((b[p[0]] - A[p[0]][1] * x[1]) - A[p[0]][2] * x[2]) / A[p[0]][0];
}

/* orientation + descriptor of one keypoint.  Returns 0 when the keypoint has no orientation sample (dropped). */
static int surf_describe(const uint8_t* gray, int w, int h, const uint32_t* S, float kx, float ky, float ksize,
                         const float* aptw, const int* aptx, const int* apty, int napt, const float* DW, float* angle_out, float* desc)
{
    const int sw = w + 1;
    const float s = ksize * 1.2f / 9.0f;
    const int gws = 2 * (int)rintf(2.0f * s);                     /* Haar wavelet size of the orientation samples */
    if (h + 1 < gws || w + 1 < gws) return 0;
    surf_box ox[2], oy[2];
    stretch(OX_P, 2, 4, gws, ox); stretch(OY_P, 2, 4, gws, oy);
    float X[128], Y[128], ang[128];
    int na = 0;
    for (int k = 0; k < napt; k++) {
        const int x = (int)rintf(kx + (float)aptx[k] * s - (float)(gws - 1) / 2.0f);
        const int y = (int)rintf(ky + (float)apty[k] * s - (float)(gws - 1) / 2.0f);
        if (y < 0 || y >= (h + 1) - gws || x < 0 || x >= (w + 1) - gws) continue;
        const float vx = haar(S, sw, x, y, ox, 2), vy = haar(S, sw, x, y, oy, 2);
        X[na] = vx * aptw[k]; Y[na] = vy * aptw[k];
        ang[na] = surf_atan2deg(Y[na], X[na]);
        na++;
    }
    if (na == 0) return 0;
    float bestx = 0.0f, besty = 0.0f, best_mod = 0.0f;
    for (int i = 0; i < 360; i += ORI_SEARCH_INC) {
        float sx = 0.0f, sy = 0.0f;
        for (int j = 0; j < na; j++) {
            int d = abs((int)rintf(ang[j]) - i);
            if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) { sx = sx + X[j]; sy = sy + Y[j]; }
        }
        const float mod = sx * sx + sy * sy;
        if (mod > best_mod) { best_mod = mod; bestx = sx; besty = sy; }
    }
    const float dir = surf_atan2deg(-besty, bestx);
    *angle_out = dir;
    /* ---- the rotated window, reduced to (PATCH_SZ+1)^2 by area averaging ---- */
    const int win = (int)((float)(PATCH_SZ + 1) * s);
    float sn, cs;
    surf_sincosdeg(dir, &sn, &cs);
    const float sin_dir = -sn, cos_dir = cs;
    const float off = -(float)(win - 1) / 2.0f;
    const float start_x = kx + off * cos_dir + off * sin_dir, start_y = ky - off * sin_dir + off * cos_dir;
    float patch[PATCH_SZ + 1][PATCH_SZ + 1];
    if (win < 1) return 0;
    const float cell = (float)win / (float)(PATCH_SZ + 1);        /* window pixels per patch cell (any positive value) */
    for (int pi = 0; pi <= PATCH_SZ; pi++)
        for (int pj = 0; pj <= PATCH_SZ; pj++) {
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
                    /* window pixel (i, j): start + i * (sin, cos) + j * (cos, -sin), nearest image pixel, clamped */
                    const float px = (start_x + (float)i * sin_dir) + (float)j * cos_dir;
                    const float py = (start_y + (float)i * cos_dir) - (float)j * sin_dir;
                    int xi = (int)rintf(px), yi = (int)rintf(py);
                    xi = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi);
                    yi = yi < 0 ? 0 : (yi > h - 1 ? h - 1 : yi);
                    acc = fmaf((float)gray[(size_t)yi * w + xi], wgt, acc);
                    wsum = wsum + wgt;
                }
            }
            float v = rintf(acc / wsum);
            patch[pi][pj] = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        }
    float DXv[PATCH_SZ][PATCH_SZ], DYv[PATCH_SZ][PATCH_SZ];
    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++) {
            const float dw = DW[i * PATCH_SZ + j];
            DXv[i][j] = (((patch[i][j + 1] - patch[i][j]) + patch[i + 1][j + 1]) - patch[i + 1][j]) * dw;
            DYv[i][j] = (((patch[i + 1][j] - patch[i][j]) + patch[i + 1][j + 1]) - patch[i][j + 1]) * dw;
        }
    double sq = 0.0;
    for (int ci = 0; ci < 4; ci++)
        for (int cj = 0; cj < 4; cj++) {
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int y = ci * 5; y < ci * 5 + 5; y++)
                for (int x = cj * 5; x < cj * 5 + 5; x++) {
                    const float tx = DXv[y][x], ty = DYv[y][x];
                    if (ty >= 0) { v[0] = v[0] + tx; v[1] = v[1] + fabsf(tx); } else { v[2] = v[2] + tx; v[3] = v[3] + fabsf(tx); }
                    if (tx >= 0) { v[4] = v[4] + ty; v[5] = v[5] + fabsf(ty); } else { v[6] = v[6] + ty; v[7] = v[7] + fabsf(ty); }
                }
            for (int q = 0; q < 8; q++) desc[(ci * 4 + cj) * 8 + q] = v[q];
        }
    for (int q = 0; q < 128; q++) sq += (double)desc[q] * (double)desc[q];
    const float scale = (float)(1.0 / (sqrt(sq) + DBL_EPSILON));
    for (int q = 0; q < 128; q++) desc[q] = desc[q] * scale;
    return 1;
}

/* kp_out: orc_keypoint (x, y, size, angle, response, octave, class_id = sign of the Laplacian); desc_out: n x 128 floats */
int orc_surf(const uint8_t* bgr, int w, int h, int ws, float hessian_threshold, orc_keypoint* kp_out, float* desc_out, int max_kp)
{
    if (w < 16 || h < 16 || max_kp <= 0) return 0;
    const int sw = w + 1, sh = h + 1;
    uint8_t* gray = (uint8_t*)malloc((size_t)w * h);
    uint32_t* S = (uint32_t*)calloc((size_t)sw * sh, sizeof(uint32_t));
    for (int y = 0; y < h; y++) {
        uint32_t row = 0;
        for (int x = 0; x < w; x++) {
            const uint8_t* p = bgr + (size_t)y * ws + 3 * x;
            const int g = (1868 * (int)p[0] + 9617 * (int)p[1] + 4899 * (int)p[2] + 8192) >> 14;
            gray[(size_t)y * w + x] = (uint8_t)g;
            row += (uint32_t)g;
            S[(size_t)(y + 1) * sw + x + 1] = S[(size_t)y * sw + x + 1] + row;
        }
    }
    /* ---- det / trace of the 16 layers ---- */
    surf_layer L[SURF_OCTAVES * (SURF_LAYERS + 2)];
    for (int o = 0; o < SURF_OCTAVES; o++)
        for (int l = 0; l < SURF_LAYERS + 2; l++) {
            surf_layer* q = &L[o * (SURF_LAYERS + 2) + l];
            q->size = (SURF_HAAR_SIZE0 + SURF_HAAR_SIZE_INC * l) << o; q->step = 1 << o;
            q->rows = (sh - 1) / q->step; q->cols = (sw - 1) / q->step;
            q->det = (float*)calloc((size_t)q->rows * q->cols + 1, sizeof(float));
            q->trace = (float*)calloc((size_t)q->rows * q->cols + 1, sizeof(float));
            if (q->size > sh - 1 || q->size > sw - 1) continue;
            surf_box dx[3], dy[3], dxy[4];
            stretch(DX_P, 3, 9, q->size, dx); stretch(DY_P, 3, 9, q->size, dy); stretch(DXY_P, 4, 9, q->size, dxy);
            const surf_box* dxy_boxes = dxy;
            const int si = 1 + (sh - 1 - q->size) / q->step, sj = 1 + (sw - 1 - q->size) / q->step;
            const int margin = (q->size / 2) / q->step;
            for (int i = 0; i < si; i++)
                for (int j = 0; j < sj; j++) {
                    const float vx = haar(S, sw, j * q->step, i * q->step, dx, 3);
                    const float vy = haar(S, sw, j * q->step, i * q->step, dy, 3);
                    const float vxy = haar(S, sw, j * q->step, i * q->step, dxy, 4);
                    const size_t idx = (size_t)(i + margin) * q->cols + (j + margin);
                    q->det[idx] = vx * vy - (0.81f * vxy) * vxy;
                    q->trace[idx] = vx + vy;
                    if (g_surf_mode == 1) {
                        const double dxy = haar_d(S, sw, j * q->step, i * q->step, dxy_boxes, 4);
                        q->det[idx] = (float)((double)vx * (double)vy - (dxy * dxy) * 0.81);
                        q->trace[idx] = (float)((double)vy + (double)vx);
                    }
                }
        }
    /* ---- maxima over (x, y, layer) + interpolation ---- */
    size_t cap = 1 << 16, nc = 0;
    surf_cand* cand = (surf_cand*)malloc(cap * sizeof(surf_cand));
    for (int o = 0; o < SURF_OCTAVES; o++)
        for (int l = 1; l <= SURF_LAYERS; l++) {
            const surf_layer* a = &L[o * (SURF_LAYERS + 2) + l - 1];
            const surf_layer* b = a + 1;
            const surf_layer* c = a + 2;
            const int step = b->step, size = b->size;
            const int margin = (c->size / 2) / step + 1;
            for (int i = margin; i < b->rows - margin; i++)
                for (int j = margin; j < b->cols - margin; j++) {
                    const float v0 = b->det[(size_t)i * b->cols + j];
                    if (!(v0 > hessian_threshold)) continue;
                    float N9[3][9];
                    const surf_layer* lay[3] = {a, b, c};
                    int is_max = 1;
                    for (int q = 0; q < 3 && is_max; q++)
                        for (int di = -1; di <= 1; di++)
                            for (int dj = -1; dj <= 1; dj++) {
                                const float v = lay[q]->det[(size_t)(i + di) * b->cols + (j + dj)];
                                N9[q][(di + 1) * 3 + (dj + 1)] = v;
                                if (!(q == 1 && di == 0 && dj == 0) && !(v0 > v)) is_max = 0;
                            }
                    if (!is_max) continue;
                    /* fill what the early exit skipped */
                    for (int q = 0; q < 3; q++)
                        for (int di = -1; di <= 1; di++)
                            for (int dj = -1; dj <= 1; dj++) N9[q][(di + 1) * 3 + (dj + 1)] = lay[q]->det[(size_t)(i + di) * b->cols + (j + dj)];
                    const int sum_i = step * (i - (size / 2) / step), sum_j = step * (j - (size / 2) / step);
                    float cx = (float)sum_j + (float)(size - 1) * 0.5f, cy = (float)sum_i + (float)(size - 1) * 0.5f;
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
                    surf_solve3(A, bb, xx);
                    const int ok = (xx[0] != 0.0f || xx[1] != 0.0f || xx[2] != 0.0f) && fabsf(xx[0]) <= 1.0f && fabsf(xx[1]) <= 1.0f && fabsf(xx[2]) <= 1.0f;
                    if (!ok) continue;
                    cx = cx + xx[0] * (float)step; cy = cy + xx[1] * (float)step;
                    const float ksz = rintf((float)size + xx[2] * (float)(size - a->size));
                    if (nc == cap) { cap *= 2; cand = (surf_cand*)realloc(cand, cap * sizeof(surf_cand)); }
                    surf_cand* k = &cand[nc++];
                    union { float f; uint32_t u; } rb; rb.f = v0;
                    k->resp_bits = rb.u; k->octave = o; k->layer = l; k->i = i; k->j = j; k->x = cx; k->y = cy; k->size = ksz; k->response = v0;
                    const float tr = b->trace[(size_t)i * b->cols + j];
                    k->lap = tr > 0.0f ? 1 : (tr < 0.0f ? -1 : 0);
                }
        }
    qsort(cand, nc, sizeof(surf_cand), surf_cmp);
    /* ---- orientation + descriptors of the strongest candidates ---- */
    float G[2 * ORI_RADIUS + 1];
    gauss_taps(2 * ORI_RADIUS + 1, 2.5, G);
    int aptx[128], apty[128], napt = 0; float aptw[128];
    for (int i = -ORI_RADIUS; i <= ORI_RADIUS; i++)
        for (int j = -ORI_RADIUS; j <= ORI_RADIUS; j++)
            if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) { aptx[napt] = j; apty[napt] = i; aptw[napt] = G[i + ORI_RADIUS] * G[j + ORI_RADIUS]; napt++; }
    float g20[PATCH_SZ], DW[PATCH_SZ * PATCH_SZ];
    gauss_taps(PATCH_SZ, 3.3, g20);
    for (int i = 0; i < PATCH_SZ; i++) for (int j = 0; j < PATCH_SZ; j++) DW[i * PATCH_SZ + j] = g20[i] * g20[j];
    /* the strongest max_kp candidates are described; one without any orientation sample is dropped from the output */
    int n = 0;
    const size_t lim = nc < (size_t)max_kp ? nc : (size_t)max_kp;
    for (size_t q = 0; q < lim; q++) {
        const surf_cand* k = &cand[q];
        float ang = 0.0f;
        if (!surf_describe(gray, w, h, S, k->x, k->y, k->size, aptw, aptx, apty, napt, DW, &ang, desc_out + (size_t)n * 128)) continue;
        kp_out[n].x = k->x; kp_out[n].y = k->y; kp_out[n].size = k->size; kp_out[n].angle = ang; kp_out[n].response = k->response;
        kp_out[n].octave = k->octave; kp_out[n].class_id = k->lap;
        n++;
    }
    for (int q = 0; q < SURF_OCTAVES * (SURF_LAYERS + 2); q++) { free(L[q].det); free(L[q].trace); }
    free(cand); free(gray); free(S);
    return n;
}

/* ---- the pair stage of the SURF variant (MosaicWithoutPos.cpp:5389-5424) -------------------------------------------------------- */
/* exact 1-NN in L2 on float descriptors: d2 = sum_k (a_k - b_k)^2 accumulated with fmaf in ascending k; ties -> lowest train index */
void orc_bf_match_f32(const float* d1, int n1, const float* d2, int n2, int32_t* nn_idx, float* nn_dist)
{
    for (int i = 0; i < n1; i++) {
        float best = INFINITY; int bi = -1;
        const float* a = d1 + (size_t)i * 128;
        for (int j = 0; j < n2; j++) {
            const float* b = d2 + (size_t)j * 128;
            float acc = 0.0f;
            for (int k = 0; k < 128; k++) { const float df = a[k] - b[k]; acc = fmaf(df, df, acc); }
            if (acc < best) { best = acc; bi = j; }
        }
        nn_idx[i] = bi; nn_dist[i] = sqrtf(best);                  /* DMatch.distance of an L2 matcher */
    }
}

/* selection :5400-5424: every match below distT, distT lowered by 0.05 (double) until at most max_features remain; matches are
 * visited in sorted order (distance, queryIdx), so the output is sorted too.  Returns the count. */
int orc_select_by_distance(const int32_t* nn_idx, const float* nn_dist, int n1, const float* kp1xy, const float* kp2xy,
                           float match_dist, int max_features, orc_sfpoint* out1, orc_sfpoint* out2)
{
    int* order = (int*)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) order[i] = i;
    /* insertion into (distance, queryIdx) order: n1 is a few thousand in the tests; qsort needs context, so a simple merge sort */
    {
        int* tmp = (int*)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
        for (int wdt = 1; wdt < n1; wdt *= 2) {
            for (int lo = 0; lo < n1; lo += 2 * wdt) {
                int mid = lo + wdt < n1 ? lo + wdt : n1, hi = lo + 2 * wdt < n1 ? lo + 2 * wdt : n1;
                int a = lo, b = mid, k = lo;
                while (a < mid && b < hi) {
                    const int ia = order[a], ib = order[b];
                    const int take_a = nn_dist[ia] < nn_dist[ib] || (nn_dist[ia] == nn_dist[ib] && ia < ib);
                    tmp[k++] = take_a ? order[a++] : order[b++];
                }
                while (a < mid) tmp[k++] = order[a++];
                while (b < hi) tmp[k++] = order[b++];
            }
            memcpy(order, tmp, sizeof(int) * (size_t)n1);
        }
        free(tmp);
    }
    float distT = match_dist;
    int cnt;
    do {
        cnt = 0;
        for (int n = 0; n < n1; n++) {
            const int q = order[n];
            if (nn_idx[q] >= 0 && nn_dist[q] < distT) {
                out1[cnt].x = kp1xy[2 * q]; out1[cnt].y = kp1xy[2 * q + 1]; out1[cnt].id = q;
                out2[cnt].x = kp2xy[2 * nn_idx[q]]; out2[cnt].y = kp2xy[2 * nn_idx[q] + 1]; out2[cnt].id = nn_idx[q];
                cnt++;
            }
        }
        distT = (float)((double)distT - 0.05);                      /* distT -= dT with double dT = 0.05, :5401, :5422 */
    } while (cnt > max_features);
    free(order);
    return cnt;
}
