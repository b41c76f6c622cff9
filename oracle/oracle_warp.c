/* oracle/oracle_warp.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's hand-written inverse-mapping bilinear warps:
 *   ImageProjectionTransform           MosaicImage.cpp:1613-1758
 *   CMosaicByPose::MosaicImagesRefined MosaicWithoutPos.cpp:2194-2352 (float version)
 *   LaplacianPyramidBlending warp stage MosaicImage.cpp:2233-2460 (chips + validity masks)
 *   FindMasksByDistMap                 MosaicImage.cpp:1761-1881
 * Float op order is part of the contract; compile with -ffp-contract=off.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* the pixel expression MosaicWithoutPos.cpp:2331-2334 == MosaicImage.cpp:1715-1719:
 * (uchar)( s00*(1-p)*(1-q) + s01*(1-p)*q + s10*p*(1-q) + s11*p*q ), each term ((s*(.))*(.)),
 * summed left to right, truncating cast */
static inline uint8_t bilin(const uint8_t* s, int ws, int step, float p, float q)
{
    float omp = 1.0f - p, omq = 1.0f - q;
    float t0 = ((float)s[0] * omp) * omq;
    float t1 = ((float)s[step] * omp) * q;
    float t2 = ((float)s[ws] * p) * omq;
    float t3 = ((float)s[ws + step] * p) * q;
    float v = ((t0 + t1) + t2) + t3;
    return (uint8_t)(int)v;
}

/* two true divisions by the same denominator expression (MosaicWithoutPos.h:331-336;
 * spelled inline at MosaicImage.cpp:1639-1640, 1675-1676, 2359-2362) */
static inline void apply_div(const float* m, float x, float y, float* X, float* Y)
{
    *X = (m[0] * x + m[1] * y + m[2]) / (m[6] * x + m[7] * y + m[8]);
    *Y = (m[3] * x + m[4] * y + m[5]) / (m[6] * x + m[7] * y + m[8]);
}
/* matrix.h:1016-1024 ApplyProjectMat9: one reciprocal */
static inline void apply_recip9(const float* m, float x, float y, float* X, float* Y)
{
    float inv = 1.0f / (m[6] * x + m[7] * y + m[8]);
    *X = (m[0] * x + m[1] * y + m[2]) * inv;
    *Y = (m[3] * x + m[4] * y + m[5]) * inv;
}

void orc_free(void* p) { free(p); }

int orc_image_projection_transform(const uint8_t* src, int w, int h, int ws, int ch, const float h9[9],
                                   uint8_t** dst, int* dw, int* dh, int* dws)
{
    if (!src) return -1;
    float maxX = -(float)(1 << 29), maxY = -(float)(1 << 29), minX = (float)(1 << 29), minY = (float)(1 << 29);  /* :1621 */
    float cx[4] = {0.0f, (float)(w - 1), (float)(w - 1), 0.0f};
    float cy[4] = {0.0f, 0.0f, (float)(h - 1), (float)(h - 1)};
    for (int i = 0; i < 4; i++) {
        float X, Y; apply_div(h9, cx[i], cy[i], &X, &Y);
        if (X > maxX) maxX = X;
        if (X < minX) minX = X;
        if (Y > maxY) maxY = Y;
        if (Y < minY) minY = Y;
    }
    int nw = (int)(maxX - minX + 1.5f), nh = (int)(maxY - minY + 1.5f);      /* :1653-1654 */
    if (nw <= 0 || nh <= 0) return -1;                                        /* CreateBitmap8U returns NULL */
    int nws = (nw * ch + 3) / 4 * 4;                                          /* ImageIO.cpp:70 */
    uint8_t* out = (uint8_t*)calloc((size_t)nws * nh, 1);
    float dx = -minX, dy = -minY;
    float inv[9];
    if (orc_inverse_matrix(h9, 3, inv, 1e-6f) != 1) { /* reference would use uninitialised pInvM */ free(out); return -3; }
    int w1 = w - 1, h1 = h - 1;
    for (int yD = 0; yD < nh; yD++) {
        uint8_t* row = out + (size_t)yD * nws;
        for (int xD = 0; xD < nw; xD++) {
            float xf = (float)xD - dx, yf = (float)yD - dy;                   /* :1674 */
            float xs, ys; apply_div(inv, xf, yf, &xs, &ys);
            if (xs >= 0.0f && xs < (float)w1 && ys >= 0.0f && ys < (float)h1) {   /* :1683 */
                int yi = (int)ys, xi = (int)xs;
                float p = ys - (float)yi, q = xs - (float)xi;
                const uint8_t* s = src + (size_t)yi * ws + (size_t)ch * xi;
                for (int c = 0; c < ch; c++) row[ch * xD + c] = bilin(s + c, ws, ch, p, q);
            }
        }
    }
    *dst = out; *dw = nw; *dh = nh; *dws = nws;
    return 0;
}

int orc_mosaic_images_refined(const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                              const float* h9s, uint8_t* canvas, int* cw, int* ch, int* cws)
{
    float minX = (float)(1 << 29), minY = (float)(1 << 29), maxX = -(float)(1 << 29), maxY = -(float)(1 << 29);  /* :2200 */
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if (m[8] == 0.0f) continue;                                                                    /* :2205 */
        float cx[4] = {0.0f, (float)(w[k] - 1), (float)(w[k] - 1), 0.0f};
        float cy[4] = {0.0f, 0.0f, (float)(h[k] - 1), (float)(h[k] - 1)};
        for (int i = 0; i < 4; i++) {
            float X, Y; apply_div(m, cx[i], cy[i], &X, &Y);
            if (X < minX) minX = X;
            if (X > maxX) maxX = X;
            if (Y < minY) minY = Y;
            if (Y > maxY) maxY = Y;
        }
    }
    int mw = (int)(maxX - minX + 1.5f), mh = (int)(maxY - minY + 1.5f);                                 /* :2242-2243 */
    if (mw <= 0 || mh <= 0) return -2;
    int mws = (mw * 3 + 3) & ~3;                                                                        /* IplImage row padding */
    *cw = mw; *ch = mh; *cws = mws;
    if (!canvas) return 0;
    memset(canvas, 0, (size_t)mws * mh);
    float dGX = -minX, dGY = -minY;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if (m[8] == 0.0f) continue;
        int wS = w[k], hS = h[k], wsS = ws[k], w1 = wS - 1, h1 = hS - 1;
        float inv[9];
        if (orc_inverse_matrix(m, 3, inv, 1e-12f) != 1) continue;           /* reference: uninitialised invH; skipped here */
        float cx[4] = {0.0f, (float)w1, (float)w1, 0.0f};
        float cy[4] = {0.0f, 0.0f, (float)h1, (float)h1};
        float bminX = (float)(1 << 29), bminY = (float)(1 << 29), bmaxX = -(float)(1 << 29), bmaxY = -(float)(1 << 29);
        for (int i = 0; i < 4; i++) {
            float X, Y; apply_div(m, cx[i], cy[i], &X, &Y);
            X = X + (0.0f + dGX); Y = Y + (0.0f + dGY);                     /* :2287-2288 */
            if (X < bminX) bminX = X;
            if (X > bmaxX) bmaxX = X;
            if (Y < bminY) bminY = Y;
            if (Y > bmaxY) bmaxY = Y;
        }
        int begY = (int)(bminY - 0.5f), endY = (int)(bmaxY + 0.5f);         /* :2305-2306 */
        int begX = (int)(bminX - 0.5f), endX = (int)(bmaxX + 0.5f);
        /* the reference does not bound these by the canvas; they are inside it by construction, clamped
         * here so that a rounding excursion cannot write out of bounds */
        if (begY < 0) begY = 0;
        if (begX < 0) begX = 0;
        if (endY > mh - 1) endY = mh - 1;
        if (endX > mw - 1) endX = mw - 1;
        for (int yD = begY; yD <= endY; yD++) {
            uint8_t* row = canvas + (size_t)yD * mws;
            for (int xD = begX; xD <= endX; xD++) {
                float xm = (float)(xD - 0) - dGX, ym = (float)(yD - 0) - dGY;   /* :2313-2314 */
                float xs, ys; apply_div(inv, xm, ym, &xs, &ys);
                /* :2324-2327 `continue` unless 0<=ys<H-1 and 0<=xs<W-1; a NaN coordinate falls through
                 * those tests in the reference and then indexes with int(NaN): rejected here */
                if (!(ys >= 0.0f && ys < (float)h1)) continue;
                if (!(xs >= 0.0f && xs < (float)w1)) continue;
                int xi = (int)xs, yi = (int)ys;
                float p = ys - (float)yi, q = xs - (float)xi;
                const uint8_t* s = imgs[k] + (size_t)yi * wsS + 3 * (size_t)xi;
                row[3 * xD + 0] = bilin(s + 0, wsS, 3, p, q);
                row[3 * xD + 1] = bilin(s + 1, wsS, 3, p, q);
                row[3 * xD + 2] = bilin(s + 2, wsS, 3, p, q);
            }
        }
    }
    return 0;
}

/* MosaicImage.cpp:2233-2343.  h9s must already carry the resScale pre-multiplication of m[0..5]
 * (:2216-2223); keep[k]!=0 <=> vecAbandonInd[k]==1.  Canvas min/max start at 0 (:2234).  Fills one
 * orc_chip_info per kept image with m[8]!=0, in image order (the order of vecMask/vecCorners). */
int orc_chip_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep,
                    int* cw, int* ch, float* dG, orc_chip_info* chips)
{
    float maxX = 0.0f, maxY = 0.0f, minX = 0.0f, minY = 0.0f;
    float* bx0 = (float*)malloc(sizeof(float) * (size_t)n * 4);
    float *by0 = bx0 + n, *bx1 = bx0 + 2 * n, *by1 = bx0 + 3 * n;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if (!keep[k] || m[8] == 0.0f) continue;
        float cx[4] = {0.0f, (float)(w[k] - 1), (float)(w[k] - 1), 0.0f};
        float cy[4] = {0.0f, 0.0f, (float)(h[k] - 1), (float)(h[k] - 1)};
        float bMaxX = -(float)(1 << 29), bMaxY = -(float)(1 << 29), bMinX = (float)(1 << 29), bMinY = (float)(1 << 29);
        for (int i = 0; i < 4; i++) {
            float X, Y; apply_div(m, cx[i], cy[i], &X, &Y);
            if (X > maxX) maxX = X;
            if (X < minX) minX = X;
            if (Y > maxY) maxY = Y;
            if (Y < minY) minY = Y;
            if (X > bMaxX) bMaxX = X;
            if (X < bMinX) bMinX = X;
            if (Y > bMaxY) bMaxY = Y;
            if (Y < bMinY) bMinY = Y;
        }
        bx0[k] = bMinX; by0[k] = bMinY; bx1[k] = bMaxX; by1[k] = bMaxY;
    }
    float dGx = -minX, dGy = -minY;
    *cw = (int)(maxX - minX + 1.5f); *ch = (int)(maxY - minY + 1.5f);        /* :2291-2292 */
    dG[0] = dGx; dG[1] = dGy;
    int nv = 0;
    for (int k = 0; k < n; k++) {
        const float* m = h9s + 9 * k;
        if (!keep[k] || m[8] == 0.0f) continue;
        float bX = bx0[k] + dGx, bY = by0[k] + dGy, eX = bx1[k] + dGx, eY = by1[k] + dGy;   /* :2314-2317 */
        int begX = (int)bX, begY = (int)bY, endX = (int)(eX + 0.5f), endY = (int)(eY + 0.5f);
        float sx = (float)begX - bX, sy = (float)begY - bY;                                  /* :2324-2325 */
        orc_chip_info* c = chips + nv;
        c->x0 = begX; c->y0 = begY; c->w = endX - begX + 1; c->h = endY - begY + 1; c->sx = sx; c->sy = sy; c->img = k;
        float ox[4] = {0.0f, (float)(w[k] - 1), (float)(w[k] - 1), 0.0f};
        float oy[4] = {0.0f, 0.0f, (float)(h[k] - 1), (float)(h[k] - 1)};
        for (int i = 0; i < 4; i++) {
            float tx, ty; apply_recip9(m, ox[i], oy[i], &tx, &ty);                           /* :2334 */
            c->quad[2 * i]     = ((tx + dGx) + sx) - (float)begX;                            /* :2335-2336 */
            c->quad[2 * i + 1] = ((ty + dGy) + sy) - (float)begY;
        }
        nv++;
    }
    free(bx0);
    return nv;
}

/* MosaicImage.cpp:2343-2448 for one image: chip (3ch u8) + validity mask (255 valid / 0).  Chip pixels
 * with mask 0 are left untouched by the reference (uninitialised cvCreateImage memory); zero here. */
int orc_chip_warp(const uint8_t* src, int w, int h, int ws, const float h9[9], const float dG[2],
                  const orc_chip_info* ci, uint8_t* chip, int chip_ws, uint8_t* mask, int mask_ws)
{
    float inv[9];
    if (orc_inverse_matrix(h9, 3, inv, 1e-12f) != 1) return -3;              /* :2348 */
    int w1 = w - 1, h1 = h - 1;
    for (int yD = 0; yD < ci->h; yD++) {
        uint8_t* row = chip + (size_t)yD * chip_ws;
        uint8_t* mrow = mask + (size_t)yD * mask_ws;
        for (int xD = 0; xD < ci->w; xD++) {
            float xT = (((float)xD - dG[0]) - ci->sx) + (float)ci->x0;        /* :2356-2357 */
            float yT = (((float)yD - dG[1]) - ci->sy) + (float)ci->y0;
            float xs, ys; apply_div(inv, xT, yT, &xs, &ys);
            if (xs >= 0.0f && xs < (float)w1 && ys >= 0.0f && ys < (float)h1) {
                int yi = (int)ys, xi = (int)xs;
                float p = ys - (float)yi, q = xs - (float)xi;
                const uint8_t* s = src + (size_t)yi * ws + 3 * (size_t)xi;
                row[3 * xD + 0] = bilin(s + 0, ws, 3, p, q);
                row[3 * xD + 1] = bilin(s + 1, ws, 3, p, q);
                row[3 * xD + 2] = bilin(s + 2, ws, 3, p, q);
                mrow[xD] = 255;
            } else {
                row[3 * xD + 0] = 0; row[3 * xD + 1] = 0; row[3 * xD + 2] = 0;
                mrow[xD] = 0;
            }
        }
    }
    return 0;
}

/* ImageMath.cpp:88-103 */
static void line_of_2_points(float* a, float* b, float* c, float x1, float y1, float x2, float y2)
{
    if (fabs((double)(x1 - x2)) < 0.000001) { *a = 1.0f; *b = 0.0f; *c = -x1; }
    else { *a = (y1 - y2) / (x1 - x2); *b = -1.0f; *c = y1 - (*a) * x1; }
}

/* MosaicImage.cpp:1761-1881: per chip, distance of every valid pixel to the nearest of the 4 quad
 * edges, divided by the chip's max; then per canvas pixel the chip with the strictly largest value
 * (first wins, start 0) owns it: masks rewritten to 255 (owner) / 0. */
int orc_find_masks_by_distmap(uint8_t** masks, const int* mask_ws, const orc_chip_info* chips, int n, int rectW, int rectH)
{
    float** maps = (float**)calloc((size_t)n, sizeof(float*));
    for (int k = 0; k < n; k++) {
        const float* q = chips[k].quad;
        float A[4], B[4], C[4], inv[4];
        line_of_2_points(&A[0], &B[0], &C[0], q[0], q[1], q[2], q[3]);
        line_of_2_points(&A[1], &B[1], &C[1], q[2], q[3], q[4], q[5]);
        line_of_2_points(&A[2], &B[2], &C[2], q[4], q[5], q[6], q[7]);
        line_of_2_points(&A[3], &B[3], &C[3], q[6], q[7], q[0], q[1]);
        for (int i = 0; i < 4; i++) inv[i] = 1.0f / sqrtf(A[i] * A[i] + B[i] * B[i]);
        int w = chips[k].w, h = chips[k].h, ws = mask_ws[k];
        float* map = (float*)calloc((size_t)ws * h, sizeof(float));
        float maxDist = 0.0f;
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++) {
                if (masks[k][(size_t)r * ws + c] == 0) continue;
                float minDist = (float)(1 << 29);
                for (int i = 0; i < 4; i++) {
                    float d = fabsf(A[i] * (float)c + B[i] * (float)r + C[i]) * inv[i];
                    if (d < minDist) minDist = d;
                }
                map[(size_t)r * ws + c] = minDist;
                if (minDist > maxDist) maxDist = minDist;
            }
        for (int r = 0; r < h; r++) for (int c = 0; c < w; c++) map[(size_t)r * ws + c] = map[(size_t)r * ws + c] / maxDist;
        maps[k] = map;
    }
    for (int k = 0; k < n; k++) memset(masks[k], 0, (size_t)mask_ws[k] * chips[k].h);
    for (int r = 0; r < rectH; r++)
        for (int c = 0; c < rectW; c++) {
            int best = -1; float bd = 0.0f;
            for (int k = 0; k < n; k++) {
                int yC = r - chips[k].y0, xC = c - chips[k].x0;
                if (yC >= 0 && yC < chips[k].h && xC >= 0 && xC < chips[k].w) {
                    float d = maps[k][(size_t)yC * mask_ws[k] + xC];
                    if (d > bd) { bd = d; best = k; }
                }
            }
            if (best >= 0) masks[best][(size_t)(r - chips[best].y0) * mask_ws[best] + (c - chips[best].x0)] = 255;
        }
    for (int k = 0; k < n; k++) free(maps[k]);
    free(maps);
    return 0;
}
