/* oracle/oracle_sift.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the SIFT detect+describe step the reference runs through OpenCV 2.4.0:
 *     SiftFeatureDetector detector(2000, 3, 0.01, 20); detector.detect(img, kp);
 *     SiftDescriptorExtractor extractor;               extractor.compute(img, kp, desc);
 *                                                      (MosaicWithoutPos.cpp:4852-4872)
 *
 * PIN STATUS: PINNED BY THE REFERENCE'S COMMITTED OUTPUT for the detector, statistically for the descriptor.
 * The arithmetic lives in OpenCV 2.4.0 (`nonfree` sift.cpp, `imgproc` filter engine, `core` Matx), which the reference vendors as
 * headers + Win32 binaries only (3rdparty/opencv240, Release/opencv_*240.dll): no source, nothing that compiles or runs here.
 * Two things pin this restatement nevertheless:
 *  (1) its STRUCTURE and constants were read off the reference's own binary (Release/opencv_nonfree240.dll and
 *      opencv_imgproc240.dll, llvm-objdump; addresses below) and off the vendored headers, not guessed from the paper;
 *  (2) the reference's ONE committed run (Release/feature_temp/matchPairs.match: the inlier keypoints cv::SIFT 2.4.0 produced on
 *      Release/test_data/DSC00004..23.JPG, float32 x / y + index) is reproduced EXACTLY: all 8220 distinct keypoints of the 20
 *      frames appear in this oracle's output with bit-identical float32 coordinates AND at the index the file gives them
 *      (tests/test_sift_reference_run.py; with fused multiply-adds in the filter taps instead of separate products and sums only
 *      8100 of the 8220 stay bit-identical -- the check resolves single roundings).  That covers gray conversion, the 16-bit
 *      pyramid, DoG, extremum test, sub-pixel refinement, contrast / edge rejection, duplicate removal and the NUMBER of
 *      orientation peaks per point.  Size / angle / response / descriptors are not in that file: for them the evidence is
 *      statistical (same test: the reference's inlier correspondences are nearest neighbours under this oracle's descriptors).
 * Facts read off the binary (nonfree DLL unless noted):
 *   - NO image doubling: SIFT::operator() calls createInitialImage(image, doubleImageSize = false, sigma) (1001aa5d: push 0) and
 *     there is no keypoint rescaling after retainBest (1001ab87-1001ac0d); nOctaves = cvRound(log(min(w, h)) / log 2 - 2)
 *     (1001aaba-1001aaf8).  (The doubling, firstOctave = -1, came with later 2.4.x releases; rounds 1-2 of this oracle assumed it.)
 *   - the pyramids are 16-BIT FIXED POINT: gray.convertTo(gray_fpt, CV_16S = 3, 48.0) (10016112), buildDoGPyramid calls
 *     cv::subtract(.., dtype = 3) (10019ea6), findScaleSpaceExtrema's threshold is floor(0.5 * contrastThreshold / nOctaveLayers *
 *     12240), 12240 = 255 * 48 (10019f4b-10019f5b), adjustLocalExtrema scales by 1/12240, 1/24480, 1/48960 (10018675-10018751):
 *     `typedef short sift_wt; SIFT_FIXPT_SCALE = 48`;
 *   - createInitialImage(.., false, ..): cvtColor(BGR2GRAY = 6) -> convertTo(16S, x48) ->
 *     GaussianBlur(sigma = sqrtf(max(1.6^2 - 0.5^2, 0.01)), BORDER_DEFAULT = 4) (100160f3, 100162d8-10016349);
 *   - keypoint angle = (360 / 36) * bin (1001a6ef), no "360 - angle" anywhere; descriptor constants 3, sqrt(2) / 2, 1 / 360,
 *     pi / 180, 0.2, 512, FLT_EPSILON (10017833-10018159); orientation constants 4.5 (radius), 1.5 (sigma), 0.8 (peak ratio),
 *     1/16 - 4/16 - 6/16 smoothing (1001a5bc-1001a643, 100174fa-10017511);
 *   - Matx33f::solve(DECOMP_LU) is Cramer's rule in float, the expression of core/operations.hpp:742-750, 882-903 (vendored header).
 * The separable filter follows OpenCV's filter engine for 16S -> 32F -> 16S with a float kernel (imgproc/filter.cpp of the 2.4 line;
 * confirmed by (2)):
 *   kernel       width cvRound(8 sigma + 1) | 1, cv::getGaussianKernel(.., CV_32F): exp() rounded to float, summed in double,
 *                tap = (float)(tap * (1 / sum)); border reflect-101
 *   row pass     RowFilter<short, float>: s = k[0] * S[0]; s += k[i] * S[i] for ascending i -- product and sum rounded SEPARATELY
 *                (SSE2 scalar code, no fused multiply-add on that target);
 *   column pass  SymmColumnFilter<Cast<float, short>>: s = k[r] * C; s += k[r + j] * (S[+j] + S[-j]) for j = 1..r;
 *                result = saturate_cast<short>(s) = round half to even;
 *   next octave  every second pixel of level 3 (resize INTER_NEAREST to half size).
 *
 * STILL DEFINED HERE (third-party approximations that (2) cannot see; each an open parity risk of a few ulp in angle / descriptor):
 *   - exp / atan2 / sin / cos are the fixed polynomial forms below (OpenCV: table-driven cv::exp, fastAtan2 whose polynomial
 *     coefficients are the ones used here, MSVCR90 cosf / sinf), evaluated with fmaf in a fixed order.  cv::exp of this build
 *     (opencv_core240.dll: export 100881a0 -> Exp_32f 10085780) is POSITION DEPENDENT: blocks of eight array elements take an SSE2
 *     single-precision path (1008598e-10085bb7: float Horner form times (float)expTab[i & 63] * 2^(i >> 6)), the last n mod 8
 *     elements a double-precision x87 path (10085bc2-), and SIFT calls it on the compacted in-window samples -- the weight of a
 *     sample depends on the number and order of the samples of its window.  Not restated (it would still not pin a descriptor
 *     byte: MSVCR's cosf / sinf are not shipped, the histogram sums are float in that order, no descriptor is committed);
 *   - both histograms (36-bin orientation, 4x4x8 descriptor) are accumulated ORDER-FREE: every contribution v is quantised to
 *     q = rint(v * 2^10) and summed as a 64-bit integer; the bin value is (float)sum * 2^-10 (gradients are in 1/48 grey
 *     levels, so the resolution is 2e-5 grey levels; OpenCV adds floats in pixel order);
 *   - with nfeatures > 0 the keypoints are ordered by (response descending, octave, layer, row, column, orientation bin) and
 *     that is the output order (OpenCV: retainBest leaves an nth_element order); like retainBest, every keypoint whose response
 *     ties with the nfeatures-th strongest is kept too; with nfeatures <= 0 the order is OpenCV's own (generation order, see orc_sift).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define N_LAYERS 3
#define N_LEVELS (N_LAYERS + 3)
#define IMG_BORDER 5
#define MAX_INTERP 5
#define ORI_BINS 36
#define MAX_OCT 16
#define FIXPT_SCALE 48

/* ---------- fixed transcendental approximations (part of the definition) ----------------------- */
static inline float det_exp2f(float x)
{
    if (x < -126.0f) return 0.0f;
    if (x > 127.0f) x = 127.0f;
    float n = rintf(x);
    float f = x - n;                           /* [-0.5, 0.5] */
    float p = 1.535336188319500e-4f;           /* Cephes exp2f polynomial */
    p = fmaf(p, f, 1.339887440266574e-3f);
    p = fmaf(p, f, 9.618437357674640e-3f);
    p = fmaf(p, f, 5.550332471162809e-2f);
    p = fmaf(p, f, 2.402264791363012e-1f);
    p = fmaf(p, f, 6.931472028550421e-1f);
    p = fmaf(p, f, 1.0f);
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}
static inline float det_expf(float x) { return det_exp2f(x * 1.4426950408889634f); }

/* angle of (x, y) in degrees, [0, 360]; 7th order odd polynomial on the smaller/larger ratio (cv::fastAtan2's coefficients) */
static inline float det_atan2deg(float y, float x)
{
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446e-16f); c2 = c * c;
        a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    } else {
        c = ax / (ay + 2.220446e-16f); c2 = c * c;
        a = 90.0f - fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

static inline void det_sincosdeg(float deg, float* sn, float* cs)
{
    float q = rintf(deg * (1.0f / 90.0f));
    float r = fmaf(-90.0f, q, deg);            /* [-45, 45] */
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

/* ---------- second mode: orientation / descriptor arithmetic in the order the reference's binary runs it --------------------------------
 * orc_sift_set_mode(1) keeps everything above the orientation histogram as it is (that part is pinned bit for bit by the reference's
 * committed run) and evaluates calcOrientationHist / calcSIFTDescriptor the way opencv_nonfree240.dll + opencv_core240.dll do, as far as
 * they were read (llvm-objdump; file offsets / addresses of the Release/ DLLs):
 *   - the in-window samples are COMPACTED in scan order into arrays and cv::exp / cv::fastAtan2 / cv::magnitude run over the arrays;
 *   - cv::exp(const float*, float*, int) (core 100881a0 -> Exp_32f 10085780): blocks of eight elements through the SSE2 path -- clamp,
 *     x * (64 / ln 2) in double, cvtpd2dq, the fraction back to float and * 1/64, Horner form (((f + A1) f + A2) f + A3) f + A4 in float
 *     with separately rounded products and sums (A4..A1 at 1016bc98: 103.40865, 71.677414, 24.841499, 5.739531), times
 *     (float)expTab[i & 63] * 2^(i >> 6) (expTab at 1016a7c0: 2^(k/64) * 0.0096703711395723377, 64 doubles; tests/test_sift_binary_order.py
 *     compares the table built here with the DLL's bytes) -- and the last n mod 8 elements through the scalar path in double (x87, 53-bit
 *     precision control), rounded to float at the end: the weight of a sample depends on its position in the compacted array;
 *   - fastAtan2 (array form 100885a0; the same in its packed and scalar parts): the 7th order polynomial with separately rounded products
 *     and sums, c = min / (max + (float)DBL_EPSILON), 90 - a, 180 - a, 360 - a (mode 0 evaluates the same coefficients with fmaf);
 *   - magnitude (10088cf0): sqrt(x x + y y) in float, the same in both parts (= mode 0);
 *   - both histograms are float sums in scan order (mode 0: order-free 2^-10 fixed point);
 *   - cosf / sinf / powf are the C library's (the DLL calls MSVCR90's, which the reference does not ship; glibc's are used here -- both
 *     are < 1 ulp routines, they can still differ in a last bit).
 * Mode 1 is a MEASURING INSTRUMENT (how many angles / descriptor bytes move between the product's definition and the closest restatement
 * of the binary that can be made here; tests/test_sift_binary_order.py), not the parity checker: the product implements mode 0. */
static int g_sift_mode = 0;
void orc_sift_set_mode(int mode) { g_sift_mode = mode; }
int orc_sift_get_mode(void) { return g_sift_mode; }

#define EXPPOLY_A0 .9670371139572337719125840413672004409288e-2
static double g_exp_tab[64];
static int g_exp_tab_ready = 0;
const double* orc_cv_exp_table(void)
{
    if (!g_exp_tab_ready) { for (int k = 0; k < 64; k++) g_exp_tab[k] = exp2((double)k / 64.0) * EXPPOLY_A0; g_exp_tab_ready = 1; }
    return g_exp_tab;
}
/* cv::exp over an array, OpenCV 2.4.0's Exp_32f as the binary runs it (see above) */
void orc_cv_exp32f(const float* x, float* y, int n)
{
    const double* tab = orc_cv_exp_table();
    const float A4 = (float)(1.000000000000002438532970795181890933776 / EXPPOLY_A0), A3 = (float)(.6931471805521448196800669615864773144641 / EXPPOLY_A0),
                A2 = (float)(.2402265109513301490103372422686535526573 / EXPPOLY_A0), A1 = (float)(.5550339366753125211915322047004666939128e-1 / EXPPOLY_A0);
    const double prescale = 1.4426950408889634073599246810019 * 64.0, postscale = 1.0 / 64.0, max_val = 3000.0 * 64.0;
    const float maxv = (float)(max_val / prescale), minv = (float)(-max_val / prescale);
    int i = 0;
    if (n >= 8) {
        for (; i <= n - 8; i += 8)
            for (int q = 0; q < 8; q++) {
                float xf = x[i + q];
                xf = xf > minv ? xf : minv; xf = xf < maxv ? xf : maxv;             /* maxps, minps */
                double xd = (double)xf * prescale;
                int xi = (int)lrint(xd);                                           /* cvtpd2dq: round to nearest even */
                xd = xd - (double)xi;
                float f = (float)xd;
                f = f * (float)postscale;
                int xs = xi < -32768 ? -32768 : (xi > 32767 ? 32767 : xi);          /* packssdw */
                int e = (xs >> 6) + 127; e = e < 0 ? 0 : (e > 255 ? 255 : e);
                union { uint32_t u; float f; } sc; sc.u = (uint32_t)e << 23;
                float yf = (float)tab[xs & 63];
                yf = yf * sc.f;
                float z = f + A1;
                z = z * f; z = z + A2;
                z = z * f; z = z + A3;
                z = z * f; z = z + A4;
                y[i + q] = z * yf;
            }
    }
    for (; i < n; i++) {
        union { uint32_t u; float f; } in; in.f = x[i];
        double x0 = (double)x[i] * prescale;
        if (((in.u >> 23) & 255) > 127 + 10) x0 = (in.u >> 31) ? -max_val : max_val;
        int val0 = (int)lrint(x0);
        int t = (val0 >> 6) + 127; t = !(t & ~255) ? t : (t < 0 ? 0 : 255);
        union { uint32_t u; float f; } b; b.u = (uint32_t)t << 23;
        x0 = (x0 - (double)val0) * postscale;
        double poly = ((((x0 + (double)A1) * x0 + (double)A2) * x0 + (double)A3) * x0 + (double)A4);
        y[i] = (float)(((double)b.f * tab[val0 & 63]) * poly);
    }
}
/* cv::fastAtan2, degrees: every product and sum rounded on its own (the binary's mulss / addss, mulps / addps) */
static inline float bin_atan2deg(float y, float x)
{
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + 2.220446e-16f); c2 = c * c; a = c2 * p7; a = a + p5; a = a * c2; a = a + p3; a = a * c2; a = a + p1; a = a * c; }
    else { c = ax / (ay + 2.220446e-16f); c2 = c * c; a = c2 * p7; a = a + p5; a = a * c2; a = a + p3; a = a * c2; a = a + p1; a = a * c; a = 90.0f - a; }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

/* ---------- Gaussian pyramid, 16-bit fixed point -------------------------------------------------- */
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}

/* saturate_cast<short>(float): round half to even (cvRound), clamp */
static inline int16_t sat_short(float v)
{
    long q = lrintf(v);
    return (int16_t)(q < -32768 ? -32768 : (q > 32767 ? 32767 : q));
}

static int gauss_kernel(double sigma, float* k)      /* returns radius */
{
    int ksize = ((int)lrint(sigma * 8.0 + 1.0)) | 1;       /* cvRound(sigma * 4 * 2 + 1) | 1 for every depth but 8U */
    int r = ksize / 2;
    /* rounding points of cv::getGaussianKernel(ksize, sigma, CV_32F): every exp() is rounded to float first, the FLOATS are summed in
       double, each tap is (float)(tap * (1 / sum)) */
    double sum = 0.0;
    double scale2x = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; i++) { double x = (double)i - (double)(ksize - 1) * 0.5; k[i] = (float)exp(scale2x * x * x); sum += (double)k[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; i++) k[i] = (float)((double)k[i] * sum);
    return r;
}

/* volatile-free way to keep gcc from contracting a * b + c: the oracle is compiled with -ffp-contract=off (Makefile) */
static void gauss_blur16(const int16_t* src, int16_t* dst, float* tmp, int w, int h, double sigma)
{
    float k[64];
    int r = gauss_kernel(sigma, k);
    /* row pass: tmp(y, x) = k[0] * S(x - r) + k[1] * S(x - r + 1) + ... (ascending taps, product and sum rounded separately) */
    float* ext = (float*)malloc(sizeof(float) * (size_t)(w + 2 * r));
    for (int y = 0; y < h; y++) {
        const int16_t* s = src + (size_t)y * w;
        for (int x = -r; x < w + r; x++) ext[x + r] = (float)s[reflect101(x, w)];
        float* t = tmp + (size_t)y * w;
        { const float k0 = k[0]; for (int x = 0; x < w; x++) t[x] = k0 * ext[x]; }
        for (int i = 1; i <= 2 * r; i++) { const float ki = k[i]; const float* e = ext + i; for (int x = 0; x < w; x++) { float p = ki * e[x]; t[x] = t[x] + p; } }
    }
    free(ext);
    /* column pass: centre tap first, then the symmetric pairs outwards; round half to even into 16 bits */
    float* acc = (float*)malloc(sizeof(float) * (size_t)w);
    for (int y = 0; y < h; y++) {
        const float* c = tmp + (size_t)y * w;
        { const float k0 = k[r]; for (int x = 0; x < w; x++) acc[x] = k0 * c[x]; }
        for (int j = 1; j <= r; j++) {
            const float kj = k[r + j];
            const float* a = tmp + (size_t)reflect101(y + j, h) * w;
            const float* b = tmp + (size_t)reflect101(y - j, h) * w;
            for (int x = 0; x < w; x++) { float sm = a[x] + b[x]; float p = kj * sm; acc[x] = acc[x] + p; }
        }
        int16_t* d = dst + (size_t)y * w;
        for (int x = 0; x < w; x++) d[x] = sat_short(acc[x]);
    }
    free(acc);
}

typedef struct { int w, h; int16_t* lv[N_LEVELS]; } octave_t;

typedef struct {
    uint32_t resp_bits; int o, layer, r, c, bin;
    float x, y, size, angle, response, xi, scl;
    float ptx, pty;      /* octave coordinates c + xc, r + xr */
} cand_t;

static inline int dogv(const octave_t* oc, int lvl, int r, int c)      /* DoG[lvl] = G[lvl+1] - G[lvl], saturated 16-bit subtract */
{
    size_t o = (size_t)r * oc->w + c;
    int v = (int)oc->lv[lvl + 1][o] - (int)oc->lv[lvl][o];
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

/* Matx33f::solve(b, DECOMP_LU) = Matx_FastSolveOp<float, 3, 1>: Cramer's rule, core/operations.hpp:742-750, 882-903.  0 if det == 0 */
static int solve3_cramer(const float a[3][3], const float b[3], float x[3])
{
    float det = (a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2])) + a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (det == 0.0f) { x[0] = x[1] = x[2] = 0.0f; return 0; }
    float d = 1.0f / det;
    x[0] = d * ((b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2])) + a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * ((a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0])) + a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * ((a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0])) + b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
    return 1;
}

/* sub-pixel refinement + contrast / edge rejection. returns 1 and fills (layer,r,c,xi,xr,xc,contr) */
static int adjust_extremum(const octave_t* oc, int* layer, int* r, int* c, float* xi, float* xr, float* xc, float* contr,
                           float contrast_thr, float edge_thr)
{
    const float img_scale = 1.0f / (float)(255 * FIXPT_SCALE);
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    int it = 0;
    float dD[3], X[3] = {0, 0, 0};
    for (; it < MAX_INTERP; it++) {
        int L = *layer, R = *r, Cc = *c;
        /* integer differences of 16-bit samples (exact), one float product each */
        dD[0] = (float)(dogv(oc, L, R, Cc + 1) - dogv(oc, L, R, Cc - 1)) * deriv_scale;
        dD[1] = (float)(dogv(oc, L, R + 1, Cc) - dogv(oc, L, R - 1, Cc)) * deriv_scale;
        dD[2] = (float)(dogv(oc, L + 1, R, Cc) - dogv(oc, L - 1, R, Cc)) * deriv_scale;
        float v2 = (float)dogv(oc, L, R, Cc) * 2.0f;
        float dxx = ((float)(dogv(oc, L, R, Cc + 1) + dogv(oc, L, R, Cc - 1)) - v2) * second_scale;
        float dyy = ((float)(dogv(oc, L, R + 1, Cc) + dogv(oc, L, R - 1, Cc)) - v2) * second_scale;
        float dss = ((float)(dogv(oc, L + 1, R, Cc) + dogv(oc, L - 1, R, Cc)) - v2) * second_scale;
        float dxy = (float)(dogv(oc, L, R + 1, Cc + 1) - dogv(oc, L, R + 1, Cc - 1) - dogv(oc, L, R - 1, Cc + 1) + dogv(oc, L, R - 1, Cc - 1)) * cross_scale;
        float dxs = (float)(dogv(oc, L + 1, R, Cc + 1) - dogv(oc, L + 1, R, Cc - 1) - dogv(oc, L - 1, R, Cc + 1) + dogv(oc, L - 1, R, Cc - 1)) * cross_scale;
        float dys = (float)(dogv(oc, L + 1, R + 1, Cc) - dogv(oc, L + 1, R - 1, Cc) - dogv(oc, L - 1, R + 1, Cc) + dogv(oc, L - 1, R - 1, Cc)) * cross_scale;
        const float A[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        solve3_cramer(A, dD, X);
        *xi = -X[2]; *xr = -X[1]; *xc = -X[0];
        if (fabsf(*xi) < 0.5f && fabsf(*xr) < 0.5f && fabsf(*xc) < 0.5f) break;
        if (fabsf(*xi) > 7.0e8f || fabsf(*xr) > 7.0e8f || fabsf(*xc) > 7.0e8f) return 0;      /* INT_MAX/3 */
        if (!(*xi == *xi) || !(*xr == *xr) || !(*xc == *xc)) return 0;
        *c += (int)rintf(*xc); *r += (int)rintf(*xr); *layer += (int)rintf(*xi);
        if (*layer < 1 || *layer > N_LAYERS || *c < IMG_BORDER || *c >= oc->w - IMG_BORDER || *r < IMG_BORDER || *r >= oc->h - IMG_BORDER) return 0;
    }
    if (it >= MAX_INTERP) return 0;
    {
        int L = *layer, R = *r, Cc = *c;
        dD[0] = (float)(dogv(oc, L, R, Cc + 1) - dogv(oc, L, R, Cc - 1)) * deriv_scale;
        dD[1] = (float)(dogv(oc, L, R + 1, Cc) - dogv(oc, L, R - 1, Cc)) * deriv_scale;
        dD[2] = (float)(dogv(oc, L + 1, R, Cc) - dogv(oc, L - 1, R, Cc)) * deriv_scale;
        float t = (0.0f + dD[0] * (*xc)) + dD[1] * (*xr);                               /* Matx::dot: s = 0; s += v[i] * m[i] */
        t = t + dD[2] * (*xi);
        *contr = (float)dogv(oc, L, R, Cc) * img_scale + t * 0.5f;
        if (fabsf(*contr) * (float)N_LAYERS < contrast_thr) return 0;
        float v2 = (float)dogv(oc, L, R, Cc) * 2.0f;
        float dxx = ((float)(dogv(oc, L, R, Cc + 1) + dogv(oc, L, R, Cc - 1)) - v2) * second_scale;
        float dyy = ((float)(dogv(oc, L, R + 1, Cc) + dogv(oc, L, R - 1, Cc)) - v2) * second_scale;
        float dxy = (float)(dogv(oc, L, R + 1, Cc + 1) - dogv(oc, L, R + 1, Cc - 1) - dogv(oc, L, R - 1, Cc + 1) + dogv(oc, L, R - 1, Cc - 1)) * cross_scale;
        float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0.0f || (tr * tr) * edge_thr >= ((edge_thr + 1.0f) * (edge_thr + 1.0f)) * det) return 0;
    }
    return 1;
}

static int cand_cmp(const void* a, const void* b)
{
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->resp_bits != y->resp_bits) return x->resp_bits > y->resp_bits ? -1 : 1;     /* response descending */
    if (x->o != y->o) return x->o < y->o ? -1 : 1;
    if (x->layer != y->layer) return x->layer < y->layer ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->bin != y->bin) return x->bin < y->bin ? -1 : 1;
    return 0;
}

#define HIST_Q 1024.0f                       /* order-free accumulation: contributions quantised to 2^-10 (gradients are x48 already) */
#define FIXQ(v) ((int64_t)llrintf((v) * HIST_Q))

static void describe(const octave_t* oc, const cand_t* k, uint8_t* out)
{
    const int d = 4, n = 8;
    const int16_t* img = oc->lv[k->layer];
    const int rows = oc->h, cols = oc->w;
    const float scl = k->scl;
    const int px = (int)rintf(k->ptx), py = (int)rintf(k->pty);
    float sin_t, cos_t;
    det_sincosdeg(k->angle, &sin_t, &cos_t);
    const float bins_per_deg = (float)n / 360.0f;
    const float exp_scale = -1.0f / ((float)(d * d) * 0.5f);
    const float hist_width = 3.0f * scl;
    const int radius = (int)rintf(hist_width * 1.4142135623730951f * (float)(d + 1) * 0.5f);
    cos_t = cos_t / hist_width; sin_t = sin_t / hist_width;
    int64_t hq[(4 + 2) * (4 + 2) * (8 + 2)];
    float hist[(4 + 2) * (4 + 2) * (8 + 2)];
    for (int i = 0; i < (d + 2) * (d + 2) * (n + 2); i++) hq[i] = 0;
    if (g_sift_mode == 1) {
        /* the binary's order (see orc_sift_set_mode): cosf / sinf of the C library, samples compacted in scan order, fastAtan2 / magnitude /
           cv::exp over the arrays, float sums in that order */
        float cs = cosf(k->angle * (float)(3.14159265358979323846 / 180.0)), sn = sinf(k->angle * (float)(3.14159265358979323846 / 180.0));
        cs = cs / hist_width; sn = sn / hist_width;
        const int len0 = (2 * radius + 1) * (2 * radius + 1);
        float* buf = (float*)malloc(sizeof(float) * 6 * (size_t)len0);
        float *X = buf, *Y = X + len0, *Ori = Y + len0, *W = Ori + len0, *RB = W + len0, *CB = RB + len0;
        int m = 0;
        for (int i = -radius; i <= radius; i++)
            for (int j = -radius; j <= radius; j++) {
                float c_rot = (float)j * cs - (float)i * sn;
                float r_rot = (float)j * sn + (float)i * cs;
                float rbin = r_rot + (float)(d / 2) - 0.5f;
                float cbin = c_rot + (float)(d / 2) - 0.5f;
                int r = py + i, c = px + j;
                if (!(rbin > -1.0f && rbin < (float)d && cbin > -1.0f && cbin < (float)d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
                X[m] = (float)((int)img[(size_t)r * cols + c + 1] - (int)img[(size_t)r * cols + c - 1]);
                Y[m] = (float)((int)img[(size_t)(r - 1) * cols + c] - (int)img[(size_t)(r + 1) * cols + c]);
                RB[m] = rbin; CB[m] = cbin;
                W[m] = (c_rot * c_rot + r_rot * r_rot) * exp_scale;
                m++;
            }
        for (int q = 0; q < m; q++) Ori[q] = bin_atan2deg(Y[q], X[q]);
        for (int q = 0; q < m; q++) { float sq = X[q] * X[q], sq2 = Y[q] * Y[q]; Y[q] = sqrtf(sq + sq2); }      /* Mag = Y, in place */
        orc_cv_exp32f(W, W, m);
        for (int i = 0; i < (d + 2) * (d + 2) * (n + 2); i++) hist[i] = 0.0f;
        for (int q = 0; q < m; q++) {
            float rbin = RB[q], cbin = CB[q];
            float obin = (Ori[q] - k->angle) * bins_per_deg;
            float mag = Y[q] * W[q];
            float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
            rbin -= r0f; cbin -= c0f; obin -= o0f;
            int r0 = (int)r0f, c0 = (int)c0f, o0 = (int)o0f;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
            float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
            float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
            hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
            hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
            hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
        }
        free(buf);
    } else
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            float c_rot = (float)j * cos_t - (float)i * sin_t;
            float r_rot = (float)j * sin_t + (float)i * cos_t;
            float rbin = r_rot + (float)(d / 2) - 0.5f;
            float cbin = c_rot + (float)(d / 2) - 0.5f;
            int r = py + i, c = px + j;
            if (!(rbin > -1.0f && rbin < (float)d && cbin > -1.0f && cbin < (float)d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
            float dx = (float)((int)img[(size_t)r * cols + c + 1] - (int)img[(size_t)r * cols + c - 1]);
            float dy = (float)((int)img[(size_t)(r - 1) * cols + c] - (int)img[(size_t)(r + 1) * cols + c]);
            float ori = det_atan2deg(dy, dx);
            float mag = sqrtf(dx * dx + dy * dy) * det_expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (ori - k->angle) * bins_per_deg;
            float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
            rbin -= r0f; cbin -= c0f; obin -= o0f;
            int r0 = (int)r0f, c0 = (int)c0f, o0 = (int)o0f;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
            float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
            float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
            float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
            int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
            hq[idx] += FIXQ(v_rco000); hq[idx + 1] += FIXQ(v_rco001);
            hq[idx + (n + 2)] += FIXQ(v_rco010); hq[idx + (n + 3)] += FIXQ(v_rco011);
            hq[idx + (d + 2) * (n + 2)] += FIXQ(v_rco100); hq[idx + (d + 2) * (n + 2) + 1] += FIXQ(v_rco101);
            hq[idx + (d + 3) * (n + 2)] += FIXQ(v_rco110); hq[idx + (d + 3) * (n + 2) + 1] += FIXQ(v_rco111);
        }
    if (g_sift_mode != 1) for (int i = 0; i < (d + 2) * (d + 2) * (n + 2); i++) hist[i] = (float)hq[i] * (1.0f / HIST_Q);
    float dst[128];
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int q = 0; q < n; q++) dst[(i * d + j) * n + q] = hist[idx + q];
        }
    float nrm2 = 0.0f;
    for (int q = 0; q < 128; q++) { float sq = dst[q] * dst[q]; nrm2 = nrm2 + sq; }
    float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.0f;
    for (int q = 0; q < 128; q++) { float v = dst[q] < thr ? dst[q] : thr; dst[q] = v; float sq = v * v; nrm2 = nrm2 + sq; }
    float nn = sqrtf(nrm2);
    float fac = 512.0f / (nn > FLT_EPSILON ? nn : FLT_EPSILON);
    for (int q = 0; q < 128; q++) {
        float v = rintf(dst[q] * fac);
        out[q] = (uint8_t)(v < 0.0f ? 0 : (v > 255.0f ? 255 : (int)v));
    }
}

int orc_sift(const uint8_t* bgr, int w, int h, int ws, int nfeatures, orc_keypoint* kp_out, uint8_t* desc_out, int max_kp)
{
    const double sigma = 1.6;
    const float contrast_thr = 0.01f, edge_thr = 20.0f;
    const int W = w, H = h;                  /* octave 0 is the image itself: 2.4.0 calls createInitialImage(image, false, sigma) */
    /* 1: gray (8-bit BGR2GRAY fixed point), x48 into 16 bits */
    int16_t* gray = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = bgr + (size_t)y * ws + 3 * x;
            gray[(size_t)y * w + x] = (int16_t)(((1868 * p[0] + 9617 * p[1] + 4899 * p[2] + 8192) >> 14) * FIXPT_SCALE);
        }
    /* 2: base = Gaussian blur of the 16-bit gray image with sqrt(sigma^2 - 0.5^2): NO image doubling in this OpenCV version */
    int16_t* up = gray;
    /* 3: pyramid: 6 Gaussian levels per octave built incrementally, sigma_i = sqrt((s k^i)^2 - (s k^(i-1))^2), k = 2^(1/3); next
     * octave = every second pixel of level 3 (resize INTER_NEAREST to half size) */
    int nOct = (int)lrint(log((double)(W < H ? W : H)) / log(2.0) - 2.0) ;
    if (nOct > MAX_OCT) nOct = MAX_OCT;
    double sig[N_LEVELS];
    {
        double k = pow(2.0, 1.0 / N_LAYERS);
        sig[0] = sigma;
        for (int i = 1; i < N_LEVELS; i++) { double sp = pow(k, (double)(i - 1)) * sigma, st = sp * k; sig[i] = sqrt(st * st - sp * sp); }
    }
    octave_t oc[MAX_OCT];
    memset(oc, 0, sizeof(oc));
    float* tmp = (float*)malloc(sizeof(float) * (size_t)W * H);
    int no = 0;
    for (int o = 0; o < nOct; o++) {
        int ow = W >> o, oh = H >> o;
        if (ow < 2 * IMG_BORDER + 2 || oh < 2 * IMG_BORDER + 2) break;      /* no keypoint can exist in smaller octaves */
        oc[o].w = ow; oc[o].h = oh;
        for (int i = 0; i < N_LEVELS; i++) oc[o].lv[i] = (int16_t*)malloc(sizeof(int16_t) * (size_t)ow * oh);
        if (o == 0) {
            float sd = sqrtf(fmaxf((float)sigma * (float)sigma - 0.25f, 0.01f));            /* float, as the binary computes it */
            gauss_blur16(up, oc[0].lv[0], tmp, ow, oh, (double)sd);
        } else {
            const int16_t* s = oc[o - 1].lv[N_LAYERS];
            int pw = oc[o - 1].w;
            for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++) oc[o].lv[0][(size_t)y * ow + x] = s[(size_t)(2 * y) * pw + 2 * x];
        }
        for (int i = 1; i < N_LEVELS; i++) gauss_blur16(oc[o].lv[i - 1], oc[o].lv[i], tmp, ow, oh, sig[i]);
        no = o + 1;
    }
    free(up); free(tmp);
    /* 4-6: extrema -> refined keypoints with orientations */
    const int threshold = (int)floor(0.5 * 0.01 / N_LAYERS * 255 * FIXPT_SCALE);          /* 20 */
    size_t cap = 1 << 16, ncand = 0;
    cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * cap);
    for (int o = 0; o < no; o++) {
        const octave_t* O = &oc[o];
        /* duplicates: several start points may converge to one location; claim each final location once */
        uint8_t* claimed = (uint8_t*)calloc((size_t)O->w * O->h, 1);
        for (int layer = 1; layer <= N_LAYERS; layer++)
            for (int r = IMG_BORDER; r < O->h - IMG_BORDER; r++)
                for (int c = IMG_BORDER; c < O->w - IMG_BORDER; c++) {
                    int val = dogv(O, layer, r, c);
                    if (!(abs(val) > threshold)) continue;
                    int ismax = val > 0, ok = 1;
                    for (int dl = -1; dl <= 1 && ok; dl++)
                        for (int dr = -1; dr <= 1 && ok; dr++)
                            for (int dc = -1; dc <= 1; dc++) {
                                if (!dl && !dr && !dc) continue;
                                int v = dogv(O, layer + dl, r + dr, c + dc);
                                if (ismax ? !(val >= v) : !(val <= v)) { ok = 0; break; }
                            }
                    if (!ok) continue;
                    int L = layer, R = r, Cc = c; float xi, xr, xc, contr;
                    if (!adjust_extremum(O, &L, &R, &Cc, &xi, &xr, &xc, &contr, contrast_thr, edge_thr)) continue;
                    uint8_t bit = (uint8_t)(1u << L);
                    if (claimed[(size_t)R * O->w + Cc] & bit) continue;
                    claimed[(size_t)R * O->w + Cc] |= bit;
                    float scl = g_sift_mode == 1 ? (float)sigma * powf(2.0f, ((float)L + xi) / (float)N_LAYERS)      /* kpt.size = sigma * powf(2.f, ..) * (1 << o) * 2, scl_octv = size * 0.5f / (1 << o) */
                                                 : (float)sigma * det_exp2f(((float)L + xi) / (float)N_LAYERS);
                    /* orientation histogram on the Gaussian level L of this octave */
                    const int16_t* img = O->lv[L];
                    int radius = (int)rintf(4.5f * scl);
                    float osig = 1.5f * scl;
                    float expf_scale = -1.0f / (2.0f * osig * osig);
                    float th[ORI_BINS], hs[ORI_BINS];
                    int64_t tq[ORI_BINS];
                    for (int b = 0; b < ORI_BINS; b++) tq[b] = 0;
                    if (g_sift_mode == 1) {
                        /* calcOrientationHist as the binary runs it: compacted samples, cv::exp / fastAtan2 / magnitude over the arrays, float sums */
                        const int len0 = (2 * radius + 1) * (2 * radius + 1);
                        float* buf = (float*)malloc(sizeof(float) * 4 * (size_t)len0);
                        float *X = buf, *Y = X + len0, *Ori = Y + len0, *W = Ori + len0;
                        int m = 0;
                        for (int i = -radius; i <= radius; i++) {
                            int y = R + i;
                            if (y <= 0 || y >= O->h - 1) continue;
                            for (int j = -radius; j <= radius; j++) {
                                int x = Cc + j;
                                if (x <= 0 || x >= O->w - 1) continue;
                                X[m] = (float)((int)img[(size_t)y * O->w + x + 1] - (int)img[(size_t)y * O->w + x - 1]);
                                Y[m] = (float)((int)img[(size_t)(y - 1) * O->w + x] - (int)img[(size_t)(y + 1) * O->w + x]);
                                W[m] = (float)(i * i + j * j) * expf_scale;
                                m++;
                            }
                        }
                        orc_cv_exp32f(W, W, m);
                        for (int q = 0; q < m; q++) Ori[q] = bin_atan2deg(Y[q], X[q]);
                        for (int q = 0; q < m; q++) { float sq = X[q] * X[q], sq2 = Y[q] * Y[q]; X[q] = sqrtf(sq + sq2); }      /* Mag = X, in place */
                        for (int b = 0; b < ORI_BINS; b++) th[b] = 0.0f;
                        for (int q = 0; q < m; q++) {
                            int bin = (int)rintf(((float)ORI_BINS / 360.0f) * Ori[q]);
                            if (bin >= ORI_BINS) bin -= ORI_BINS;
                            if (bin < 0) bin += ORI_BINS;
                            float t = W[q] * X[q];
                            th[bin] = th[bin] + t;
                        }
                        free(buf);
                    } else
                    for (int i = -radius; i <= radius; i++) {
                        int y = R + i;
                        if (y <= 0 || y >= O->h - 1) continue;
                        for (int j = -radius; j <= radius; j++) {
                            int x = Cc + j;
                            if (x <= 0 || x >= O->w - 1) continue;
                            float dx = (float)((int)img[(size_t)y * O->w + x + 1] - (int)img[(size_t)y * O->w + x - 1]);
                            float dy = (float)((int)img[(size_t)(y - 1) * O->w + x] - (int)img[(size_t)(y + 1) * O->w + x]);
                            float wgt = det_expf((float)(i * i + j * j) * expf_scale);
                            float ang = det_atan2deg(dy, dx);
                            float mag = sqrtf(dx * dx + dy * dy);
                            int bin = (int)rintf(((float)ORI_BINS / 360.0f) * ang);
                            if (bin >= ORI_BINS) bin -= ORI_BINS;
                            if (bin < 0) bin += ORI_BINS;
                            float t = wgt * mag;
                            tq[bin] += FIXQ(t);
                        }
                    }
                    if (g_sift_mode != 1) for (int b = 0; b < ORI_BINS; b++) th[b] = (float)tq[b] * (1.0f / HIST_Q);
                    float omax = 0.0f;
                    for (int b = 0; b < ORI_BINS; b++) {
                        float m2 = th[(b + ORI_BINS - 2) % ORI_BINS], m1 = th[(b + ORI_BINS - 1) % ORI_BINS];
                        float p1 = th[(b + 1) % ORI_BINS], p2 = th[(b + 2) % ORI_BINS];
                        hs[b] = ((m2 + p2) * (1.0f / 16.0f) + (m1 + p1) * (4.0f / 16.0f)) + th[b] * (6.0f / 16.0f);
                        if (hs[b] > omax) omax = hs[b];
                    }
                    float mag_thr = omax * 0.8f;
                    for (int b = 0; b < ORI_BINS; b++) {
                        int l = b > 0 ? b - 1 : ORI_BINS - 1, r2 = b < ORI_BINS - 1 ? b + 1 : 0;
                        if (hs[b] > hs[l] && hs[b] > hs[r2] && hs[b] >= mag_thr) {
                            float bf = (float)b + (0.5f * (hs[l] - hs[r2])) / ((hs[l] - 2.0f * hs[b]) + hs[r2]);
                            bf = bf < 0.0f ? (float)ORI_BINS + bf : (bf >= (float)ORI_BINS ? bf - (float)ORI_BINS : bf);
                            if (ncand == cap) { cap *= 2; cand = (cand_t*)realloc(cand, sizeof(cand_t) * cap); }
                            cand_t* k = &cand[ncand++];
                            float resp = fabsf(contr);
                            memcpy(&k->resp_bits, &resp, 4);
                            k->o = o; k->layer = L; k->r = R; k->c = Cc; k->bin = b;
                            k->ptx = (float)Cc + xc; k->pty = (float)R + xr;
                            /* image coordinates: kpt.pt = (c + xc) * (1 << octave), kpt.size = sigma * 2^((layer + xi) / 3) * (1 << octave) * 2 */
                            float s2 = (float)(1 << o);
                            k->x = k->ptx * s2; k->y = k->pty * s2;
                            k->size = (scl * s2) * 2.0f;
                            k->angle = (360.0f / (float)ORI_BINS) * bf;
                            k->response = resp; k->xi = xi; k->scl = scl;
                        }
                    }
                }
        free(claimed);
    }
    /* 7: order, keep the strongest.  nfeatures <= 0 (cv::SIFT's "keep all"): every keypoint in GENERATION order -- octave, layer,
     * row, column of the extremum the refinement started from, orientation peaks in bin order, first of identical ones kept --
     * which is the order of OpenCV's vector when retainBest does not run: the ids stored in the reference's committed
     * matchPairs.match are indices into exactly this list (tests/test_sift_reference_run.py) */
    if (nfeatures > 0) qsort(cand, ncand, sizeof(cand_t), cand_cmp);
    size_t keep = (nfeatures <= 0 || ncand < (size_t)nfeatures) ? ncand : (size_t)nfeatures;
    /* KeyPointsFilter::retainBest (features2d keypoint.cpp, called by SIFT::operator() with nfeatures): nth_element, then everything
     * whose response is >= that of the nfeatures-th strongest stays -- ties at the boundary are all kept (two orientations of one point
     * share their response, so a tie there is an ordinary event) */
    if (nfeatures > 0) while (keep < ncand && cand[keep].response == cand[nfeatures - 1].response) keep++;
    if (keep > (size_t)max_kp) keep = (size_t)max_kp;
    for (size_t i = 0; i < keep; i++) {
        const cand_t* k = &cand[i];
        kp_out[i].x = k->x; kp_out[i].y = k->y; kp_out[i].size = k->size; kp_out[i].angle = k->angle; kp_out[i].response = k->response;
        /* OpenCV packing: octave | layer << 8 | round((xi + 0.5) * 255) << 16 */
        kp_out[i].octave = (k->o & 255) | (k->layer << 8) | (((int)rintf((k->xi + 0.5f) * 255.0f)) << 16);
        kp_out[i].class_id = -1;
        if (desc_out) describe(&oc[k->o], k, desc_out + 128 * i);      /* 8 */
    }
    free(cand);
    for (int o = 0; o < no; o++) for (int i = 0; i < N_LEVELS; i++) free(oc[o].lv[i]);
    return (int)keep;
}
