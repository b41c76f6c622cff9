// oracle/ref/ref_oracle.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" harness around the REFERENCE'S OWN code, compiled from line-range
// extracts made at build time by build_ref.sh (see that file for file:line of every
// extract).  Nothing here re-implements reference arithmetic; the harness only
// marshals plain arrays in and out and pins srand(time(0)) to a caller-chosen seed.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <ctime>
#include <iostream>
#include <vector>
#include <algorithm>
using namespace std;

#include "Point.h"                       // reference header, included whole (no platform deps)
using namespace pool;
namespace pool {
#include "distance_struct.inc"           // matrix.h:14-23
#include "bitmapimage.inc"               // Bitmap.h:105-128
}
#include "projectmat.inc"                // Bitmap.h:42-45
#include "dist.inc"                      // mvMath.h:186-192,209-213
#include "apply.inc"                     // matrix.h:1003-1036
#include "mat_basic.inc"                 // matrix.h:67-120
#include "inverse.inc"                   // matrix.h:139-296
#include "lls2.inc"                      // matrix.h:332-403
#include "homography.inc"                // matrix.h:780-877
#include "affine.inc"                    // matrix.h:564-676 (dead branch, compile only)
#include "nlls.inc"                      // LeastSquare.h:352-531

static unsigned g_seed = 1;
#define time(x) ((time_t)g_seed)
#include "ransac2d.inc"                  // mosaicimage.h:24-34,1729-2035
#undef time

#include "createbitmap.inc"              // ImageIO.cpp:58-76
#include "zeroimage.inc"                 // Bitmap.cpp:20-28
#include "imgproj.inc"                   // MosaicImage.cpp:1613-1758

// OpenCV 2.4.0 struct layouts from the headers vendored in the reference tree.
#include "opencv2/core/types_c.h"        // IplImage
#include "opencv2/features2d/features2d.hpp"   // cv::KeyPoint, cv::DMatch (header-only use)
using namespace cv;
#include "select.inc"                    // MosaicWithoutPos.cpp:4977-5028
#include "imagetransform.inc"            // MosaicWithoutPos.h:224-228
#include "applyproject9.inc"             // MosaicWithoutPos.h:330-336

struct RefImagePose { IplImage* pImg; };  // harness-side holder: the extract only reads .pImg

// ---- quadrilateral geometry + ResampleByOverlap, FindMasksByDistMap, LaplacianPyramidBlending warp stage -------------------
namespace pool {
#include "in_pi.inc"                     // Bitmap.h:17 (#define _IN), :54 (const float pi)
}
#include "rect4.inc"                     // mosaicimage.h:19-22
#include "angleofpoint.inc"              // imageMath.h:26-88
#include "angle360.inc"                  // ImageMath.cpp:9-54
#include "lineof2.inc"                   // ImageMath.cpp:88-103
#include "abctopolar.inc"                // ImageMath.cpp:144-176
#include "intersec.inc"                  // ImageMath.cpp:399-413
#include "quad_geom.inc"                 // MosaicImage.cpp:1884-2067
#include "resample.inc"                  // MosaicImage.cpp:2069-2201
#include "fm_head.inc"                   // MosaicImage.cpp:1761-1836  FindMasksByDistMap: header + distance maps
    for (int n = 0; n < nImages; n++)    // :1837-1841 cvZero(pMasks[n]) -- zeroing done by the harness (no OpenCV library here)
        memset(pMasks[n]->imageData, 0, (size_t)pMasks[n]->widthStep * pMasks[n]->height);
#include "fm_tail.inc"                   // MosaicImage.cpp:1842-1881  ownership loop, frees, return

static void quiet() {
    static bool done = false;
    if (!done) { std::cout.setstate(std::ios_base::failbit); done = true; }
}

extern "C" {

int ref_inverse_matrix(const float* src, int order, float* dst, float eps) {
    return InverseMatrix(src, order, dst, eps);
}

int ref_solve_homography(const float* p1xy, const float* p2xy, int n, float* H9) {
    vector<SfPoint> a(n), b(n);
    for (int i = 0; i < n; i++) { a[i].x = p1xy[2*i]; a[i].y = p1xy[2*i+1]; a[i].id = i;
                                  b[i].x = p2xy[2*i]; b[i].y = p2xy[2*i+1]; b[i].id = i; }
    for (int i = 0; i < 9; i++) H9[i] = 0;
    return SolveHomographyMatrix(&a[0], &b[0], n, H9) ? 1 : 0;
}

int ref_nlls(const float* p1xy, const float* p2xy, int n, const float* H0, float* Hout) {
    vector<SfPoint> a(n), b(n);
    for (int i = 0; i < n; i++) { a[i].x = p1xy[2*i]; a[i].y = p1xy[2*i+1]; a[i].id = i;
                                  b[i].x = p2xy[2*i]; b[i].y = p2xy[2*i+1]; b[i].id = i; }
    float h0[9]; memcpy(h0, H0, sizeof(h0));
    return NonlinearLeastSquareProjection2(&a[0], &b[0], n, Hout, h0, 1e-10f);
}

// pts: n x {x,y,id} as float,float,int32 (12-byte SfPoint layout)
int ref_ransac2d(const void* p1, const void* p2, int n, float dist, int sample_times, unsigned seed,
                 void* in1, void* in2, int* n_in, float* H9) {
    quiet();
    g_seed = seed;
    const SfPoint* a = (const SfPoint*)p1; const SfPoint* b = (const SfPoint*)p2;
    vector<SfPoint> v1(a, a + n), v2(b, b + n), i1, i2;
    for (int i = 0; i < 9; i++) H9[i] = 0;
    bool ok = Ransac2D(v1, v2, i1, i2, H9, dist, sample_times);
    *n_in = (int)i1.size();
    if (*n_in) { memcpy(in1, &i1[0], sizeof(SfPoint) * i1.size()); memcpy(in2, &i2[0], sizeof(SfPoint) * i2.size()); }
    return ok ? 1 : 0;
}

// matches: n x {queryIdx, trainIdx} int32 pairs, already sorted; kp: xy float pairs
int ref_select_match_pairs(const int* matches, int n_matches, const float* kp1xy, int nk1,
                           const float* kp2xy, int nk2, int nMatch, int width, int height,
                           int gridX, int gridY, void* out1, void* out2, int* n_out) {
    vector<DMatch> m(n_matches);
    for (int i = 0; i < n_matches; i++) { m[i].queryIdx = matches[2*i]; m[i].trainIdx = matches[2*i+1]; m[i].imgIdx = 0; m[i].distance = (float)i; }
    vector<KeyPoint> k1(nk1), k2(nk2);
    for (int i = 0; i < nk1; i++) { k1[i].pt.x = kp1xy[2*i]; k1[i].pt.y = kp1xy[2*i+1]; }
    for (int i = 0; i < nk2; i++) { k2[i].pt.x = kp2xy[2*i]; k2[i].pt.y = kp2xy[2*i+1]; }
    vector<SfPoint> v1, v2;
    SelectMatchPairs(m, k1, k2, nMatch, width, height, gridX, gridY, v1, v2);
    *n_out = (int)v1.size();
    if (*n_out) { memcpy(out1, &v1[0], sizeof(SfPoint) * v1.size()); memcpy(out2, &v2[0], sizeof(SfPoint) * v2.size()); }
    return 0;
}

// returns a buffer allocated by the reference's CreateBitmap8U; free with ref_free_u8
int ref_image_projection_transform(unsigned char* src, int w, int h, int ws, int ch, float* h9,
                                   unsigned char** dst, int* dw, int* dh, int* dws) {
    quiet();
    BitmapImage in(src, w, h, ws, ch);
    BitmapImage* out = NULL;
    int rc = ImageProjectionTransform(&in, out, h9);
    if (out) { *dst = out->imageData; *dw = out->width; *dh = out->height; *dws = out->widthStep; delete out; }
    return rc;
}
void ref_free_u8(unsigned char* p) { delete[] p; }

// MosaicImagesRefined (float version), MWP.cpp:2194-2352: bbox part (2199-2244) + registration loop
// (2250-2349).  The three cv* allocation lines (2246-2248) are replaced by the harness filling an
// IplImage header itself (struct from the vendored types_c.h): 3 channels, widthStep=(3w+3)&~3, zeroed.
// canvas may be NULL to query the size only.
int ref_mosaic_images_refined(unsigned char** imgs, const int* ws_, const int* hs_, const int* wss_, int nImages,
                              const float* h9s, unsigned char* canvas, int* cw, int* chh, int* cws) {
    quiet();
    vector<IplImage> hdr(nImages);
    vector<RefImagePose> poses(nImages);
    vector<ImageTransform> tr(nImages);
    for (int i = 0; i < nImages; i++) {
        memset(&hdr[i], 0, sizeof(IplImage));
        hdr[i].nSize = sizeof(IplImage); hdr[i].nChannels = 3; hdr[i].depth = 8;
        hdr[i].width = ws_[i]; hdr[i].height = hs_[i]; hdr[i].widthStep = wss_[i];
        hdr[i].imageData = (char*)imgs[i];
        poses[i].pImg = &hdr[i];
        memcpy(tr[i].h.m, h9s + 9 * i, 9 * sizeof(float)); tr[i].fixed = (i == 0);
    }
    const RefImagePose* pImgPoses = &poses[0];
    const ImageTransform* pRectified = &tr[0];
    IplImage res; memset(&res, 0, sizeof(res));
    IplImage* m_pMosaicResult = &res;
#include "mir_bbox.inc"
    res.nSize = sizeof(IplImage); res.nChannels = 3; res.depth = 8;
    res.width = (int)mosaicWidth; res.height = (int)mosaicHeight; res.widthStep = ((int)mosaicWidth * 3 + 3) & ~3;
    *cw = res.width; *chh = res.height; *cws = res.widthStep;
    if (!canvas) return 0;
    res.imageData = (char*)canvas;
    memset(res.imageData, 0, (size_t)res.widthStep * res.height);
    {
#include "mir_loop.inc"
    }
    return 0;
}

// ResampleByOverlap(pImages, n, overlapT, pImgT, vecAbandonInd), MosaicImage.cpp:2069-2201: only width / height of the images are read
int ref_resample_by_overlap(const int* ws_, const int* hs_, int imagesNum, const float* h9s, float overlapT, int* keep) {
    quiet();
    vector<IplImage> hdr(imagesNum); vector<IplImage*> pv(imagesNum); vector<ProjectMat> T(imagesNum);
    for (int i = 0; i < imagesNum; i++) {
        memset(&hdr[i], 0, sizeof(IplImage));
        hdr[i].nSize = sizeof(IplImage); hdr[i].nChannels = 3; hdr[i].depth = 8; hdr[i].width = ws_[i]; hdr[i].height = hs_[i];
        pv[i] = &hdr[i];
        memcpy(T[i].m, h9s + 9 * i, 9 * sizeof(float));
    }
    vector<int> v;
    int rc = ResampleByOverlap(&pv[0], imagesNum, overlapT, &T[0], v);
    for (int i = 0; i < imagesNum; i++) keep[i] = v[i];
    return rc;
}

// LaplacianPyramidBlending's warp stage (MosaicImage.cpp:2216-2460) + FindMasksByDistMap (:2471-2472): results are parked in
// g_lpb and fetched chip by chip.  keep_in == NULL runs the reference's own ResampleByOverlap(.., 0.7f, ..) (:2227-2230).
struct LpbChip { int x0, y0, w, h, img; float quad[8]; IplImage* chip; IplImage* mask; };
static vector<LpbChip> g_lpb;
static int g_lpb_w = 0, g_lpb_h = 0;
static IplImage* harness_image(int w, int h, int ch) {      // stands where cvCreateImage(cvSize(w, h), 8, ch) stood: rows aligned to 4 bytes
    IplImage* im = (IplImage*)calloc(1, sizeof(IplImage));
    im->nSize = sizeof(IplImage); im->nChannels = ch; im->depth = 8; im->width = w; im->height = h;
    im->widthStep = (w * ch + 3) & ~3;
    im->imageData = (char*)calloc((size_t)im->widthStep * (h > 0 ? h : 1), 1);   // zeroed: pixels without a sample are left untouched by the reference
    return im;
}
static void harness_free(IplImage* im) { if (im) { free(im->imageData); free(im); } }

int ref_lpb_run(unsigned char** imgs, const int* ws_, const int* hs_, const int* wss_, int imagesNum, const float* h9s, float resScale,
                const unsigned char* keep_in, int find_masks, int* canvas_w, int* canvas_h) {
    quiet();
    for (size_t k = 0; k < g_lpb.size(); k++) { harness_free(g_lpb[k].chip); harness_free(g_lpb[k].mask); }
    g_lpb.clear();
    vector<IplImage> hdr(imagesNum); vector<IplImage*> pv(imagesNum); vector<ProjectMat> T(imagesNum);
    for (int i = 0; i < imagesNum; i++) {
        memset(&hdr[i], 0, sizeof(IplImage));
        hdr[i].nSize = sizeof(IplImage); hdr[i].nChannels = 3; hdr[i].depth = 8;
        hdr[i].width = ws_[i]; hdr[i].height = hs_[i]; hdr[i].widthStep = wss_[i]; hdr[i].imageData = (char*)imgs[i];
        pv[i] = &hdr[i];
        memcpy(T[i].m, h9s + 9 * i, 9 * sizeof(float));
    }
    IplImage** pImages = &pv[0];
    ProjectMat* pImgT = &T[0];
#include "lpb_scale.inc"                 // :2216-2223
    vector<int> vecAbandonInd;
    if (keep_in) { for (int i = 0; i < imagesNum; i++) vecAbandonInd.push_back(keep_in[i] ? 1 : 0); }
    else ResampleByOverlap(pImages, imagesNum, 0.7f, pImgT, vecAbandonInd);      // :2227-2230
#include "lpb_bbox.inc"                  // :2233-2294
    vector<IplImage*> harness_chips; vector<int> harness_img;
#include "lpb_head.inc"                  // :2302-2340  (opens the per-image loop)
        IplImage* pChipImage = harness_image(wChip, hChip, pImages[0]->nChannels);   // :2342
        IplImage* pMask = harness_image(wChip, hChip, 1);                            // :2343
#include "lpb_loop.inc"                  // :2344-2448
        harness_chips.push_back(pChipImage); harness_img.push_back(n);               // instead of the cv::Mat conversion, :2450-2453
#include "lpb_tail.inc"                  // :2454-2456, :2459-2460 (closes the loop)
    if (find_masks && !vecMask.empty())
        FindMasksByDistMap(&vecMask[0], (int)vecMask.size(), &vecRectPoints[0], &vecCorners[0], newWidth, newHeight);   // :2471-2472
    for (int k = 0; k < nValid; k++) {
        LpbChip c; c.x0 = vecCorners[k].x; c.y0 = vecCorners[k].y; c.w = harness_chips[k]->width; c.h = harness_chips[k]->height; c.img = harness_img[k];
        for (int i = 0; i < 4; i++) { c.quad[2 * i] = vecRectPoints[k].pt[i].x; c.quad[2 * i + 1] = vecRectPoints[k].pt[i].y; }
        c.chip = harness_chips[k]; c.mask = vecMask[k];
        g_lpb.push_back(c);
    }
    delete[] pBegBox; delete[] pEndBox; delete[] pQuadrangleCorners;
    g_lpb_w = newWidth; g_lpb_h = newHeight;
    *canvas_w = newWidth; *canvas_h = newHeight;
    return nValid;
}
// geometry of chip k: x0 y0 w h img (5 ints) + quad (8 floats); chip / mask rows are (w*3+3)&~3 and (w+3)&~3 bytes
int ref_lpb_chip(int k, int* geom5, float* quad8, unsigned char* chip, unsigned char* mask) {
    if (k < 0 || k >= (int)g_lpb.size()) return -1;
    const LpbChip& c = g_lpb[k];
    geom5[0] = c.x0; geom5[1] = c.y0; geom5[2] = c.w; geom5[3] = c.h; geom5[4] = c.img;
    memcpy(quad8, c.quad, sizeof(c.quad));
    if (chip) memcpy(chip, c.chip->imageData, (size_t)c.chip->widthStep * c.h);
    if (mask) memcpy(mask, c.mask->imageData, (size_t)c.mask->widthStep * c.h);
    return 0;
}

int ref_sizeof_matchpointpairs();
}  // extern "C"

#include "matchpointpairs.inc"           // MosaicWithoutPos.h:135-153
extern "C" int ref_sizeof_matchpointpairs() { return (int)sizeof(MatchPointPairs); }
