// include/mi355_adaptor.h -- header-only C++ adaptor: the reference's OWN per-pair function signatures,
// forwarded to libmi355mosaic.so (include/mi355_mosaic.h).  A maintainer of YuhuaXu/ImageMosaicing includes
// this header instead of calling the CPU implementations; the driver code (CMosaicByPose::MosaicWithoutPose,
// MosaicWithoutPos.cpp:4430-4679) stays as it is.  See INTEGRATION.md for the exact edit.
//
// The adaptor is written against the reference's POD layouts (Point.h:27-47 SfPoint, Bitmap.h:42-45 ProjectMat,
// Bitmap.h:105-128 BitmapImage, MosaicWithoutPos.h:135-153 MatchPointPairs, :224-228 ImageTransform).  When it
// is compiled INSIDE the reference tree those types already exist: define MI355_ADAPTOR_USE_REFERENCE_TYPES
// before including it.  Stand-alone (this repo's tests) it declares layout-identical types in namespace
// mi355ref.
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "mi355_mosaic.h"
// Needs a C++11 compiler (std::call_once guards the process-wide context; the reference's VS2008 project moves to VS2012+).

#ifndef MI355_ADAPTOR_USE_REFERENCE_TYPES
namespace mi355ref {
struct SfPoint { float x, y; int id; SfPoint() {} SfPoint(float x_, float y_) : x(x_), y(y_), id(0) {} };   // Point.h:27-47
struct ProjectMat { float m[9]; };                                                                           // Bitmap.h:42-45
struct BitmapImage {                                                                                         // Bitmap.h:105-128
    unsigned char* imageData; int width, height, widthStep, nChannels;
    BitmapImage() : imageData(NULL), width(0), height(0), widthStep(0), nChannels(0) {}
};
inline void ReleaseBitmap8U(BitmapImage*& p) { if (!p) return; delete[] p->imageData; p->imageData = NULL; delete p; p = NULL; }      // ImageIO.cpp:78-93
struct MatchPointPairs { SfPoint ptA; int ptA_i, ptA_Fixed; SfPoint ptB; int ptB_i, ptB_Fixed; };            // MosaicWithoutPos.h:135-153
struct ImageTransform { ProjectMat h; int fixed; };                                                          // MosaicWithoutPos.h:224-228
// IplImage, OpenCV 2.4.0 core/types_c.h (field for field); stand-alone images own imageData through malloc
struct IplImage {
    int nSize, ID, nChannels, alphaChannel, depth; char colorModel[4], channelSeq[4];
    int dataOrder, origin, align, width, height; void* roi; void* maskROI; void* imageId; void* tileInfo;
    int imageSize; char* imageData; int widthStep; int BorderMode[4], BorderConst[4]; char* imageDataOrigin;
};
inline IplImage* cvCreateImage8U(int w, int h, int ch) {              // cvCreateImage(cvSize(w, h), 8, ch): rows aligned to 4 bytes
    IplImage* im = (IplImage*)std::calloc(1, sizeof(IplImage));
    if (!im) return NULL;
    im->nSize = (int)sizeof(IplImage); im->nChannels = ch; im->depth = 8; im->width = w; im->height = h; im->align = 4;
    im->widthStep = (w * ch + 3) & ~3; im->imageSize = im->widthStep * h;
    im->imageData = im->imageDataOrigin = (char*)std::malloc((size_t)im->imageSize > 0 ? (size_t)im->imageSize : 1);
    if (!im->imageData) { std::free(im); return NULL; }
    return im;
}
inline void cvReleaseImage(IplImage** p) { if (p && *p) { std::free((*p)->imageDataOrigin); std::free(*p); *p = NULL; } }
struct ImagePoseInfo { IplImage* pImg; int fixed; ImagePoseInfo() : pImg(NULL), fixed(0) {} };              // MosaicWithoutPos.h:283-297 (camPose omitted)
}  // namespace mi355ref
#define MI355_NS mi355ref::
#define MI355_CREATE_IMAGE_8U(w, h, ch) mi355ref::cvCreateImage8U((w), (h), (ch))
#else
#define MI355_NS
#define MI355_CREATE_IMAGE_8U(w, h, ch) cvCreateImage(cvSize((w), (h)), 8, (ch))
#endif

namespace mi355 {

static_assert(sizeof(MI355_NS SfPoint) == sizeof(mi355_sfpoint), "SfPoint layout");
static_assert(sizeof(MI355_NS MatchPointPairs) == sizeof(mi355_match_point_pairs), "MatchPointPairs layout");
static_assert(sizeof(MI355_NS ImageTransform) == sizeof(mi355_image_transform), "ImageTransform layout");

// One process-wide context per device (the reference is a single-process program).  Creation is guarded: the reference calls the
// per-pair code from up to 8 worker threads at once (MosaicWithoutPos.cpp:5246-5292); the ctx itself is thread-safe per the C ABI.
inline mi355_ctx* context(int device = 0) {
    struct Slot { std::once_flag once; mi355_ctx* ctx; Slot() : ctx(NULL) {} };
    static Slot slots[16];
    if (device < 0 || device >= 16) return NULL;
    Slot& s = slots[device];
    std::call_once(s.once, [&s, device]() { if (mi355_create(&s.ctx, NULL, device) != MI355_OK) s.ctx = NULL; });
    return s.ctx;
}

// bool Ransac2D(const vector<PointType>&, const vector<PointType>&, vector<PointType>&, vector<PointType>&,
//               float aProjectMat[9], float fRansacDist = 1, int sampleTimes = 1000)          mosaicimage.h:1729-1735
// `seed` stands for the reference's srand((unsigned)time(0)) (mosaicimage.h:1777).  Up to 65535 correspondences (the live path passes
// <= 396: maxNum, MosaicWithoutPos.cpp:5146; above 400 a second kernel keeps the work arrays in HBM, above 4096 a third the points too);
// larger inputs return false with aProjectMat zeroed, where the reference would run on any n.
inline bool Ransac2D(const std::vector<MI355_NS SfPoint>& p1, const std::vector<MI355_NS SfPoint>& p2,
                     std::vector<MI355_NS SfPoint>& in1, std::vector<MI355_NS SfPoint>& in2, float aProjectMat[9],
                     float fRansacDist = 1.0f, int sampleTimes = 1000, unsigned seed = 1) {
    in1.clear(); in2.clear();
    if (p1.empty() || p1.size() != p2.size()) return false;
    mi355_ctx* c = context();
    if (!c) return false;
    const size_t cap = p1.size() > MI355_MAX_SELECTED ? p1.size() : MI355_MAX_SELECTED;
    std::vector<mi355_sfpoint> a(cap), b(cap);
    int n_in = 0;
    const int ok = mi355_ransac2d(c, reinterpret_cast<const mi355_sfpoint*>(&p1[0]), reinterpret_cast<const mi355_sfpoint*>(&p2[0]), (int)p1.size(),
                                  fRansacDist, sampleTimes, seed, &a[0], &b[0], &n_in, aProjectMat);
    if (ok < 0) return false;
    in1.resize(n_in); in2.resize(n_in);
    if (n_in) { std::memcpy(&in1[0], &a[0], sizeof(mi355_sfpoint) * n_in); std::memcpy(&in2[0], &b[0], sizeof(mi355_sfpoint) * n_in); }
    return ok == 1;
}

// int SelectMatchPairs(const vector<DMatch>&, const vector<KeyPoint>&, const vector<KeyPoint>&, int nMatch, int width, int height,
//                      int gridX, int gridY, vector<SfPoint>&, vector<SfPoint>&)                    MosaicWithoutPos.cpp:4977-4983
// One body for both spellings of the element types: cv::DMatch / cv::KeyPoint (OpenCV 2.4.0 features2d.hpp: the reference's own call,
// MosaicWithoutPos.cpp:5146-5153, compiles unchanged against this) and the C-ABI PODs mi355_dmatch / mi355_keypoint.  Nothing of OpenCV
// is included here: the element types are template parameters and only .queryIdx/.trainIdx/.imgIdx/.distance and the point are read.
namespace detail {
inline float kp_x(const mi355_keypoint& k) { return k.x; }
inline float kp_y(const mi355_keypoint& k) { return k.y; }
template <class KP> inline float kp_x(const KP& k) { return k.pt.x; }      // cv::KeyPoint
template <class KP> inline float kp_y(const KP& k) { return k.pt.y; }
}  // namespace detail
template <class DMatchT, class KeyPointT>
inline int SelectMatchPairs(const std::vector<DMatchT>& matches, const std::vector<KeyPointT>& kp1, const std::vector<KeyPointT>& kp2,
                            int nMatch, int width, int height, int gridX, int gridY,
                            std::vector<MI355_NS SfPoint>& v1, std::vector<MI355_NS SfPoint>& v2) {
    v1.clear(); v2.clear();
    mi355_ctx* c = context();
    if (!c) return -1;
    std::vector<mi355_dmatch> dm(matches.size());
    for (size_t i = 0; i < matches.size(); i++) {
        dm[i].queryIdx = matches[i].queryIdx; dm[i].trainIdx = matches[i].trainIdx; dm[i].imgIdx = matches[i].imgIdx; dm[i].distance = matches[i].distance;
    }
    std::vector<float> xy1(kp1.size() * 2), xy2(kp2.size() * 2);
    for (size_t i = 0; i < kp1.size(); i++) { xy1[2 * i] = detail::kp_x(kp1[i]); xy1[2 * i + 1] = detail::kp_y(kp1[i]); }
    for (size_t i = 0; i < kp2.size(); i++) { xy2[2 * i] = detail::kp_x(kp2[i]); xy2[2 * i + 1] = detail::kp_y(kp2[i]); }
    std::vector<mi355_sfpoint> a(MI355_MAX_SELECTED), b(MI355_MAX_SELECTED);
    int n = 0;
    const int rc = mi355_select_grid(c, dm.empty() ? NULL : &dm[0], (int)dm.size(), xy1.empty() ? NULL : &xy1[0], (int)kp1.size(),
                                     xy2.empty() ? NULL : &xy2[0], (int)kp2.size(), nMatch, width, height, gridX, gridY, &a[0], &b[0], &n);
    if (rc != MI355_OK) return rc;
    v1.resize(n); v2.resize(n);
    if (n) { std::memcpy(&v1[0], &a[0], sizeof(mi355_sfpoint) * n); std::memcpy(&v2[0], &b[0], sizeof(mi355_sfpoint) * n); }
    return 0;
}

// int ImageProjectionTransform(BitmapImage* pImage, BitmapImage*& pResult, float h[9])             MosaicImage.cpp:1613
// pResult is allocated the way CreateBitmap8U does (ImageIO.cpp:58-76: new BitmapImage, new unsigned char[widthStep * height]), so the
// reference's callers go on releasing it with ReleaseBitmap8U (ImageIO.cpp:78-93: delete[] imageData, delete) -- stand-alone,
// mi355ref::ReleaseBitmap8U is that function.  The library's own buffer (malloc) never leaves this function.
inline int ImageProjectionTransform(MI355_NS BitmapImage* pImage, MI355_NS BitmapImage*& pResult, float h[9]) {
    if (pImage == NULL) return -1;
    mi355_ctx* c = context();
    if (!c) return -1;
    uint8_t* dst = NULL; int dw = 0, dh = 0, dws = 0;
    const int rc = mi355_warp_image(c, pImage->imageData, pImage->width, pImage->height, pImage->widthStep, pImage->nChannels, h, &dst, &dw, &dh, &dws);
    if (rc != MI355_OK) return rc;
    pResult = new MI355_NS BitmapImage();
    pResult->width = dw; pResult->height = dh; pResult->widthStep = dws; pResult->nChannels = pImage->nChannels;
    pResult->imageData = new unsigned char[(size_t)dws * dh];
    std::memcpy(pResult->imageData, dst, (size_t)dws * dh);
    mi355_free(dst);
    return 0;
}

// The per-pair loop body of GetMatchedPairsOneToAllSIFTThread (MosaicWithoutPos.cpp:5084-5232) for every pair of the
// reference's schedule, appending MatchPointPairs exactly like :5201-5221.  Features must have been extracted with
// mi355_sift_extract(ctx, image_index, ...).
inline int GetMatchedPairsOneToAllSIFT(int nImages, float ransacDist, unsigned seed, const int* fixedFlags,
                                       std::vector<MI355_NS MatchPointPairs>& vecMatchPairs, int window = 182) {
    mi355_ctx* c = context();
    if (!c) return -1;
    int n_pairs = 0;
    mi355_pair_schedule(nImages, window, 0, 1, NULL, 0, &n_pairs);
    if (n_pairs == 0) return 0;
    std::vector<int32_t> pairs((size_t)n_pairs * 2);
    mi355_pair_schedule(nImages, window, 0, 1, &pairs[0], n_pairs, &n_pairs);
    mi355_pair_result* res = (mi355_pair_result*)std::malloc(sizeof(mi355_pair_result) * (size_t)n_pairs);
    if (!res) return -1;
    int rc = mi355_match_pairs(c, &pairs[0], n_pairs, ransacDist, seed, res);
    if (rc == MI355_OK) {
        mi355_match_point_pairs* v = NULL; int n = 0;
        rc = mi355_results_to_match_pairs(res, n_pairs, fixedFlags, &v, &n);
        if (rc == MI355_OK) {
            const size_t old = vecMatchPairs.size();
            vecMatchPairs.resize(old + n);
            if (n) std::memcpy(&vecMatchPairs[old], v, sizeof(mi355_match_point_pairs) * n);
            mi355_free(v);
        }
    }
    std::free(res);
    return rc;
}

// int CMosaicByPose::GetMatchedPairsOneToAllSIFT_MultiThread()                                 MosaicWithoutPos.cpp:5244-5295
// The reference's member reads m_pImgPoses / m_nImages / m_ransacDist and fills m_vecMatchPairs / m_nSuccess through its threads
// (extraction :4832-4887, matching :5031-5241, results pushed under a mutex :10137-10145; the caller sets
// m_pImgPoses / m_nImages at :4484-4486).  This is that whole call in one: SIFT(2000,3,0.01,20) of every image (parked as batches,
// no keypoint_%d.key / discriptor_%d.xml round trip), then every pair of the window j in (i, i + 182) (:5083-5084), MatchPointPairs
// appended like :5201-5221, nSuccess = accepted pairs.  PoseT is the reference's ImagePoseInfo (.pImg and .fixed are read; the images
// must stay valid until the call returns).  `seed` stands for srand((unsigned)time(0)) (:5061).  Returns 0 like the reference, < 0 on error.
template <class PoseT>
inline int GetMatchedPairsOneToAllSIFT_MultiThread(const PoseT* pImgPoses, const int nImages, std::vector<MI355_NS MatchPointPairs>& vecMatchPairs,
                                                   int& nSuccess, float ransacDist = 2.5f, unsigned seed = 1, int window = 182) {
    nSuccess = 0;
    mi355_ctx* c = context();
    if (!c || !pImgPoses || nImages < 0) return -1;
    std::vector<int32_t> fixed(nImages > 0 ? nImages : 1, 0);
    for (int i = 0; i < nImages; i++) {
        const MI355_NS IplImage* im = pImgPoses[i].pImg;
        if (!im) return -1;
        fixed[i] = pImgPoses[i].fixed;
        // deferred form (no output pointers): the frame is staged in HBM and joins a batch; the match call below waits for the features
        const int rc = mi355_sift_extract(c, i, (const uint8_t*)im->imageData, im->width, im->height, im->widthStep, NULL, NULL, 0, NULL);
        if (rc != MI355_OK) return rc;
    }
    int n_pairs = 0;
    mi355_pair_schedule(nImages, window, 0, 1, NULL, 0, &n_pairs);
    if (n_pairs == 0) return 0;
    std::vector<int32_t> pairs((size_t)n_pairs * 2);
    mi355_pair_schedule(nImages, window, 0, 1, &pairs[0], n_pairs, &n_pairs);
    mi355_pair_result* res = (mi355_pair_result*)std::malloc(sizeof(mi355_pair_result) * (size_t)n_pairs);
    if (!res) return -1;
    int rc = mi355_match_pairs(c, &pairs[0], n_pairs, ransacDist, seed, res);
    if (rc == MI355_OK) {
        for (int p = 0; p < n_pairs; p++) nSuccess += res[p].accepted ? 1 : 0;
        mi355_match_point_pairs* v = NULL; int n = 0;
        rc = mi355_results_to_match_pairs(res, n_pairs, &fixed[0], &v, &n);
        if (rc == MI355_OK) {
            const size_t old = vecMatchPairs.size();
            vecMatchPairs.resize(old + n);
            if (n) std::memcpy(&vecMatchPairs[old], v, sizeof(mi355_match_point_pairs) * n);
            mi355_free(v);
        }
    }
    std::free(res);
    return rc;
}

// int CMosaicByPose::GetMatchedPairsOneToAllSurf(const ImagePoseInfo* pImgPoses, const int nImages, vector<MatchPointPairs>& vecMatchPairs,
//                                                int& nSuccess)                                  MosaicWithoutPos.cpp:5300-5533
// (m_minHessian, m_matchDist, m_maxFeatureNum, m_ransacDist are members there: UavMatchParam defaults 50 / 0.5 / 200 / 2.5.)
// SURF features of every image, the ring schedule, exact float matching + distance selection + RANSAC; MatchPointPairs appended
// like :5497-5517; nSuccess = 1 + accepted pairs (:5320, :5516).  `seed` stands for srand((unsigned)time(0)).
template <class PoseT>
inline int GetMatchedPairsOneToAllSurf(const PoseT* pImgPoses, const int nImages, std::vector<MI355_NS MatchPointPairs>& vecMatchPairs, int& nSuccess,
                                       int minHessian = 50, float matchDist = 0.5f, int maxFeatureNum = 200, float ransacDist = 2.5f, unsigned seed = 1) {
    nSuccess = 1;
    mi355_ctx* c = context();
    if (!c || !pImgPoses) return -1;
    std::vector<int32_t> fixed(nImages > 0 ? nImages : 1, 0);
    for (int i = 0; i < nImages; i++) {
        const MI355_NS IplImage* im = pImgPoses[i].pImg;
        if (!im) return -1;
        fixed[i] = pImgPoses[i].fixed;
        int n = 0;
        const int rc = mi355_surf_extract(c, i, (const uint8_t*)im->imageData, im->width, im->height, im->widthStep, (float)minHessian, 1 << 21 /* every keypoint, like the reference */, NULL, NULL, &n);
        if (rc != MI355_OK) return rc;
    }
    int np = 0;
    mi355_surf_pair_schedule(nImages, NULL, 0, &np);
    if (np == 0) return 0;
    std::vector<int32_t> pairs((size_t)2 * np);
    mi355_surf_pair_schedule(nImages, &pairs[0], np, &np);
    mi355_pair_result* res = (mi355_pair_result*)std::malloc(sizeof(mi355_pair_result) * (size_t)np);
    if (!res) return -1;
    int rc = mi355_surf_match_pairs(c, &pairs[0], np, ransacDist, seed, matchDist, maxFeatureNum, 18, res);
    if (rc == MI355_OK) {
        for (int p = 0; p < np; p++) nSuccess += res[p].accepted ? 1 : 0;
        mi355_match_point_pairs* v = NULL; int n = 0;
        rc = mi355_results_to_match_pairs(res, np, &fixed[0], &v, &n);
        if (rc == MI355_OK) {
            const size_t old = vecMatchPairs.size();
            vecMatchPairs.resize(old + n);
            if (n) std::memcpy(&vecMatchPairs[old], v, sizeof(mi355_match_point_pairs) * n);
            mi355_free(v);
        }
    }
    std::free(res);
    return rc;
}

// int CMosaicByPose::MosaicImagesRefined(const ImagePoseInfo* pImgPoses, const int nImages, const ImageTransform* pRectified)
//                                                                                              MosaicWithoutPos.cpp:2194-2352
// The member writes m_pMosaicResult; here it is the last argument (released first when not NULL, like a second call would leak in
// the reference).  PoseT is the reference's ImagePoseInfo (only .pImg is read).  Returns 0 / -1 / -2 like the reference.
template <class PoseT>
inline int MosaicImagesRefined(const PoseT* pImgPoses, const int nImages, const MI355_NS ImageTransform* pRectified, MI355_NS IplImage*& pMosaicResult) {
    if (NULL == pImgPoses || NULL == pRectified || nImages <= 0) return -1;
    mi355_ctx* c = context();
    if (!c) return -2;
    std::vector<const uint8_t*> imgs(nImages); std::vector<int> w(nImages), h(nImages), ws(nImages); std::vector<float> h9((size_t)9 * nImages);
    for (int n = 0; n < nImages; n++) {
        const MI355_NS IplImage* im = pImgPoses[n].pImg;
        std::memcpy(&h9[(size_t)9 * n], pRectified[n].h.m, 9 * sizeof(float));
        if (!im) { imgs[n] = NULL; w[n] = h[n] = ws[n] = 0; h9[(size_t)9 * n + 8] = 0.0f; continue; }     // no image: skipped like h.m[8] == 0 (:2256)
        imgs[n] = (const uint8_t*)im->imageData; w[n] = im->width; h[n] = im->height; ws[n] = im->widthStep;
    }
    uint8_t* canvas = NULL; int cw = 0, ch = 0, cws = 0;
    const int rc = mi355_mosaic_refined(c, &imgs[0], &w[0], &h[0], &ws[0], nImages, &h9[0], &canvas, &cw, &ch, &cws);
    if (rc != MI355_OK) return rc == MI355_ERR_ARG ? -1 : -2;
    MI355_NS IplImage* out = MI355_CREATE_IMAGE_8U(cw, ch, 3);          // :2246-2248
    if (!out) { mi355_free(canvas); return -2; }
    for (int y = 0; y < ch; y++) std::memcpy(out->imageData + (size_t)y * out->widthStep, canvas + (size_t)y * cws, (size_t)3 * cw);
    mi355_free(canvas);
    if (pMosaicResult) cvReleaseImage(&pMosaicResult);
    pMosaicResult = out;
    return 0;
}

// IplImage* LaplacianPyramidBlending(IplImage** pImages, int imagesNum, ProjectMat* pImgT, int band, float resScale)
//                                                                                              MosaicImage.cpp:2205-2510
// Same contract as the reference: pImgT[i].m[0..5] are multiplied by resScale IN PLACE (:2216-2223), images overlapping a kept
// earlier one by more than 0.7 are dropped (ResampleByOverlap, :2227-2230), EVERY input image is released and its pointer set
// to NULL (:2464-2467) -- ownership passes to this function -- and the returned image belongs to the caller (cvReleaseImage).
inline MI355_NS IplImage* LaplacianPyramidBlending(MI355_NS IplImage** pImages, int imagesNum, MI355_NS ProjectMat* pImgT, int band, float resScale) {
    if (NULL == pImages || NULL == pImgT || imagesNum <= 0) return NULL;
    mi355_ctx* c = context();
    if (!c) return NULL;
    for (int i = 0; i < imagesNum; i++) for (int j = 0; j < 6; j++) pImgT[i].m[j] *= resScale;
    std::vector<const uint8_t*> imgs(imagesNum); std::vector<int> w(imagesNum), h(imagesNum), ws(imagesNum); std::vector<float> h9((size_t)9 * imagesNum);
    for (int n = 0; n < imagesNum; n++) {
        std::memcpy(&h9[(size_t)9 * n], pImgT[n].m, 9 * sizeof(float));
        if (!pImages[n]) { imgs[n] = NULL; w[n] = h[n] = 2; ws[n] = 8; h9[(size_t)9 * n + 8] = 0.0f; continue; }
        imgs[n] = (const uint8_t*)pImages[n]->imageData; w[n] = pImages[n]->width; h[n] = pImages[n]->height; ws[n] = pImages[n]->widthStep;
    }
    std::vector<uint8_t> keep(imagesNum, 1);
    MI355_NS IplImage* result = NULL;
    if (mi355_resample_by_overlap(&w[0], &h[0], imagesNum, &h9[0], 0.7f, &keep[0]) == MI355_OK) {
        uint8_t* out = NULL; int ow = 0, oh = 0, ows = 0;
        if (mi355_mosaic_blended(c, &imgs[0], &w[0], &h[0], &ws[0], imagesNum, &h9[0], &keep[0], band, &out, &ow, &oh, &ows) == MI355_OK) {
            result = MI355_CREATE_IMAGE_8U(ow, oh, 3);
            if (result) for (int y = 0; y < oh; y++) std::memcpy(result->imageData + (size_t)y * result->widthStep, out + (size_t)y * ows, (size_t)3 * ow);
            mi355_free(out);
        }
    }
    for (int n = 0; n < imagesNum; n++) cvReleaseImage(&pImages[n]);     // :2464-2467: the sources are gone whatever happened
    return result;
}

// int CMosaicByPose::MergeImagesRefined(ImagePoseInfo* pImgPoses, const int nImages, const ImageTransform* pRectified)
//                                                                                              MosaicWithoutPos.cpp:2161-2188
// (m_scale and m_pMosaicResult are members there.)  The images are consumed: pImgPoses[i].pImg = NULL on return (:2182-2185).
template <class PoseT>
inline int MergeImagesRefined(PoseT* pImgPoses, const int nImages, const MI355_NS ImageTransform* pRectified, float m_scale, MI355_NS IplImage*& pMosaicResult) {
    if (nImages <= 1) return -2;                                         // :2164-2167
    std::vector<MI355_NS IplImage*> vecImages(nImages); std::vector<MI355_NS ProjectMat> vecHomo(nImages);
    for (int i = 0; i < nImages; i++) { vecImages[i] = pImgPoses[i].pImg; vecHomo[i] = pRectified[i].h; }
    const int band = 5;                                                  // :2179
    pMosaicResult = LaplacianPyramidBlending(&vecImages[0], nImages, &vecHomo[0], band, m_scale);
    for (int i = 0; i < nImages; i++) pImgPoses[i].pImg = NULL;           // :2182-2185
    return 0;
}

}  // namespace mi355
