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
#include <vector>
#include "mi355_mosaic.h"

#ifndef MI355_ADAPTOR_USE_REFERENCE_TYPES
namespace mi355ref {
struct SfPoint { float x, y; int id; SfPoint() {} SfPoint(float x_, float y_) : x(x_), y(y_), id(0) {} };   // Point.h:27-47
struct ProjectMat { float m[9]; };                                                                           // Bitmap.h:42-45
struct BitmapImage {                                                                                         // Bitmap.h:105-128
    unsigned char* imageData; int width, height, widthStep, nChannels;
    BitmapImage() : imageData(NULL), width(0), height(0), widthStep(0), nChannels(0) {}
};
struct MatchPointPairs { SfPoint ptA; int ptA_i, ptA_Fixed; SfPoint ptB; int ptB_i, ptB_Fixed; };            // MosaicWithoutPos.h:135-153
struct ImageTransform { ProjectMat h; int fixed; };                                                          // MosaicWithoutPos.h:224-228
}  // namespace mi355ref
#define MI355_NS mi355ref::
#else
#define MI355_NS
#endif

namespace mi355 {

static_assert(sizeof(MI355_NS SfPoint) == sizeof(mi355_sfpoint), "SfPoint layout");
static_assert(sizeof(MI355_NS MatchPointPairs) == sizeof(mi355_match_point_pairs), "MatchPointPairs layout");
static_assert(sizeof(MI355_NS ImageTransform) == sizeof(mi355_image_transform), "ImageTransform layout");

// One process-wide context per device (the reference is a single-process program); thread-safe per the C ABI.
inline mi355_ctx* context(int device = 0) {
    static mi355_ctx* ctx[16] = {0};
    if (device < 0 || device >= 16) return NULL;
    if (!ctx[device]) { if (mi355_create(&ctx[device], NULL, device) != MI355_OK) ctx[device] = NULL; }
    return ctx[device];
}

// bool Ransac2D(const vector<PointType>&, const vector<PointType>&, vector<PointType>&, vector<PointType>&,
//               float aProjectMat[9], float fRansacDist = 1, int sampleTimes = 1000)          mosaicimage.h:1729-1735
// `seed` stands for the reference's srand((unsigned)time(0)) (mosaicimage.h:1777).
inline bool Ransac2D(const std::vector<MI355_NS SfPoint>& p1, const std::vector<MI355_NS SfPoint>& p2,
                     std::vector<MI355_NS SfPoint>& in1, std::vector<MI355_NS SfPoint>& in2, float aProjectMat[9],
                     float fRansacDist = 1.0f, int sampleTimes = 1000, unsigned seed = 1) {
    in1.clear(); in2.clear();
    if (p1.empty() || p1.size() != p2.size()) return false;
    mi355_ctx* c = context();
    if (!c) return false;
    std::vector<mi355_sfpoint> a(MI355_MAX_SELECTED), b(MI355_MAX_SELECTED);
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
// DMatch / KeyPoint are passed as the C-ABI PODs (identical field layout to cv::DMatch / cv::KeyPoint 2.4.0).
inline int SelectMatchPairs(const std::vector<mi355_dmatch>& matches, const std::vector<mi355_keypoint>& kp1, const std::vector<mi355_keypoint>& kp2,
                            int nMatch, int width, int height, int gridX, int gridY,
                            std::vector<MI355_NS SfPoint>& v1, std::vector<MI355_NS SfPoint>& v2) {
    v1.clear(); v2.clear();
    mi355_ctx* c = context();
    if (!c) return -1;
    std::vector<float> xy1(kp1.size() * 2), xy2(kp2.size() * 2);
    for (size_t i = 0; i < kp1.size(); i++) { xy1[2 * i] = kp1[i].x; xy1[2 * i + 1] = kp1[i].y; }
    for (size_t i = 0; i < kp2.size(); i++) { xy2[2 * i] = kp2[i].x; xy2[2 * i + 1] = kp2[i].y; }
    std::vector<mi355_sfpoint> a(MI355_MAX_SELECTED), b(MI355_MAX_SELECTED);
    int n = 0;
    const int rc = mi355_select_grid(c, matches.empty() ? NULL : &matches[0], (int)matches.size(), xy1.empty() ? NULL : &xy1[0], (int)kp1.size(),
                                     xy2.empty() ? NULL : &xy2[0], (int)kp2.size(), nMatch, width, height, gridX, gridY, &a[0], &b[0], &n);
    if (rc != MI355_OK) return rc;
    v1.resize(n); v2.resize(n);
    if (n) { std::memcpy(&v1[0], &a[0], sizeof(mi355_sfpoint) * n); std::memcpy(&v2[0], &b[0], sizeof(mi355_sfpoint) * n); }
    return 0;
}

// int ImageProjectionTransform(BitmapImage* pImage, BitmapImage*& pResult, float h[9])             MosaicImage.cpp:1613
// pResult->imageData is malloc'd by the library: release with mi355_free (the reference uses ReleaseBitmap8U).
inline int ImageProjectionTransform(MI355_NS BitmapImage* pImage, MI355_NS BitmapImage*& pResult, float h[9]) {
    if (pImage == NULL) return -1;
    mi355_ctx* c = context();
    if (!c) return -1;
    uint8_t* dst = NULL; int dw = 0, dh = 0, dws = 0;
    const int rc = mi355_warp_image(c, pImage->imageData, pImage->width, pImage->height, pImage->widthStep, pImage->nChannels, h, &dst, &dw, &dh, &dws);
    if (rc != MI355_OK) return rc;
    pResult = new MI355_NS BitmapImage();
    pResult->imageData = dst; pResult->width = dw; pResult->height = dh; pResult->widthStep = dws; pResult->nChannels = pImage->nChannels;
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

}  // namespace mi355
