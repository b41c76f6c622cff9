/* oracle/oracle.h -- CPU restatement of the reference's per-pair hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or
 * executed from the product (imagemosaicing_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so,
 * and only as the checker / the timed CPU baseline.
 *
 * Parity status:
 *   - homography / NLLS / RANSAC / grid selection / warps: restated from the
 *     reference's own C++ (file:line cited at each function) and PINNED against the
 *     reference itself compiled by oracle/ref/build_ref.sh (oracle/_ref) through
 *     tests/test_oracle_vs_ref.py and the committed vectors in tests/golden/.
 *   - SIFT detect+describe and brute-force matching: the arithmetic lives in
 *     OpenCV 2.4.0 (nonfree/features2d/flann), vendored in the reference only as
 *     headers + Win32 binaries => PARITY UNPINNED; oracle_sift.c restates the
 *     published algorithm (Lowe 2004) with the reference's parameters.
 *
 * All paths cited relative to /root/reference/code/MosaicingCode/mosaicing/.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Point.h:27-47  SfPoint {float x,y; int id;} (12 bytes) */
typedef struct { float x, y; int32_t id; } orc_sfpoint;
/* MosaicWithoutPos.h:135-153  MatchPointPairs (40 bytes) */
typedef struct { orc_sfpoint ptA; int32_t ptA_i, ptA_Fixed; orc_sfpoint ptB; int32_t ptB_i, ptB_Fixed; } orc_matchpair;

/* ---- oracle_homography.c ------------------------------------------------- */
int  orc_inverse_matrix(const float* src, int order, float* dst, float eps);            /* matrix.h:147-296 */
int  orc_solve_homography(const orc_sfpoint* p1, const orc_sfpoint* p2, int n, float H[9]); /* matrix.h:783-877 */
int  orc_nlls_projection2(const orc_sfpoint* p1, const orc_sfpoint* p0, int n,
                          float motion[9], const float motion0[9], float stop);         /* LeastSquare.h:353-531 */

/* ---- oracle_ransac.c ----------------------------------------------------- */
typedef struct { int32_t r[34]; int f, b; } orc_glibc_rand;                              /* glibc TYPE_3 rand() */
void orc_srand(orc_glibc_rand* g, unsigned seed);
int  orc_rand(orc_glibc_rand* g);
/* mosaicimage.h:1729-2035; returns 1/0 like the bool */
int  orc_ransac2d(const orc_sfpoint* p1, const orc_sfpoint* p2, int n, float dist, int sample_times,
                  unsigned seed, orc_sfpoint* in1, orc_sfpoint* in2, int* n_in, float H[9]);

/* ---- oracle_select.c ----------------------------------------------------- */
/* MosaicWithoutPos.cpp:4977-5028; matches = (queryIdx,trainIdx) pairs already sorted */
int  orc_select_match_pairs(const int32_t* matches, int n_matches, const float* kp1xy, const float* kp2xy,
                            int nMatch, int width, int height, int gridX, int gridY,
                            orc_sfpoint* out1, orc_sfpoint* out2, int* n_out);
/* exact brute force 1-NN + 2-NN on integer-valued descriptors (u8 stored as u8), squared L2 */
void orc_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2,
                  int32_t* nn1_idx, int32_t* nn1_d2, int32_t* nn2_d2);
/* total order used in place of std::sort(matches) (MosaicWithoutPos.cpp:5111): (dist2, queryIdx) */
void orc_sort_matches(const int32_t* nn1_idx, const int32_t* nn1_d2, int n1, int32_t* matches_out);
/* the whole j-loop body MosaicWithoutPos.cpp:5108-5221 on precomputed features; returns n inliers (0 if rejected) */
int  orc_match_pair(const float* kp1xy, const uint8_t* d1, int n1, const float* kp2xy, const uint8_t* d2, int n2,
                    int width, int height, float ransac_dist, unsigned seed,
                    orc_sfpoint* in1, orc_sfpoint* in2, float H[9], int* n_selected);

int  orc_match_pair_ratio(const float* kp1xy, const uint8_t* d1, int n1, const float* kp2xy, const uint8_t* d2, int n2,
                          int width, int height, float ransac_dist, unsigned seed, float ratio,
                          orc_sfpoint* in1, orc_sfpoint* in2, float H[9], int* n_selected);

/* ---- oracle_warp.c ------------------------------------------------------- */
/* MosaicImage.cpp:1613-1758; *dst malloc'd (free with orc_free) */
int  orc_image_projection_transform(const uint8_t* src, int w, int h, int ws, int ch, const float h9[9],
                                    uint8_t** dst, int* dw, int* dh, int* dws);
/* MosaicWithoutPos.cpp:2194-2352 (float); canvas==NULL: size query only */
int  orc_mosaic_images_refined(const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                               const float* h9s, uint8_t* canvas, int* cw, int* ch, int* cws);
/* MosaicImage.cpp:2233-2460 warp stage of LaplacianPyramidBlending for ONE kept image, given the canvas
 * offsets; see oracle_warp.c */
typedef struct { int32_t x0, y0, w, h, img; float sx, sy; float quad[8]; } orc_chip_info;
int  orc_chip_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep,
                     int* cw, int* ch, float* dG, orc_chip_info* chips);
int  orc_chip_warp(const uint8_t* src, int w, int h, int ws, const float h9[9], const float dG[2],
                   const orc_chip_info* ci, uint8_t* chip, int chip_ws, uint8_t* mask, int mask_ws);
/* MosaicImage.cpp:1761-1881 */
int  orc_find_masks_by_distmap(uint8_t** masks, const int* mask_ws, const orc_chip_info* chips, int n, int rectW, int rectH);

/* ---- oracle_blend.c: multiband blend of the chips (MosaicImage.cpp:2296-2299, 2451-2486); PARITY UNPINNED */
void orc_pyr_down16(const int16_t* src, int w, int h, int16_t* dst);
void orc_pyr_down_f(const float* src, int w, int h, float* dst);
void orc_set_float_reduce_mode(int binary_order);   /* 0 (default): the published source's single order; 1: the loop-dependent associations of opencv_imgproc240.dll */
void orc_pyr_up16(const int16_t* src, int w, int h, int16_t* dst);
int  orc_multiband_blend(const uint8_t* const* chips, const uint8_t* const* masks, const int* cx0, const int* cy0, const int* cw, const int* chh,
                         int n, int W, int H, int band, uint8_t* out);
void orc_free(void* p);

/* ---- oracle_sift.c ------------------------------------------------------- */
/* 28-byte cv::KeyPoint layout (features2d.hpp): pt.x pt.y size angle response octave class_id */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } orc_keypoint;
/* SIFT(nfeatures,3,0.01,20,1.6) detect + compute on a BGR u8 image (MosaicWithoutPos.cpp:4852-4872).
 * desc: n x 128 u8 (OpenCV stores the same integers in a float Mat). returns n */
int  orc_sift(const uint8_t* bgr, int w, int h, int ws, int nfeatures, orc_keypoint* kp, uint8_t* desc, int max_kp);
/* 0 (default): the definition the product implements; 1: orientation / descriptor arithmetic in the order the reference's binary runs it
 * (a measuring instrument, see oracle_sift.c).  Process-wide switch: set it, call orc_sift, set it back. */
void orc_sift_set_mode(int mode);
int  orc_sift_get_mode(void);
void orc_cv_exp32f(const float* x, float* y, int n);      /* OpenCV 2.4.0 cv::exp over an array, as opencv_core240.dll runs it */
const double* orc_cv_exp_table(void);                     /* its 64-entry table 2^(k/64) * A0 */

/* ---- oracle_surf.c: the SURF variant of the path (MosaicWithoutPos.cpp:5300-5533); PARITY UNPINNED ---- */
/* SURF(hessianThreshold, 4 octaves, 2 layers, extended, oriented) detect + compute (:5313-5335): the strongest max_kp keypoints,
 * ordered by (response descending, octave, layer, row, column); desc: n x 128 floats of unit norm.  returns n */
int  orc_surf(const uint8_t* bgr, int w, int h, int ws, float hessian_threshold, orc_keypoint* kp, float* desc, int max_kp);
void orc_surf_set_mode(int mode);      /* 1: det / trace evaluated as the reference's binary does on the x87 unit (measuring instrument, oracle_surf.c) */
/* exact 1-NN in L2 on float descriptors (what FlannBasedMatcher approximates, :5389-5391); distance = sqrt(sum of squares) */
void orc_bf_match_f32(const float* d1, int n1, const float* d2, int n2, int32_t* nn_idx, float* nn_dist);
/* the distance-threshold selection :5400-5424 on matches sorted by (distance, queryIdx) (:5392) */
int  orc_select_by_distance(const int32_t* nn_idx, const float* nn_dist, int n1, const float* kp1xy, const float* kp2xy,
                            float match_dist, int max_features, orc_sfpoint* out1, orc_sfpoint* out2);

#ifdef __cplusplus
}
#endif
#endif
