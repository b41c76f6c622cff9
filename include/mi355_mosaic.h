/* include/mi355_mosaic.h -- C ABI of libmi355mosaic.so
 *
 * MI355X-native (gfx950 / CDNA4, HIP) replacement for ONE hot path of YuhuaXu/ImageMosaicing:
 *   detect+describe -> descriptor match + selection -> RANSAC homography -> inverse-warp into the canvas.
 * Every entry point names the reference interface it replaces; paths are relative to
 * code/MosaicingCode/mosaicing/ of the reference.  INTEGRATION.md shows the binding a maintainer adds
 * on the reference side (mi355_adaptor.h re-exports the reference's own C++ signatures on top of this ABI).
 *
 * Conventions
 *  - plain pointers and sizes, no C++/torch types, no exceptions across the boundary;
 *  - return 0 on success, <0 on error (MosaicVavImages convention, MosaicWithoutPos.h:631:
 *    -1 bad arguments, -2 operation failed); mi355_last_error() gives the text;
 *    bool-like reference functions (Ransac2D) return 1/0;
 *  - images are caller-owned, BGR u8, rows padded to widthStep bytes (IplImage / BitmapImage layout);
 *  - homographies are row-major float[9]; pair homographies map image-j points onto image-i points and
 *    carry H[8] = max residual (matrix.h:866, LeastSquare.h:519), exactly like the reference;
 *  - there is NO CPU fallback: every compute entry point runs HIP kernels on the ctx's device and
 *    fails with MI355_ERR_DEVICE when no gfx950 device is usable;
 *  - a ctx is thread-safe (one internal lock + one HIP stream); results do not depend on call order.
 */
#ifndef MI355_MOSAIC_H
#define MI355_MOSAIC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK            0
#define MI355_ERR_ARG      (-1)   /* bad arguments            (MosaicVavImages: -1) */
#define MI355_ERR_FAILED   (-2)   /* operation failed         (MosaicVavImages: -2) */
#define MI355_ERR_SINGULAR (-3)   /* homography not invertible (reference: undefined behaviour) */
#define MI355_ERR_DEVICE   (-4)   /* no usable gfx950 device / HIP error */
#define MI355_ERR_NOMEM    (-5)

#define MI355_MAX_SELECTED 400    /* maxNum, MosaicWithoutPos.cpp:5146 */
#define MI355_RANSAC_BIG_MAX 65535 /* largest n mi355_ransac2d accepts: what a 16-bit draw table addresses (Ransac2D itself accepts any n, mosaicimage.h:1729-1761) */
#define MI355_DESC_DIM     128

typedef struct mi355_ctx mi355_ctx;

/* Point.h:27-47 SfPoint */
typedef struct { float x, y; int32_t id; } mi355_sfpoint;
/* MosaicWithoutPos.h:135-153 MatchPointPairs (40 bytes; the record of matchPairs.match) */
typedef struct { mi355_sfpoint ptA; int32_t ptA_i, ptA_Fixed; mi355_sfpoint ptB; int32_t ptB_i, ptB_Fixed; } mi355_match_point_pairs;
/* cv::KeyPoint, OpenCV 2.4.0 features2d.hpp (28 bytes; the record of keypoint_%d.key, MosaicWithoutPos.cpp:4691-4700) */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } mi355_keypoint;
/* cv::DMatch */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } mi355_dmatch;
/* MosaicWithoutPos.h:224-228 ImageTransform {ProjectMat h; int fixed;} */
typedef struct { float m[9]; int32_t fixed; } mi355_image_transform;

/* Result of one image pair (i,j): what GetMatchedPairsOneToAllSIFTThread appends (MosaicWithoutPos.cpp:5201-5221)
 * plus the homography Ransac2D returned.  Fixed size (9664 B) so that ranks can all-gather arrays of it. */
typedef struct {
    int32_t i, j;            /* image indices (ptA_i, ptB_i) */
    int32_t n_in;            /* inlier count; the reference accepts the pair iff n_in > min_inliers (30) */
    int32_t n_selected;      /* correspondences after grid selection (<= 396) */
    int32_t ok;              /* Ransac2D's bool; 0 for a pair that is not accepted: match_pairs skips Ransac2D's closing refinement for it */
    int32_t accepted;        /* n_in > min_inliers */
    float   H[9];            /* image j -> image i, H[8] = max residual (accepted pairs; zero otherwise) */
    int32_t _pad;            /* diagnostic: RANSAC draws that took the generic solve path (0 in the common case) */
    mi355_sfpoint a[MI355_MAX_SELECTED];   /* inliers in image i (id = keypoint index) */
    mi355_sfpoint b[MI355_MAX_SELECTED];   /* inliers in image j */
} mi355_pair_result;

/* Literals of the live path; mi355_default_params() fills the reference's values. */
typedef struct {
    int32_t nfeatures;         /* 2000   SIFT(2000,3,0.01,20)            MosaicWithoutPos.cpp:4852; 1 .. 2048, or <= 0 = cv::SIFT's keep-all
                                  (CVI/nonfree/features2d.hpp:61: every keypoint, OpenCV's generation order, <= 32768 per frame; what the
                                  reference's committed run used; such frames can be read back but the matcher takes <= 2048 keypoints) */
    int32_t n_octave_layers;   /* 3 */
    float   contrast_threshold;/* 0.01 */
    float   edge_threshold;    /* 20 */
    float   sigma;             /* 1.6    cv::SIFT default */
    int32_t max_selected;      /* 400    maxNum                           MosaicWithoutPos.cpp:5146 */
    double  select_fraction;   /* 0.3 (double literal: Min(400, 0.3*M))   MosaicWithoutPos.cpp:5147 */
    int32_t grid_x, grid_y;    /* 3,3                                     MosaicWithoutPos.cpp:5041-5042 */
    int32_t min_inliers;       /* 30     MIN_INNER_POINTS                 MosaicWithoutPos.cpp:5049 */
    float   ransac_dist;       /* 2.5    UavMatchParam.ransacDist         MosaicWithoutPos.h:74 */
    int32_t sample_times;      /* 1000                                    mosaicimage.h:1735 */
    int32_t pair_window;       /* 182    j in (i, min(N, i+182))          MosaicWithoutPos.cpp:5083 */
    float   ratio;             /* 0 = reference-compatible sort+grid selection; >0: Lowe ratio test on
                                  squared distances (d1 < ratio^2 * d2) before the grid walk (north_star option) */
} mi355_params;

void mi355_default_params(mi355_params* p);

/* ---- context -------------------------------------------------------------------------------------- */
int  mi355_create(mi355_ctx** out, const mi355_params* params /* NULL = defaults */, int device_ordinal);
void mi355_destroy(mi355_ctx* ctx);
const char* mi355_last_error(mi355_ctx* ctx);          /* ctx may be NULL: last creation error */
/* Run on a caller-provided hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = ctx-owned stream. */
int  mi355_set_stream(mi355_ctx* ctx, void* hip_stream);
int  mi355_synchronize(mi355_ctx* ctx);
/* Tunables: "sift_batch" = frames per detect+describe batch (1..8, default 8): frames handed to mi355_sift_extract_dev
 * collect until the batch is full (or until a call needs their features) and are then enqueued together, the small
 * pyramid octaves and the keypoint stages of all frames of the batch in one launch each; "sift_slots" = batch work areas
 * that may be in flight at once, each on its own stream (1..4, default 3; a work area holds sift_batch pyramids and
 * candidate lists, 3.2 GB per frame at 4000x3000); "blur_stream" = 1 (default) runs pyramid levels of >= 2048x1536
 * through the barrier-free streaming Gaussian, 0 forces the tiled kernels everywhere (same bits either way);
 * "xstream_min_w" (3000) / "xstream_min_frames" (4): octaves at least that wide, in batches of at least that many frames,
 * take the streamed extrema kernel instead of the tiled one (same candidates either way); "serial_heavy" = 1 (measurement only,
 * default 0): the pyramid + extrema phase of a batch waits for the previous batch's, so that the chip-filling kernels of different
 * batches never overlap and their event-bracketed durations are exclusive (the whole job loses ~10 %); "sift_cascade" (0..3,
 * default 3): how octaves of at least 2000 x 1500 compute their Gaussian levels -- 0: every level with its own launch; 3: the
 * first three levels in one pass, handed from wave to wave through LDS and written to HBM once, the other levels on their own
 * (+4 % end to end); 2: the other three in a second such pass; 1: all six in one pass.  The same bits in every mode (2 and 1 are
 * VALU-issue-bound and no faster end to end on MI355X); "profile_every:<class>" = n (measurement only, default 1): with
 * mi355_profile_enable only every n-th launch of that kernel class is bracketed by events (the average duration is then a sample). */
int  mi355_set_option(mi355_ctx* ctx, const char* name, int value);
void mi355_free(void* p);                               /* frees host buffers returned by this library */

/* ---- features: replaces the body of SiftExtraction_Thread, MosaicWithoutPos.cpp:4861-4881 ----------- */
/* BGR u8 host image in, keypoints + 128-D descriptors out (desc128: n x 128 floats holding the integers
 * 0..255 like OpenCV's SIFT; either output may be NULL).  Features stay device-resident under img_id for
 * mi355_match_pairs (the reference round-trips them through d:/feature_temp files instead).  With kp, desc128 and n_kp all
 * NULL nothing is waited for: the frame is copied into a staging ring in HBM (bgr may be reused on return) and joins a
 * batch like a device frame -- the fast way to feed host images. */
int  mi355_sift_extract(mi355_ctx* ctx, int img_id, const uint8_t* bgr, int w, int h, int width_step,
                        mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp);
/* Same, image already in HBM (device pointer); nothing is copied back.  With n_kp == NULL the call returns at once and
 * the frame joins the current batch (see "sift_batch"): d_bgr must stay valid and unchanged until a call that needs the
 * features (match, get_features, synchronize) has returned. */
int  mi355_sift_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int width_step, int* n_kp);
/* Fetch / install device-resident features (LoadSurfKeyPoints / WriteSurfKeyPoints counterparts,
 * MosaicWithoutPos.cpp:4682-4734).  desc128 holds integer-valued floats. */
int  mi355_get_features(mi355_ctx* ctx, int img_id, mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp);
int  mi355_set_features(mi355_ctx* ctx, int img_id, const mi355_keypoint* kp, const float* desc128, int n_kp, int w, int h);
int  mi355_drop_features(mi355_ctx* ctx, int img_id);   /* img_id < 0: all */

/* ---- match + select + RANSAC for a batch of pairs ------------------------------------------------------
 * replaces the j-loop body MosaicWithoutPos.cpp:5084-5232 (FLANN match -> sort -> SelectMatchPairs ->
 * Ransac2D -> accept).  pairs: n_pairs x {i,j} image ids with features resident.  seed plays the role of
 * time(0) in srand((unsigned)time(0)), mosaicimage.h:1777 (every pair of the batch uses it, as all pairs the
 * reference processes within one second do).  out: host array of n_pairs results. */
int  mi355_match_pairs(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed,
                       mi355_pair_result* out);
/* Same, results left in HBM at d_out (device pointer to n_pairs records), e.g. as the RCCL all-gather input. */
int  mi355_match_pairs_dev(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed,
                           mi355_pair_result* d_out);
/* Exact brute-force 1-NN / 2-NN (replaces cv::FlannBasedMatcher().match, MosaicWithoutPos.cpp:5108-5110,
 * with the exact answer): squared L2 distances as int32, sorted=1 orders the output by (distance, queryIdx)
 * like std::sort(matches) at :5111.  second_d2 may be NULL. Returns number of matches in *n_matches (= K_i). */
int  mi355_bf_match(mi355_ctx* ctx, int img_i, int img_j, int sorted, mi355_dmatch* matches, int32_t* d2, int32_t* second_d2,
                    int max_matches, int* n_matches);

/* ---- stand-alone pieces with the reference's semantics ------------------------------------------------ */
/* SelectMatchPairs (grid form), MosaicWithoutPos.cpp:4977-5028. sorted: matches already ordered. */
int  mi355_select_grid(mi355_ctx* ctx, const mi355_dmatch* sorted, int n, const float* kp1_xy, int n_kp1,
                       const float* kp2_xy, int n_kp2, int nMatch, int width, int height, int gridX, int gridY,
                       mi355_sfpoint* v1, mi355_sfpoint* v2, int* n_out);
/* Ransac2D<SfPoint>, mosaicimage.h:1729-2035.  Returns 1/0 like the bool (or <0 on error). */
int  mi355_ransac2d(mi355_ctx* ctx, const mi355_sfpoint* p1, const mi355_sfpoint* p2, int n, float dist, int sample_times,
                    uint32_t seed, mi355_sfpoint* in1, mi355_sfpoint* in2, int* n_in, float H[9]);
/* ImageProjectionTransform(BitmapImage*, BitmapImage*&, float h[9]), MosaicImage.cpp:1613-1758.
 * *dst is allocated by the library (rows padded to (w*ch+3)/4*4 like CreateBitmap8U) -> mi355_free. */
int  mi355_warp_image(mi355_ctx* ctx, const uint8_t* src, int w, int h, int ws, int ch, const float h9[9],
                      uint8_t** dst, int* dw, int* dh, int* dws);
/* CMosaicByPose::MosaicImagesRefined (float), MosaicWithoutPos.cpp:2194-2352: bbox of all images with
 * h[8]!=0, zeroed 3-channel canvas, per-image inverse bilinear warp, later images overwrite earlier ones.
 * *canvas allocated by the library -> mi355_free. */
int  mi355_mosaic_refined(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                          const float* h9s /* n x 9 */, uint8_t** canvas, int* cw, int* ch, int* cws);
/* Canvas geometry only (MosaicWithoutPos.cpp:2199-2249): size and the (dGX,dGY) shift. */
int  mi355_mosaic_layout(const int* w, const int* h, int n, const float* h9s, int* cw, int* ch, int* cws, float* dGxy);
/* Device-resident form: d_imgs[k] and d_canvas are device pointers; canvas must hold cws*ch bytes and is
 * zeroed by the call.  Only canvas rows [row0, row0+rows) are rendered (canvas stripes for multi-GPU,
 * SURVEY 8e); pass 0, ch for the whole canvas. */
/* d_imgs[k] == NULL: the caller holds no copy of image k and thereby states that these rows do not read it (owner-only frames: the pointers
 * mi355_exchange_frames returns; true of an image that lies under later ones wherever its box meets the rows, MI355_COVER_REFINED_EXACT); the
 * image is left out of the walk.  mi355_set_option("strict_frames", 1) checks the statement first (one extra pass) and names an image that is
 * read after all (MI355_ERR_ARG). */
int  mi355_mosaic_refined_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n,
                              const float* h9s, uint8_t* d_canvas, int cw, int ch, int cws, int row0, int rows);

/* LaplacianPyramidBlending warp stage (MosaicImage.cpp:2233-2460) + FindMasksByDistMap (:1761-1881):
 * per kept image a tight chip (3ch u8; the reference then converts to CV_16S), its validity mask and, with
 * find_masks!=0, the exclusive distance-map ownership masks.  h9s must carry the resScale multiplication of
 * m[0..5] (:2216-2223); keep[k] = vecAbandonInd (NULL = keep all). Outputs are library-allocated arrays of
 * n_chips buffers (free each and the arrays with mi355_free). */
typedef struct { int32_t x0, y0, w, h, img; float sx, sy; float quad[8]; } mi355_chip_info;
int  mi355_chips_and_masks(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n,
                           const float* h9s, const uint8_t* keep, int find_masks,
                           int* n_chips, mi355_chip_info** chips, uint8_t*** chip_imgs, uint8_t*** masks,
                           int* canvas_w, int* canvas_h);

/* ResampleByOverlap(pImages, n, overlapT, pImgT, vecAbandonInd), MosaicImage.cpp:2069-2201 (LaplacianPyramidBlending calls it with
 * overlapT = 0.7f, :2227-2230): keep[k] = vecAbandonInd[k] -- image k is dropped (0) when the quadrilateral it covers overlaps an
 * earlier KEPT image's by more than overlapT of its own area; image 0 and image n-1 are always kept.  Host geometry (no ctx): the
 * result is the keep[] argument of the two calls around it.  h9s must already carry the resScale multiplication (:2216-2223). */
int  mi355_resample_by_overlap(const int* w, const int* h, int n, const float* h9s, float overlapT, uint8_t* keep);

/* Multiband blend of those chips ("next" row f3 of SURVEY 8f): replaces detail::MultiBandBlender blender(false, band) --
 * prepare(Rect(0,0,canvas_w,canvas_h)), feed(chip as CV_16S, mask, corner) per chip, blend, convertTo(CV_8U)
 * (MosaicImage.cpp:2296-2299, 2451-2486; the reference passes band = 5).  chips / masks / info exactly as
 * mi355_chips_and_masks returns them (BGR u8 rows padded to 4 bytes, masks after FindMasksByDistMap).  The arithmetic of
 * OpenCV 2.4.0's blender is not available: the definition is the one in oracle/oracle_blend.c (parity unpinned).
 * *out: BGR u8 canvas, rows padded to 4 bytes, library-allocated (mi355_free). */
int  mi355_multiband_blend(mi355_ctx* ctx, const uint8_t* const* chips, const uint8_t* const* masks, const mi355_chip_info* info, int n,
                           int canvas_w, int canvas_h, int band, uint8_t** out, int* out_w, int* out_h, int* out_ws);

/* Both stages in one call, chips and masks never leaving HBM: the whole of LaplacianPyramidBlending(pImages, n, pImgT, band, ...)
 * (MosaicImage.cpp:2205-2510) except that the inputs are NOT released (the adaptor does that, :2464-2467).  Same bytes as
 * mi355_chips_and_masks(find_masks = 1) followed by mi355_multiband_blend. */
int  mi355_mosaic_blended(mi355_ctx* ctx, const uint8_t* const* imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                          const uint8_t* keep, int band, uint8_t** out, int* out_w, int* out_h, int* out_ws);

/* Device form for surveys that live in HBM (C5: 2000 frames + chips + masks + distance maps + the blender's pyramids co-resident):
 * d_imgs are DEVICE pointers (no staging copy), the finished canvas is written to the caller's DEVICE buffer d_canvas whose
 * geometry (cw, ch, cws = rows padded to 4 bytes) must be what mi355_blend_layout returns for the same arguments.  Same bytes as
 * mi355_mosaic_blended.  Returns when the work is enqueued on the ctx stream (mi355_synchronize). */
int  mi355_mosaic_blended_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                              const uint8_t* keep, int band, uint8_t* d_canvas, int cw, int ch, int cws);
/* One horizontal STRIPE of that canvas: rows row0 .. row0 + rows - 1 go to d_rows (rows x cws bytes; cw, ch, cws are still the WHOLE
 * canvas's, from mi355_blend_layout).  This is a rank's share of the reference's default compositing path (blending = 2,
 * MosaicWithoutPos.h:57-79 -> LaplacianPyramidBlending, MosaicImage.cpp:2205-2510) when the canvas is cut into G stripes like
 * mi355_mosaic_refined_dev's: the rank forms the chips that reach its rows plus the pyramids' reach (3 * 2^band rows of feed gap, the
 * REDUCE / EXPAND taps per level), FindMasksByDistMap's ownership there (MosaicImage.cpp:1842-1872: per canvas pixel) and the rows of
 * every blender level its output depends on (MosaicImage.cpp:2296-2299, 2471-2486).  Stripes put side by side are the bytes of
 * mi355_mosaic_blended_dev, whatever the cut. */
int  mi355_mosaic_blended_rows_dev(mi355_ctx* ctx, const uint8_t* const* d_imgs, const int* w, const int* h, const int* ws, int n, const float* h9s,
                                   const uint8_t* keep, int band, uint8_t* d_rows, int cw, int ch, int cws, int row0, int rows);
/* canvas size of LaplacianPyramidBlending for these transforms (MosaicImage.cpp:2233-2292; host geometry, no ctx) */
int  mi355_blend_layout(const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int* cw, int* ch, int* cws);

/* ---- callers / formats either side of the path ("next" rows f1, f2 of SURVEY 8f) ---------------------- */
/* matchPairs.match: int32 n + n x 40-byte records (WriteMatchPairs / LoadMatchPairs, MosaicWithoutPos.cpp:4736-4797) */
int  mi355_write_match_pairs(const char* path, const mi355_match_point_pairs* v, int n);
int  mi355_load_match_pairs(const char* path, mi355_match_point_pairs** v, int* n);          /* *v -> mi355_free */
/* matchPairs.txt (WriteMatchPairs_ASC2, MosaicWithoutPos.cpp:4751-4772) */
int  mi355_write_match_pairs_txt(const char* path, const mi355_match_point_pairs* v, int n);
/* tran0.txt (OutTransform, MosaicWithoutPos.cpp:2798-2818): rows for images 1..n-1: m0..m7 fixed */
int  mi355_write_transforms(const char* path, const mi355_image_transform* t, int n);
/* ImportTransform (MosaicWithoutPos.cpp:2820-2843): "n" then n x 9 floats; fixed = 1 for the first transform.  *t -> mi355_free. */
int  mi355_load_transforms(const char* path, mi355_image_transform** t, int* n);
/* Reads back what OutTransform / mi355_write_transforms wrote (rows "m0..m7 fixed" for images 1..n-1): returns n transforms with the
 * identity of image 0 in front and m8 = 1.  *t -> mi355_free. */
int  mi355_load_tran0(const char* path, mi355_image_transform** t, int* n);
/* keypoint_%d.key (WriteSurfKeyPoints, MosaicWithoutPos.cpp:4691-4700): int32 n + n x 28-byte cv::KeyPoint */
int  mi355_write_keypoints(const char* path, const mi355_keypoint* kp, int n);
int  mi355_load_keypoints(const char* path, mi355_keypoint** kp, int* n);
/* discriptor_%d.xml, the other half of WriteSurfKeyPoints / LoadSurfKeyPoints (MosaicWithoutPos.cpp:4685-4688, 4710-4711):
 * cv::FileStorage fs(path, WRITE); fs << "descriptor" << Mat(n_rows x n_cols, CV_32F).  The text follows OpenCV 2.4's XML emitter
 * (persistence.cpp: icvFloatToString -- integers as "12.", others "%.8e" --, lines of the <data> block wrapped at column 71, indent 4,
 * the closing tags on the last data line).  PARITY UNPINNED: the reference commits no such file; a reader that takes any
 * white-space-separated numbers (as cv::FileStorage does) makes files written by OpenCV readable whatever the wrapping.
 * The live path never needs these files: features stay resident in HBM.  *desc -> mi355_free. */
int  mi355_write_descriptors_xml(const char* path, const float* desc, int n_rows, int n_cols);
int  mi355_load_descriptors_xml(const char* path, float** desc, int* n_rows, int* n_cols);
/* Flatten accepted pair results into the driver's m_vecMatchPairs order (pair order, inlier order). */
int  mi355_results_to_match_pairs(const mi355_pair_result* r, int n_pairs, const int32_t* fixed_flags /* per image or NULL */,
                                  mi355_match_point_pairs** v, int* n);
/* Global affine alignment: Select_Connected_Matched_Images is the caller's; this is BundleAdjustmentSparse's
 * linear system (MosaicWithoutPos.cpp:6971-7202) solved by dense Cholesky of the normal equations.
 * Image 0 (and every image with fixed[k]!=0) is held at identity. out: n_images transforms. */
int  mi355_global_affine_align(const mi355_match_point_pairs* v, int n, int n_images, const int32_t* fixed,
                               mi355_image_transform* out);

/* Select_Connected_Matched_Images (MosaicWithoutPos.cpp:2754-2796, ClusterMatchNode :2673-2752): label[k]=1 for
 * the images of the largest group connected through match pairs, else 0 (the driver then drops the other pairs
 * and flags those images invalid, h.m[8]=0, :4512-4523, 4646-4652).  Ties -> the group holding the lowest index. */
int  mi355_select_connected(const mi355_match_point_pairs* v, int n, int n_images, int32_t* label);
/* The two driver steps above straight from the pair records of mi355_match_pairs (no m_vecMatchPairs copy: at C5 that vector holds
 * 28 M correspondences, 1.1 GB): accepted pairs with at least one inlier are the edges / the equations; with label != NULL the
 * alignment uses only the pairs whose two images carry a non-zero label (the driver drops the others, :4512-4523).  Same sums in
 * the same order as the m_vecMatchPairs forms: the transforms are the same bits. */
int  mi355_select_connected_results(const mi355_pair_result* r, int n_pairs, int n_images, int32_t* label);
int  mi355_global_affine_align_results(const mi355_pair_result* r, int n_pairs, int n_images, const int32_t* fixed,
                                       const int32_t* label, mi355_image_transform* out);

/* ---- SURF variant of the path ("next" row f4 of SURVEY 8f): GetMatchedPairsOneToAllSurf, MosaicWithoutPos.cpp:5300-5533 ----------
 * (every call site of it in the reference is commented out, :4461-4499; named in north_star).  cv::SURF's arithmetic is not
 * available: the definition is oracle/oracle_surf.c (parity unpinned).  SURF features live in their own id space of the ctx. */
/* SurfFeatureDetector detector(minHessian).detect + SurfDescriptorExtractor.compute (:5313-5335): SURF(hessianThreshold, 4 octaves,
 * 2 layers, extended 128-float descriptors, oriented).  The reference keeps every keypoint; here the max_kp strongest by Hessian response
 * are kept (max_kp <= 2^21 = the candidate list: pass that to keep all; an image with more maxima above the threshold than that fails), ordered by (response descending, octave, layer, row, column).  desc128: n x 128 floats, unit norm;
 * kp[].class_id = sign of the Laplacian.  Synchronous. */
int  mi355_surf_extract(mi355_ctx* ctx, int img_id, const uint8_t* bgr, int w, int h, int width_step, float hessian_threshold, int max_kp,
                        mi355_keypoint* kp, float* desc128, int* n_kp);
int  mi355_surf_extract_dev(mi355_ctx* ctx, int img_id, const uint8_t* d_bgr, int w, int h, int width_step, float hessian_threshold, int max_kp, int* n_kp);
int  mi355_surf_get_features(mi355_ctx* ctx, int img_id, mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp);
int  mi355_surf_drop_features(mi355_ctx* ctx, int img_id);   /* img_id < 0: all */
/* the ring schedule of that function: ext = min(15, n/2 - 1), j0 in (i, i + ext] wrapped modulo n (:5370-5377) */
int  mi355_surf_pair_schedule(int n_images, int32_t* pairs_ij, int max_pairs, int* n_pairs);
/* the j-loop body :5379-5527 for a batch of pairs: exact float 1-NN (what FlannBasedMatcher approximates, :5389-5391), sort by
 * (distance, queryIdx) (:5392), every match below matchDist with matchDist lowered by 0.05 until at most max_features remain
 * (:5400-5424; UavMatchParam: matchDist 0.5, maxFeatruesNum 200), CMosaicHarris::Ransac (:5459 -- Ransac2D's arithmetic with the
 * pool allocator, see tests/test_surf.py), accepted when more than min_inliers (18, :5306) inliers. */
int  mi355_surf_match_pairs(mi355_ctx* ctx, const int32_t* pairs_ij, int n_pairs, float ransac_dist, uint32_t seed, float match_dist, int max_features,
                            int min_inliers, mi355_pair_result* out);

/* ---- multi-GPU: one process per GPU ------------------------------------------------------- */
/* Deterministic shard of the reference's pair schedule (i strided by rank like the threads at
 * MosaicWithoutPos.cpp:5066, j in (i, min(N, i+window))): writes pairs of rank `rank` of `world`.
 * Rank r extracts the frames k mod world == r (:4861) and matches the pairs whose i it owns; j may be any frame
 * of the window, so every rank needs every frame's features before matching: mi355_allgather_features. */
int  mi355_pair_schedule(int n_images, int window, int rank, int world, int32_t* pairs_ij, int max_pairs, int* n_pairs);

/* Fixed-size feature record of one frame (what the reference keeps in keypoint_%d.key + discriptor_%d.xml,
 * MosaicWithoutPos.cpp:4682-4734): bytes [0, 57344) 2048 x cv::KeyPoint, [57344, 319488) 2048 x 128 u8 descriptors,
 * zero beyond the frame's n_kp.  The header travels separately (host-readable). */
#define MI355_FEATURE_RECORD_BYTES 319488
typedef struct { int32_t img_id /* < 0: padding record */, n_kp, w, h; } mi355_feature_header;
/* Transport-agnostic halves: pack resident features into records at d_payload (device, n x MI355_FEATURE_RECORD_BYTES; headers to
 * the HOST array hdr) / install records as resident features (as if extracted here).  Used by callers that bring their own
 * transport (MPI, the gloo CPU tests); mi355_allgather_features does both around one RCCL all-gather. */
int  mi355_pack_features_dev(mi355_ctx* ctx, const int32_t* img_ids, int n, mi355_feature_header* hdr, void* d_payload);
int  mi355_install_features_dev(mi355_ctx* ctx, const mi355_feature_header* hdr, const void* d_payload, int n);
/* Accepted pair records to the front of d_out (order kept): what the reference pushes to the driver (:5201-5227). */
int  mi355_compact_accepted_dev(mi355_ctx* ctx, const mi355_pair_result* d_in, int n, mi355_pair_result* d_out, int* n_out);

/* RCCL communicator of the ctx (librccl is bound at run time; a process that already holds one -- PyTorch -- shares it).
 * Rank 0 obtains the id and hands the 128 bytes to the other ranks by any means (the reference has none: file, socket,
 * torch.distributed store); then every rank calls mi355_comm_init.  Collective calls, like every RCCL collective. */
int  mi355_comm_unique_id(uint8_t id128[128]);
int  mi355_comm_init(mi355_ctx* ctx, const uint8_t id128[128], int rank, int world);
int  mi355_comm_destroy(mi355_ctx* ctx);
/* MI355_OK when librccl can be bound in this process (no communicator is touched): lets every rank agree on the transport
 * BEFORE any of them blocks inside ncclCommInitRank. */
int  mi355_comm_available(void);
/* what the communicator itself reports: ncclCommUserRank / ncclCommCount (both 0 ranks -> MI355_ERR_ARG without a communicator) */
int  mi355_comm_info(mi355_ctx* ctx, int* rank, int* n_ranks);
/* ncclAllGather over xGMI of the feature records of this rank's frames (img_ids, n_local <= n_max_per_rank, the same
 * n_max_per_rank on every rank): afterwards the features of every rank's frames are resident on every rank
 * (replaces the d:/feature_temp hand-off between SiftExtraction_Thread and the matcher threads, :4874-4880 / :5100-5103).
 * A rank whose own arguments / features are bad still takes part in the collective (it sends records flagged img_id == -2),
 * so that EVERY rank returns the error together instead of the others hanging in ncclAllGather. */
int  mi355_allgather_features(mi355_ctx* ctx, const int32_t* img_ids, int n_local, int n_max_per_rank);
/* The result exchange (PushMatchPairs, :10137-10145): d_local = this rank's n_local device records (mi355_match_pairs_dev);
 * flags: MI355_GATHER_ACCEPTED_ONLY sends the accepted pairs only (C4: 96 % of the window pairs do not overlap);
 * MI355_GATHER_NO_WAIT (root >= 0 only) returns on the root once the receives and the device-to-host copy are ENQUEUED: *all is complete
 * after the next mi355_synchronize -- the 1.1 GB of C5 then cross PCIe while the host runs the alignment on the moments.
 *   root < 0   ncclAllGather: every rank's host receives the records of all ranks (rank-major);
 *   root >= 0  only rank `root`'s host does -- the rank that runs the reference's unchanged driver on the inlier lists
 *              (Select_Connected_Matched_Images / BundleAdjustmentSparse, MosaicWithoutPos.cpp:4575-4591).  The other ranks ncclSend their
 *              records to it and get *all = NULL; a deployment gives them the moments (mi355_allgather_moments) for the replicated
 *              alignment that places their canvas stripes.  C5: 1.1 GB of records then cross PCIe on one rank instead of on eight.
 * *n_all = the number of records of all ranks, on every rank.  *all points into PINNED host memory owned by the ctx (grown when needed, never
 * shrunk): valid until the next mi355_allgather_results call on this ctx or mi355_destroy -- do NOT free it.  (Until round 5 this was a
 * fresh malloc per call: page faults and a staged copy held the device-to-host leg to 3.7 GB/s.) */
#define MI355_GATHER_ACCEPTED_ONLY 1
#define MI355_GATHER_NO_WAIT       2
int  mi355_allgather_results(mi355_ctx* ctx, const mi355_pair_result* d_local, int n_local, int flags, int root,
                             const mi355_pair_result** all, int* n_all);

/* What the global alignment needs of an accepted pair (round 5): the second moments of its inlier coordinates -- the sums the normal
 * equations of BundleAdjustmentSparse's system are made of (MosaicWithoutPos.cpp:6971-7202) -- instead of the 9664-byte record with its two
 * inlier lists.  ca = (xa, ya, 1) from the record's a[], cb = (xb, yb, 1) from b[]; aa = sum ca ca^T (lower triangle, row-major: 00 10 11 20
 * 21 22), ab = sum ca cb^T (row-major 3 x 3), bb = sum cb cb^T; doubles, summed in inlier order, every product and sum rounded separately:
 * the device forms the very sums mi355_global_affine_align_results forms on the host, so the transforms are the same bits either way. */
typedef struct { int32_t i, j, n_in, _pad; double aa[6], ab[9], bb[6]; } mi355_pair_moments;      /* 184 bytes */
/* one record per pair record, on the device (a wave per pair); pairs that are not accepted get n_in = 0 and zero sums */
int  mi355_pair_moments_dev(mi355_ctx* ctx, const mi355_pair_result* d_results, int n, mi355_pair_moments* d_out);
/* the same sums on the host (callers without a device, tests) */
int  mi355_pair_moments_host(const mi355_pair_result* r, int n, mi355_pair_moments* out);
/* mi355_allgather_results for callers that only align: this rank's ACCEPTED pairs -> their moments (on the device) -> ncclAllGather -> host
 * array, rank-major, the same on every rank; pinned and owned by the ctx like mi355_allgather_results' (valid until the next
 * mi355_allgather_moments call; do NOT free).  52 x fewer bytes over xGMI and PCIe than the records (C5: 117 621 accepted pairs are 1.1 GB of
 * records per rank), and the replicated host step starts from the sums instead of 28 M correspondences. */
int  mi355_allgather_moments(mi355_ctx* ctx, const mi355_pair_result* d_local, int n_local, const mi355_pair_moments** all, int* n_all);
/* Select_Connected_Matched_Images / the global alignment from the moments (entries with n_in <= 0 are skipped): the same labels and, bit
 * for bit, the same transforms as the _results forms give on the records the moments were formed from */
int  mi355_select_connected_moments(const mi355_pair_moments* m, int n, int n_images, int32_t* label);
int  mi355_global_affine_align_moments(const mi355_pair_moments* m, int n, int n_images, const int32_t* fixed, const int32_t* label,
                                       mi355_image_transform* out);

/* ---- frame ownership for the compositing phase (primary form) --------------------------------------------------------------
 * A rank uploads and holds only the frames it extracts (k mod G == rank, MosaicWithoutPos.cpp:4861).  After the (replicated) alignment
 * every rank knows every rank's canvas stripe and therefore which frames each stripe reads; a frame a stripe reads and its rank does not
 * hold is sent by its owner over xGMI.  Replaces "every rank holds all N frames" (72 GB per GPU at C5; 8 x the PCIe upload). */
/* need[k] = 1 when rendering canvas rows [row0, row0 + rows) reads frame k.  mode:
 *   MI355_COVER_REFINED        mi355_mosaic_refined_dev, by host geometry alone: every frame whose clipped canvas box meets the rows (a superset
 *                              of what is read: a frame lying entirely under later frames is in it);
 *   MI355_COVER_BLENDED        mi355_mosaic_blended_rows_dev (keep, band as there): the chips that reach the rows plus the blender pyramids' reach;
 *   MI355_COVER_REFINED_EXACT  mi355_mosaic_refined_dev, exactly: the frames that GIVE at least one pixel of the rows its sample -- the tile
 *                              kernel's own walk (descending image index, first valid sample wins, MosaicWithoutPos.cpp:2254-2348 read backwards)
 *                              with its loads and stores left out, one extra launch and a 4-bytes-per-frame copy back; at C5, where ~60 frames
 *                              cover a canvas pixel, a stripe reads 300-340 frames of the 640 whose boxes meet it.
 * Every mode runs the stripe call's own code path with the pixel work left out, so the list cannot drift from what the call dereferences. */
#define MI355_COVER_REFINED       0
#define MI355_COVER_BLENDED       1
#define MI355_COVER_REFINED_EXACT 2
int  mi355_mosaic_stripe_cover(mi355_ctx* ctx, int mode, const int* w, const int* h, int n, const float* h9s, const uint8_t* keep, int band,
                               int row0, int rows, uint8_t* need);
/* The exchange.  d_frames[k]: this rank's device pointer of frame k where it holds it (owner[k] == its rank; owner == NULL: k mod G), else
 * ignored.  need: G x n bytes, need[r * n + k] != 0 = rank r's stripe reads frame k (mi355_mosaic_stripe_cover with rank r's rows); the SAME
 * table on every rank.  For every (r, k) with need set and r != owner[k] the owner ncclSends the frame (ws[k] * h[k] bytes) and rank r ncclRecvs
 * it into storage owned by the ctx ("frame_exchange", grown when needed); the transfers are grouped by runs of 64 frames, every rank walks the
 * same table in the same order.  d_out[k]: this rank's pointer to frame k afterwards -- its own frame, the received copy, or NULL when its
 * stripe does not read it -- ready to be the d_imgs of the stripe calls.  Received copies stay valid until the next call.
 * flags: MI355_EXCHANGE_OWN_THROUGH_RCCL also routes the rank's own needed frames through ncclSend / ncclRecv to itself (a communicator
 * of one rank then exercises the whole path: tests).  bytes_recv / bytes_sent (NULL allowed): this rank's traffic.  Enqueued on the ctx stream.
 * A rank must hold (d_frames[k] != NULL) every frame it owns.  What can fail on one rank alone (an owned frame without a pointer, bad geometry,
 * no memory for the landing area) is found BEFORE any transfer is posted and its verdict travels with the cover rows (one all-gather): every
 * rank returns the error together, none is left waiting in ncclRecv. */
#define MI355_EXCHANGE_OWN_THROUGH_RCCL 1
/* MI355_EXCHANGE_NEED_IS_LOCAL: `need` holds n bytes, this rank's OWN row (what its stripe reads, e.g. from MI355_COVER_REFINED_EXACT on its
 * own device); the call all-gathers the rows of all ranks first (n bytes per rank). */
#define MI355_EXCHANGE_NEED_IS_LOCAL   2
int  mi355_exchange_frames(mi355_ctx* ctx, const uint8_t* const* d_frames, const int* h, const int* ws, int n, const int32_t* owner,
                           const uint8_t* need, int flags, const uint8_t** d_out, uint64_t* bytes_recv, uint64_t* bytes_sent);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------------------- */
/* When enabled, every launch of the named kernel class is bracketed by hipEvents on the ctx stream. */
int  mi355_profile_enable(mi355_ctx* ctx, int on);
/* SIFT stage populations of the last extracted frame: [0] DoG extrema, [1] refined points, [2] oriented keypoints,
 * [3] kept (nfeatures + ties with the last one as KeyPointsFilter::retainBest keeps them, <= 2048), [4] overflow flag */
int  mi355_last_sift_counters(mi355_ctx* ctx, int32_t out8[8]);
int  mi355_profile_reset(mi355_ctx* ctx);
int  mi355_profile_only(mi355_ctx* ctx, const char* kernel_class /* NULL or "" = every class */);
/* class: "gauss", "extrema", "orient", "describe", "match", "select", "ransac", "warp", "gray" ...
 * Synchronises the stream. total_ms/launches may be NULL. */
int  mi355_profile_get(mi355_ctx* ctx, const char* kernel_class, double* total_ms, int64_t* launches, double* alg_bytes);

/* Synthetic input (bench / tests only, never timed): BGR frame sampled from a seeded procedural terrain through the
 * affine map (u,v) = A6 (x,y,1), written straight into HBM at d_dst. */
int  mi355_synth_frame_dev(mi355_ctx* ctx, uint8_t* d_dst, int w, int h, int ws, const float A6[6], uint32_t seed, uint32_t frame_seed,
                           float gain, float noise_sigma);

#ifdef __cplusplus
}
#endif
#endif /* MI355_MOSAIC_H */
