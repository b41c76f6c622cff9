#!/bin/bash
# Build recipe for oracle/_ref/libref_oracle.so : the REFERENCE'S OWN arithmetic
# (RANSAC, homography solve, NLLS, grid selection, warps, chips + distance-map masks, ResampleByOverlap) compiled with g++ from
# the sources where they lie under /root/reference.  Test infrastructure only.
#
#  * No reference source is copied into the repo: line ranges are extracted at
#    build time into oracle/_ref/gen/ (git-ignored) and deleted after the compile.
#  * No stand-in headers/libraries: only the reference's own files and the
#    OpenCV 2.4.0 headers vendored inside the reference tree (struct layouts).
#  * Sources are GB18030; iconv keeps line numbers (a 0x5C trail byte inside a
#    // comment would otherwise splice lines).
#  * Only MSVC-ism rewritten (mechanically, by sed):  unsigned char( e ) -> (unsigned char)( e )
#    and 'return false;' -> 'return NULL;' in CreateBitmap8U (pointer return).
# If /root/reference is absent (GPU box) the script exits 0 and keeps the prebuilt .so.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
R=/root/reference/code/MosaicingCode/mosaicing
CVI=/root/reference/code/MosaicingCode/3rdparty/opencv240/opencv/build/include
if [ ! -d "$R" ]; then echo "[build_ref] /root/reference absent: keeping prebuilt $OUT"; exit 0; fi
mkdir -p "$OUT/gen"
G="$OUT/gen"
u8() { iconv -f GB18030 -t UTF-8 "$R/$1"; }
# --- extracts (file:first-last) ------------------------------------------------
u8 matrix.h        | sed -n '14,23p'      > "$G/distance_struct.inc"      # pool::Distance
u8 matrix.h        | sed -n '67,120p'     > "$G/mat_basic.inc"            # TransposeMatrix, MulMatrix
u8 matrix.h        | sed -n '139,296p'    > "$G/inverse.inc"              # InverseMatrix
u8 matrix.h        | sed -n '332,403p'    > "$G/lls2.inc"                 # SolveLinearLeastSquare2
u8 matrix.h        | sed -n '992,1036p'   > "$G/apply.inc"                # ApplyAffineMat2, ApplyProjectMat3/9/2
u8 matrix.h        | sed -n '780,877p'    > "$G/homography.inc"           # SolveHomographyMatrix
u8 matrix.h        | sed -n '564,676p'    > "$G/affine.inc"               # SolveAffineMotion(9): dead AFFINE branch of Ransac2D, needed to compile
u8 mvMath.h        | sed -n '186,192p;209,213p' > "$G/dist.inc"           # DistanceOfTwoPoints, DistanceSquareOfTwoPoints
u8 LeastSquare.h   | sed -n '352,531p'    > "$G/nlls.inc"                 # NonlinearLeastSquareProjection2
u8 mosaicimage.h   | sed -n '24,34p;1729,2035p' > "$G/ransac2d.inc"       # TRANSFORM_TYPE, SampleIndexs, Ransac2D
u8 Bitmap.h        | sed -n '42,45p'      > "$G/projectmat.inc"           # ProjectMat
u8 Bitmap.h        | sed -n '105,128p'    > "$G/bitmapimage.inc"          # pool::BitmapImage
u8 ImageIO.cpp     | sed -n '58,76p' | sed 's/return false;/return NULL;/' > "$G/createbitmap.inc"   # CreateBitmap8U
u8 Bitmap.cpp      | sed -n '20,28p'      > "$G/zeroimage.inc"            # ZeroImage
u8 MosaicImage.cpp | sed -n '1613,1758p' | sed 's/unsigned char(/(unsigned char)(/g' > "$G/imgproj.inc"  # ImageProjectionTransform
u8 MosaicWithoutPos.cpp | sed -n '4977,5028p' > "$G/select.inc"           # SelectMatchPairs (grid)
u8 MosaicWithoutPos.h   | sed -n '224,228p'   > "$G/imagetransform.inc"   # ImageTransform
u8 MosaicWithoutPos.h   | sed -n '330,336p'   > "$G/applyproject9.inc"    # ApplyProject9
u8 MosaicWithoutPos.h   | sed -n '135,153p'   > "$G/matchpointpairs.inc"  # MatchPointPairs
# MosaicImagesRefined (float): bbox part and registration loop, without the cv* allocation lines 2245-2248
u8 MosaicWithoutPos.cpp | sed -n '2199,2244p' > "$G/mir_bbox.inc"
u8 MosaicWithoutPos.cpp | sed -n '2250,2349p' | sed 's/unsigned char(/(unsigned char)(/g' > "$G/mir_loop.inc"
# ResampleByOverlap and its quadrilateral geometry (MosaicImage.cpp:1884-2201) with the helpers it calls
u8 Bitmap.h        | sed -n '17p;54p'     > "$G/in_pi.inc"                # #define _IN ; const float pi
u8 mosaicimage.h   | sed -n '19,22p'      > "$G/rect4.inc"                # Rectangle4Points
u8 imageMath.h     | sed -n '26,88p'      > "$G/angleofpoint.inc"         # template AngleofPoint
u8 ImageMath.cpp   | sed -n '9,54p'       > "$G/angle360.inc"             # AngleofPoint360
u8 ImageMath.cpp   | sed -n '88,103p'     > "$G/lineof2.inc"              # LineOf2Points1
u8 ImageMath.cpp   | sed -n '144,176p'    > "$G/abctopolar.inc"           # ABCToPolar
u8 ImageMath.cpp   | sed -n '399,413p'    > "$G/intersec.inc"             # IntersectionPointOf2PolarLines
u8 MosaicImage.cpp | sed -n '1884,2067p'  > "$G/quad_geom.inc"            # AreaOfQuadrangle .. GetPointsInOverlapRegion
u8 MosaicImage.cpp | sed -n '2069,2201p'  > "$G/resample.inc"             # ResampleByOverlap
# FindMasksByDistMap (MosaicImage.cpp:1761-1881) without its cvZero loop (:1837-1841: the harness zeroes the masks there)
u8 MosaicImage.cpp | sed -n '1761,1836p'  > "$G/fm_head.inc"
u8 MosaicImage.cpp | sed -n '1842,1881p'  > "$G/fm_tail.inc"
# LaplacianPyramidBlending, warp stage (MosaicImage.cpp:2216-2460) without the OpenCV object lines: scale (:2216-2223), canvas
# box (:2233-2294), per-image head (:2302-2340), pixel loop (:2344-2448), bookkeeping (:2454-2456, :2459-2460).  Skipped:
# :2296-2301 (MultiBandBlender / cv::Mat vectors), :2342-2343 (cvCreateImage: the harness allocates the two IplImages),
# :2450-2453 (cv::Mat convertTo), :2458 (cvReleaseImage).
u8 MosaicImage.cpp | sed -n '2216,2223p'  > "$G/lpb_scale.inc"
u8 MosaicImage.cpp | sed -n '2233,2294p'  > "$G/lpb_bbox.inc"
u8 MosaicImage.cpp | sed -n '2302,2340p'  > "$G/lpb_head.inc"
u8 MosaicImage.cpp | sed -n '2344,2448p' | sed 's/unsigned char(/(unsigned char)(/g' > "$G/lpb_loop.inc"
u8 MosaicImage.cpp | sed -n '2454,2456p;2459,2460p' > "$G/lpb_tail.inc"
# --- compile ---------------------------------------------------------------------
g++ -std=c++11 -O2 -ffp-contract=off -fpermissive -w -fPIC -shared \
    -I "$G" -I "$R" -I "$CVI" "$HERE/ref_oracle.cpp" -o "$OUT/libref_oracle.so"
rm -rf "$G"
echo "[build_ref] built $OUT/libref_oracle.so"
