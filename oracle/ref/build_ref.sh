#!/bin/bash
# Build recipe for oracle/_ref/libref_oracle.so : the REFERENCE'S OWN arithmetic
# (RANSAC, homography solve, NLLS, grid selection, warps) compiled with g++ from
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
# --- compile ---------------------------------------------------------------------
g++ -std=c++11 -O2 -ffp-contract=off -fpermissive -w -fPIC -shared \
    -I "$G" -I "$R" -I "$CVI" "$HERE/ref_oracle.cpp" -o "$OUT/libref_oracle.so"
rm -rf "$G"
echo "[build_ref] built $OUT/libref_oracle.so"
