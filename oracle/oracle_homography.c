/* oracle/oracle_homography.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's float32 homography arithmetic:
 *   InverseMatrix                      matrix.h:147-296
 *   SolveLinearLeastSquare2            matrix.h:334-403
 *   SolveHomographyMatrix              matrix.h:783-877
 *   ApplyProjectMat2 / ApplyProjectMat3 matrix.h:1003-1036
 *   NonlinearLeastSquareProjection2    LeastSquare.h:353-531
 * Compile with -ffp-contract=off: every product and sum below is a separately
 * rounded IEEE binary32 (or binary64 where the reference uses double) operation.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* matrix.h:147-296.  Gauss-Jordan on [A|I].  Column i: pivot = FIRST not-yet-used row whose
 * |a| > eps (matrix.h:173-190); returns 0 when none.  Row normalised by true division (:195-198);
 * elimination skips rows whose entry is < eps in magnitude (:204-205) and applies
 * t += (-e) * r  as mul-then-add (:209-212).  Afterwards rows are permuted so that the row holding an
 * exact 1.0f in column r moves to position r (:242-279); the search scans rows then columns and takes
 * the first exact 1 whose column == r. */
int orc_inverse_matrix(const float* src, int order, float* dst, float eps)
{
    if (order > 13 || order < 2) return -1;
    float t[400];
    int used[13];
    int o2 = order * 2;
    memset(t, 0, sizeof(float) * (size_t)(order * o2));
    for (int i = 0; i < order; i++) {
        used[i] = 0;
        t[i * o2 + order + i] = 1.0f;
        for (int j = 0; j < order; j++) t[i * o2 + j] = src[i * order + j];
    }
    for (int i = 0; i < order; i++) {
        float ei = 0.0f; int rowI = 0;
        for (int j = 0; j < order; j++) {
            if (used[j]) continue;
            if (fabsf(t[j * o2 + i]) > eps) { used[j] = 1; ei = t[j * o2 + i]; rowI = j; break; }
        }
        if (fabsf(ei) < eps) return 0;
        for (int c = 0; c < o2; c++) t[rowI * o2 + c] = t[rowI * o2 + c] / ei;
        for (int j = 0; j < order; j++) {
            if (j == rowI) continue;
            if (fabsf(t[j * o2 + i]) < eps) continue;
            float e2 = t[j * o2 + i];
            float ne = -e2;
            for (int c = 0; c < o2; c++) {
                float prod = ne * t[rowI * o2 + c];
                t[j * o2 + c] = t[j * o2 + c] + prod;
            }
        }
    }
    for (int r = 0; r < order; r++) {
        int targetRow = -1;
        for (int i = 0; i < order && targetRow < 0; i++)
            for (int j = 0; j < order; j++)
                if (t[i * o2 + j] == 1.0f && j == r) { targetRow = i; break; }
        if (targetRow >= 0 && targetRow != r)
            for (int j = 0; j < o2; j++) { float s = t[r * o2 + j]; t[r * o2 + j] = t[targetRow * o2 + j]; t[targetRow * o2 + j] = s; }
    }
    for (int i = 0; i < order; i++)
        for (int j = 0; j < order; j++) dst[i * order + j] = t[i * o2 + order + j];
    return 1;
}

/* matrix.h:100-116 MulMatrix: acc = 0; acc += a*b over the shared dimension, ascending */
static void mul_matrix(const float* a, int r1, int c1, const float* b, int c2, float* d)
{
    for (int r = 0; r < r1; r++)
        for (int c = 0; c < c2; c++) {
            float acc = 0.0f;
            for (int k = 0; k < c1; k++) { float p = a[r * c1 + k] * b[k * c2 + c]; acc = acc + p; }
            d[r * c2 + c] = acc;
        }
}

/* matrix.h:334-403: X = ((A^T A)^-1 A^T) B with explicit transpose, InverseMatrix(eps = 1e-20f) whose
 * failure is ignored (the zero-initialised inverse is used, :357,377). colA <= 8 here. */
static void solve_lls2(const float* A, int rowA, int colA, const float* B, float* X)
{
    float* AT = (float*)malloc(sizeof(float) * (size_t)(rowA * colA));
    float* invAT = (float*)malloc(sizeof(float) * (size_t)(rowA * colA));
    for (int r = 0; r < rowA; r++) for (int c = 0; c < colA; c++) AT[c * rowA + r] = A[r * colA + c];
    float ATA[64], inv[64];
    memset(ATA, 0, sizeof(ATA)); memset(inv, 0, sizeof(inv));
    mul_matrix(AT, colA, rowA, A, colA, ATA);
    orc_inverse_matrix(ATA, colA, inv, 1e-20f);
    mul_matrix(inv, colA, colA, AT, rowA, invAT);
    mul_matrix(invAT, colA, rowA, B, 1, X);
    free(AT); free(invAT);
}

/* matrix.h:1027-1036 ApplyProjectMat2: inv = 1/(m6 x + m7 y + 1); X = (m0 x + m1 y + m2)*inv */
static void apply2(const float* M, float x, float y, float* X, float* Y)
{
    float inv = 1.0f / (M[6] * x + M[7] * y + 1.0f);
    *X = (M[0] * x + M[1] * y + M[2]) * inv;
    *Y = (M[3] * x + M[4] * y + M[5]) * inv;
}

/* matrix.h:783-877.  Rows [x2 y2 1 0 0 0 -x1*x2 -x1*y2], [0 0 0 x2 y2 1 -y1*x2 -y1*y2], rhs (x1,y1)
 * (:817-834).  H[8] := max point residual, computed in double from the float-projected point (:848-866). */
int orc_solve_homography(const orc_sfpoint* p1, const orc_sfpoint* p2, int n, float H[9])
{
    if (n < 4) return 0;
    int rowA = 2 * n;
    float* A = (float*)calloc((size_t)rowA * 8, sizeof(float));
    float* B = (float*)calloc((size_t)rowA, sizeof(float));
    for (int r = 0; r < n; r++) {
        A[2 * r * 8 + 0] = p2[r].x; A[2 * r * 8 + 1] = p2[r].y; A[2 * r * 8 + 2] = 1.0f;
        A[2 * r * 8 + 6] = (-p1[r].x) * p2[r].x; A[2 * r * 8 + 7] = (-p1[r].x) * p2[r].y;
        A[(2 * r + 1) * 8 + 3] = p2[r].x; A[(2 * r + 1) * 8 + 4] = p2[r].y; A[(2 * r + 1) * 8 + 5] = 1.0f;
        A[(2 * r + 1) * 8 + 6] = (-p1[r].y) * p2[r].x; A[(2 * r + 1) * 8 + 7] = (-p1[r].y) * p2[r].y;
        B[2 * r] = p1[r].x; B[2 * r + 1] = p1[r].y;
    }
    float X[8];
    solve_lls2(A, rowA, 8, B, X);
    for (int i = 0; i < 8; i++) H[i] = X[i];
    double emax = 0.0;
    for (int i = 0; i < n; i++) {
        float fx, fy; apply2(H, p2[i].x, p2[i].y, &fx, &fy);
        double xd = (double)fx, yd = (double)fy;
        double dx = (double)p1[i].x - xd, dy = (double)p1[i].y - yd;
        double d = sqrt(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    H[8] = (float)emax;
    free(A); free(B);
    return 1;
}

/* LeastSquare.h:353-531.  Gauss-Newton, <= 15 iterations, p1 = targets (matchsort1), p0 = sources
 * (matchsort0).  The 8x8 inverse uses the default eps 1e-6 (:451) and its failure is ignored: the
 * reference then multiplies with whatever aJTEMP2 held (uninitialised stack on the first iteration);
 * this restatement starts aJTEMP2 at zero and otherwise keeps the stale contents -- the only place where
 * reference behaviour is undefined and therefore not reproducible. motion[8] = max residual (float). */
int orc_nlls_projection2(const orc_sfpoint* p1, const orc_sfpoint* p0, int n, float motion[9], const float motion0[9], float stop)
{
    if (n < 4) return 0;
    if (!p1 || !p0) return 0;
    float w[8];
    float* J  = (float*)malloc(sizeof(float) * (size_t)n * 16);
    float* C  = (float*)malloc(sizeof(float) * (size_t)n * 2);
    float* JT = (float*)malloc(sizeof(float) * (size_t)n * 16);
    float* JL = (float*)malloc(sizeof(float) * (size_t)n * 16);
    float T1[64], T2[64], dX[8];
    memset(T2, 0, sizeof(T2));
    for (int t = 0; t < 15; t++) {
        if (t == 0) memcpy(w, motion0, sizeof(float) * 8);
        for (int i = 0; i < n; i++) {
            float x2 = p1[i].x, y2 = p1[i].y, x1 = p0[i].x, y1 = p0[i].y;
            float d  = w[6] * x1 + w[7] * y1 + 1.0f;
            float nx = w[0] * x1 + w[1] * y1 + w[2];
            float ny = w[3] * x1 + w[4] * y1 + w[5];
            float* j = J + i * 16;
            j[0] = x1 / d; j[1] = y1 / d; j[2] = 1.0f / d; j[3] = 0.0f; j[4] = 0.0f; j[5] = 0.0f;
            j[6] = ((-x1) * nx) / (d * d); j[7] = ((-y1) * nx) / (d * d);
            j[8] = 0.0f; j[9] = 0.0f; j[10] = 0.0f; j[11] = x1 / d; j[12] = y1 / d; j[13] = 1.0f / d;
            j[14] = ((-x1) * ny) / (d * d); j[15] = ((-y1) * ny) / (d * d);
            C[2 * i] = x2 - nx / d; C[2 * i + 1] = y2 - ny / d;
        }
        for (int r = 0; r < 2 * n; r++) for (int c = 0; c < 8; c++) JT[c * 2 * n + r] = J[r * 8 + c];
        mul_matrix(JT, 8, 2 * n, J, 8, T1);
        orc_inverse_matrix(T1, 8, T2, 1e-6f);
        mul_matrix(T2, 8, 8, JT, 2 * n, JL);
        mul_matrix(JL, 8, 2 * n, C, 1, dX);
        for (int i = 0; i < 8; i++) w[i] = w[i] + dX[i];
        int done = 1;
        for (int i = 0; i < 8; i++) if (!(fabsf(dX[i]) < stop)) done = 0;
        if (done) break;
    }
    for (int i = 0; i < 8; i++) motion[i] = w[i];
    float emax = 0.0f;
    for (int i = 0; i < n; i++) {
        float fx, fy; apply2(motion, p0[i].x, p0[i].y, &fx, &fy);
        float dx = p1[i].x - fx, dy = p1[i].y - fy;
        float d = sqrtf(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    motion[8] = emax;
    free(J); free(C); free(JT); free(JL);
    return 1;
}
