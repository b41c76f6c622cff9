// csrc/hmath.h -- float32 homography arithmetic with the reference's operation order, usable from host
// and device code.  The translation units including this header are compiled with -ffp-contract=off and
// HIP's default correctly-rounded fp32 division / sqrt, so every operation below is one IEEE rounding on
// both the x86 host and gfx950: results are bit-identical to the reference compiled with g++ (SSE2).
//
// Reference semantics reproduced here (paths relative to code/MosaicingCode/mosaicing/):
//   InverseMatrix                       matrix.h:147-296   (first-pivot Gauss-Jordan + exact-1 row permutation)
//   MulMatrix                           matrix.h:93-120    (acc = 0; acc += a*b, ascending k)
//   SolveLinearLeastSquare2             matrix.h:334-403   (explicit A^T, (A^T A)^-1 A^T b, eps 1e-20)
//   SolveHomographyMatrix               matrix.h:783-877   (H[8] = max residual, double bookkeeping)
//   NonlinearLeastSquareProjection2     LeastSquare.h:353-531
//   ApplyProjectMat2 / 3 / 9            matrix.h:1003-1036 ; ApplyProject9  MosaicWithoutPos.h:331-336
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#define HD __host__ __device__ __forceinline__

namespace hm {

// reciprocal-multiply form with implicit m8 = 1 (ApplyProjectMat2)
HD void apply_recip1(const float* M, float x, float y, float& X, float& Y) {
    float inv = 1.0f / (M[6] * x + M[7] * y + 1.0f);
    X = (M[0] * x + M[1] * y + M[2]) * inv;
    Y = (M[3] * x + M[4] * y + M[5]) * inv;
}
// true-division form with implicit m8 = 1 (ApplyProjectMat3)
HD void apply_div1(const float* M, float x, float y, float& X, float& Y) {
    X = (M[0] * x + M[1] * y + M[2]) / (M[6] * x + M[7] * y + 1.0f);
    Y = (M[3] * x + M[4] * y + M[5]) / (M[6] * x + M[7] * y + 1.0f);
}
// two true divisions with explicit m8 (ApplyProject9 and the inline spellings in the warps)
HD void apply_div9(const float* M, float x, float y, float& X, float& Y) {
    X = (M[0] * x + M[1] * y + M[2]) / (M[6] * x + M[7] * y + M[8]);
    Y = (M[3] * x + M[4] * y + M[5]) / (M[6] * x + M[7] * y + M[8]);
}
// one reciprocal with explicit m8 (ApplyProjectMat9)
HD void apply_recip9(const float* M, float x, float y, float& X, float& Y) {
    float inv = 1.0f / (M[6] * x + M[7] * y + M[8]);
    X = (M[0] * x + M[1] * y + M[2]) * inv;
    Y = (M[3] * x + M[4] * y + M[5]) * inv;
}

// InverseMatrix.  t: caller scratch of 2*order*order floats.  Returns 1 ok, 0 no pivot, -1 bad order.
template <int MAXO>
HD int inverse_matrix(const float* src, int order, float* dst, float eps, float* t) {
    if (order > 13 || order < 2 || order > MAXO) return -1;
    const int o2 = order * 2;
    bool used[MAXO];
    for (int i = 0; i < order * o2; i++) t[i] = 0.0f;
    for (int i = 0; i < order; i++) {
        used[i] = false;
        t[i * o2 + order + i] = 1.0f;
        for (int j = 0; j < order; j++) t[i * o2 + j] = src[i * order + j];
    }
    for (int i = 0; i < order; i++) {
        float ei = 0.0f;
        int rowI = 0;
        for (int j = 0; j < order; j++) {
            if (used[j]) continue;
            if (fabsf(t[j * o2 + i]) > eps) { used[j] = true; ei = t[j * o2 + i]; rowI = j; break; }
        }
        if (fabsf(ei) < eps) return 0;
        for (int c = 0; c < o2; c++) t[rowI * o2 + c] = t[rowI * o2 + c] / ei;
        for (int j = 0; j < order; j++) {
            if (j == rowI) continue;
            if (fabsf(t[j * o2 + i]) < eps) continue;
            float ne = -t[j * o2 + i];
            for (int c = 0; c < o2; c++) {
                float prod = ne * t[rowI * o2 + c];
                t[j * o2 + c] = t[j * o2 + c] + prod;
            }
        }
    }
    for (int r = 0; r < order; r++) {
        int target = -1;
        for (int i = 0; i < order; i++)
            if (t[i * o2 + r] == 1.0f) { target = i; break; }
        if (target >= 0 && target != r)
            for (int j = 0; j < o2; j++) { float s = t[r * o2 + j]; t[r * o2 + j] = t[target * o2 + j]; t[target * o2 + j] = s; }
    }
    for (int i = 0; i < order; i++)
        for (int j = 0; j < order; j++) dst[i * order + j] = t[i * o2 + order + j];
    return 1;
}

// 4-point SolveHomographyMatrix: p = {x1,y1,x2,y2} x 4 (1 = target image i, 2 = source image j).
// scratch: >= 64+64+128+64 = 320 floats.
HD void solve_h4(const float* p, float* H, float* scratch) {
    float* A = scratch;            // 8x8
    float* ATA = scratch + 64;     // 8x8
    float* t = scratch + 128;      // 8x16
    float* inv = scratch + 256;    // 8x8
    float B[8];
    for (int i = 0; i < 64; i++) A[i] = 0.0f;
    for (int r = 0; r < 4; r++) {
        float x1 = p[4 * r], y1 = p[4 * r + 1], x2 = p[4 * r + 2], y2 = p[4 * r + 3];
        float* a0 = A + 2 * r * 8;
        float* a1 = A + (2 * r + 1) * 8;
        a0[0] = x2; a0[1] = y2; a0[2] = 1.0f; a0[6] = (-x1) * x2; a0[7] = (-x1) * y2;
        a1[3] = x2; a1[4] = y2; a1[5] = 1.0f; a1[6] = (-y1) * x2; a1[7] = (-y1) * y2;
        B[2 * r] = x1; B[2 * r + 1] = y1;
    }
    // ATA = A^T A
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            float acc = 0.0f;
            for (int k = 0; k < 8; k++) { float pr = A[k * 8 + r] * A[k * 8 + c]; acc = acc + pr; }
            ATA[r * 8 + c] = acc;
        }
    for (int i = 0; i < 64; i++) inv[i] = 0.0f;
    inverse_matrix<8>(ATA, 8, inv, 1e-20f, t);          // failure ignored like matrix.h:377 (inv stays 0)
    // invAT = inv * A^T (8x8), reuse ATA storage
    float* invAT = ATA;
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            float acc = 0.0f;
            for (int k = 0; k < 8; k++) { float pr = inv[r * 8 + k] * A[c * 8 + k]; acc = acc + pr; }
            invAT[r * 8 + c] = acc;
        }
    for (int r = 0; r < 8; r++) {
        float acc = 0.0f;
        for (int k = 0; k < 8; k++) { float pr = invAT[r * 8 + k] * B[k]; acc = acc + pr; }
        H[r] = acc;
    }
    double emax = 0.0;
    for (int i = 0; i < 4; i++) {
        float fx, fy;
        apply_recip1(H, p[4 * i + 2], p[4 * i + 3], fx, fy);
        double dx = (double)p[4 * i] - (double)fx, dy = (double)p[4 * i + 1] - (double)fy;
        double d = sqrt(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    H[8] = (float)emax;
}

// 4-point NonlinearLeastSquareProjection2 (stop 1e-10f). scratch: >= 320 floats.
HD void nlls4(const float* p, const float* H0, float* Hout, float* scratch) {
    float* J = scratch;            // 8x8 (2N x 8 with N = 4)
    float* T1 = scratch + 64;      // J^T J
    float* t = scratch + 128;      // 8x16
    float* T2 = scratch + 256;     // inverse (stale contents survive a failed inversion, zero initially)
    float w[8], C[8], dX[8];
    for (int i = 0; i < 64; i++) T2[i] = 0.0f;
    for (int i = 0; i < 8; i++) w[i] = H0[i];
    for (int it = 0; it < 15; it++) {
        for (int i = 0; i < 4; i++) {
            float x2 = p[4 * i], y2 = p[4 * i + 1], x1 = p[4 * i + 2], y1 = p[4 * i + 3];
            float d = w[6] * x1 + w[7] * y1 + 1.0f;
            float nx = w[0] * x1 + w[1] * y1 + w[2];
            float ny = w[3] * x1 + w[4] * y1 + w[5];
            float* j = J + i * 16;
            j[0] = x1 / d; j[1] = y1 / d; j[2] = 1.0f / d; j[3] = 0.0f; j[4] = 0.0f; j[5] = 0.0f;
            j[6] = ((-x1) * nx) / (d * d); j[7] = ((-y1) * nx) / (d * d);
            j[8] = 0.0f; j[9] = 0.0f; j[10] = 0.0f; j[11] = x1 / d; j[12] = y1 / d; j[13] = 1.0f / d;
            j[14] = ((-x1) * ny) / (d * d); j[15] = ((-y1) * ny) / (d * d);
            C[2 * i] = x2 - nx / d; C[2 * i + 1] = y2 - ny / d;
        }
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) {
                float acc = 0.0f;
                for (int k = 0; k < 8; k++) { float pr = J[k * 8 + r] * J[k * 8 + c]; acc = acc + pr; }
                T1[r * 8 + c] = acc;
            }
        inverse_matrix<8>(T1, 8, T2, 1e-6f, t);
        // JL = T2 * J^T (8x8) into T1, then dX = JL * C
        for (int r = 0; r < 8; r++)
            for (int c = 0; c < 8; c++) {
                float acc = 0.0f;
                for (int k = 0; k < 8; k++) { float pr = T2[r * 8 + k] * J[c * 8 + k]; acc = acc + pr; }
                T1[r * 8 + c] = acc;
            }
        bool done = true;
        for (int r = 0; r < 8; r++) {
            float acc = 0.0f;
            for (int k = 0; k < 8; k++) { float pr = T1[r * 8 + k] * C[k]; acc = acc + pr; }
            dX[r] = acc;
            w[r] = w[r] + acc;
            if (!(fabsf(acc) < 1e-10f)) done = false;
        }
        if (done) break;
    }
    for (int i = 0; i < 8; i++) Hout[i] = w[i];
    float emax = 0.0f;
    for (int i = 0; i < 4; i++) {
        float fx, fy;
        apply_recip1(Hout, p[4 * i + 2], p[4 * i + 3], fx, fy);
        float dx = p[4 * i] - fx, dy = p[4 * i + 1] - fy;
        float d = sqrtf(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    Hout[8] = emax;
}

// ---- register-resident fast path of the 4-point solve / NLLS ---------------------------------------------------------------
// ---- the 4-point systems exploit their structural zeros ------------------------------------------------------------------
// The design matrix of a 4-point solve and the Jacobian of its Gauss-Newton polish have the same fixed pattern
//     even rows [a b c 0 0 0 g h]      odd rows [0 0 0 d e f g h]
// so that J^T J = [[P 0 u] [0 Q v] [u^T v^T s]] (3 + 3 + 2) and InverseMatrix meets exact zeros in known places: it skips the
// rows whose entry in the pivot column is zero (|e| < eps, matrix.h:231) and, where it does operate on a zero, produces a zero
// again (0 / e, 0 + e * 0) or the first value of an entry (0 + x = x).  The plan below is that elimination run symbolically at
// compile time; the code generated from it performs exactly the reference's operations whose operands are not structural zeros,
// in the reference's order: 46 divisions, 176 multiply-adds, 56 first values instead of 128 / 896 / 0, and J^T J from 120
// products (upper triangle, mirrored: a*b = b*a) instead of 512, (J^T J)^-1 J^T from 320 instead of 512.
// A dropped operation is x + (+-0) = x, (+0) + (+-0) = +0 (an accumulator that starts at +0 never turns -0 by adding zeros),
// or works on a column that is never read again, PROVIDED no operand is infinite / NaN (0 * inf = NaN in the reference), no
// first value underflows to zero (then the sign of the zero it replaces would show), every pivot is the diagonal one and every
// expected row update really happens (|e| >= eps).  All of that is checked; a draw that fails a check is redone by the generic
// routines (caller), so the result is always the reference's.
// ---- exact division with a shared reciprocal (device) ---------------------------------------------------------------------
// hipcc expands a correctly rounded a / b into v_div_scale x2, v_rcp, two Newton fmas for the reciprocal, q = a r and two
// residual corrections q += (a - b q) r, v_div_fmas, v_div_fixup: 11 instructions, one of them quarter rate, and nothing is
// shared between divisions by the same b.  v_div_scale changes its operands only when b is denormal or above 2^126, a is below
// 2^-103, or a / b is denormal or reaches 2^96; v_div_fmas then is a plain fma, and v_div_fixup changes the result only for zero,
// infinite or NaN operands.  Outside those cases the same roundings are reproduced by the core sequence alone: the reciprocal (3
// instructions) once per divisor, 5 instructions per quotient.  The guard proves it per Gauss-Newton step: every divisor in
// [2^-30, 2^62], every quotient in [2^-70, 2^38] (so every dividend in [2^-100, 2^100]: no residual a - b q is denormal), the sum of
// the |q| catches infinite and NaN dividends.  A step whose guard fails is redone with true divisions (caller).
struct DivGuard { float s, mn; bool bok; };
HD void guard_init(DivGuard& g) { g.s = 0.0f; g.mn = 1.0f; g.bok = true; }
HD bool guard_ok(const DivGuard& g) { return g.bok && g.mn >= 0x1p-70f && g.s <= 0x1p38f; }
#if defined(__HIP_DEVICE_COMPILE__)
HD float rcp_nr(float b, DivGuard& g) {
    g.bok = g.bok && fabsf(b) >= 0x1p-30f && fabsf(b) <= 0x1p62f;
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return r;
}
HD float div_nr(float a, float b, float r, DivGuard& g) {
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    g.s = g.s + fabsf(q); g.mn = fminf(g.mn, fabsf(q));
    return q;
}
#else
HD float rcp_nr(float b, DivGuard&) { return b; }
HD float div_nr(float a, float b, float, DivGuard&) { return a / b; }
#endif

namespace sp {
constexpr bool anz(int r, int c) { return (r & 1) ? (c >= 3) : (c <= 2 || c >= 6); }
struct Plan {
    bool m[8][8];                 // J^T J entry is not a structural zero
    bool div[8][16];              // step i divides t[i][c] by the pivot
    bool upd[8][8];               // step i updates row j
    unsigned char op[8][8][16];   // step i, row j, column c: 0 nothing, 1 t[j][c] += ne * t[i][c], 2 t[j][c] = ne * t[i][c]
};
constexpr Plan make_plan() {
    Plan P{};
    bool nz[8][16] = {};
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            bool any = false;
            for (int k = 0; k < 8; k++) any = any || (anz(k, r) && anz(k, c));
            P.m[r][c] = any; nz[r][c] = any; nz[r][8 + c] = (r == c);
        }
    for (int i = 0; i < 8; i++) {
        for (int c = i + 1; c < 16; c++) P.div[i][c] = nz[i][c];
        for (int j = 0; j < 8; j++) {
            if (j == i || !nz[j][i]) continue;
            P.upd[i][j] = true;
            for (int c = i + 1; c < 16; c++)
                if (nz[i][c]) { P.op[i][j][c] = nz[j][c] ? 1 : 2; nz[j][c] = true; }
        }
        for (int j = 0; j < 8; j++) nz[j][i] = false;          // column i is never read again
    }
    return P;
}
inline constexpr Plan PLAN = make_plan();
template <int I, int N, class F> HD void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
constexpr float FMAXV = 3.402823466e+38f;
}  // namespace sp

// J^T J of a matrix with the pattern above (MulMatrix order: ascending k); false when an entry of A or of the product is not finite
HD bool jtj_sparse(const float* A, float* M) {
    float s = 0.0f;
    sp::sfor<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        sp::sfor<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (sp::anz(r, c)) s = s + fabsf(A[r * 8 + c]);
        });
    });
    bool ok = s <= sp::FMAXV;
    float sm = 0.0f;
    sp::sfor<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        sp::sfor<r, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            float acc = 0.0f;
            if constexpr (sp::PLAN.m[r][c]) {
                sp::sfor<0, 8>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (sp::anz(k, r) && sp::anz(k, c)) { const float pr = A[k * 8 + r] * A[k * 8 + c]; acc = acc + pr; }
                });
                sm = sm + fabsf(acc);
            }
            M[r * 8 + c] = acc; M[c * 8 + r] = acc;
        });
    });
    return ok && sm <= sp::FMAXV;
}

// InverseMatrix of such a J^T J (see above).  Returns 0 = inverted (dst written), 1 = the reference's routine fails here (no
// unused row has an entry above eps in some pivot column, matrix.h:206-222: dst is left as it was -- duplicated points among the
// four make this common), 2 = not the fast case (a pivot below the diagonal, a zero or non-finite entry in the result): use the
// generic routine.
// Structural zeros are kept as +0 and never operated on; a row update the reference skips because its pivot-column entry is
// tiny (|e| < eps, data dependent) is skipped here too -- the entries it would have given their first value then stay +0 until
// a later planned operation (0 + x = x) reaches them.  The reference's zeros may be -0 where these are +0, which can only show
// in an entry of the inverse that is itself zero: checked at the end.
template <bool FAST = false>
HD int inverse8_sparse(const float* src, float* dst, float eps) {
    float t[8][16];
    DivGuard g; guard_init(g);
    bool ok_at_fail = true;
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sp::sfor<0, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < 8) t[i][j] = sp::PLAN.m[i][j] ? src[i * 8 + j] : 0.0f;
            else t[i][j] = (j - 8 == i) ? 1.0f : 0.0f;
        });
    });
    int state = 0;
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const float ei = t[i][i];
        if (state == 0 && !(fabsf(ei) > eps)) {          // the pivot search goes on below the diagonal
            bool other = false;
            sp::sfor<i + 1, 8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (sp::PLAN.upd[i][j]) other = other || (fabsf(t[j][i]) > eps);
            });
            state = other ? 2 : 1;
            if constexpr (FAST) ok_at_fail = guard_ok(g);       // the verdict rests on the quotients so far; what follows is never used
        }
        if constexpr (FAST) {
            const float ri = rcp_nr(ei, g);
            sp::sfor<i + 1, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (sp::PLAN.div[i][c]) t[i][c] = div_nr(t[i][c], ei, ri, g);
            });
        } else {
            sp::sfor<i + 1, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (sp::PLAN.div[i][c]) t[i][c] = t[i][c] / ei;
            });
        }
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (sp::PLAN.upd[i][j]) {
                const float e2 = t[j][i];
                if (!(fabsf(e2) < eps)) {
                    const float ne = -e2;
                    sp::sfor<i + 1, 16>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if constexpr (sp::PLAN.op[i][j][c] != 0) { const float prod = ne * t[i][c]; t[j][c] = t[j][c] + prod; }
                    });
                }
            }
        });
    });
    if constexpr (FAST) { if (!(state != 0 ? ok_at_fail : guard_ok(g))) return 3; }      // 3 = a division left the guarded range: redo exactly
    if (state != 0) return state;
    float s = 0.0f, mn = sp::FMAXV;
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float v = t[i][8 + j];
            s = s + fabsf(v); mn = fminf(mn, fabsf(v));
        });
    });
    if (!((mn > 0.0f) && (s <= sp::FMAXV))) return 2;
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            dst[i * 8 + j] = t[i][8 + j];
        });
    });
    return 0;
}

// InverseMatrix of order 8 with the reference's pivot search, every array index static: the pivot row is picked with selects
// (first not-yet-used row, in row order, whose entry in column i exceeds eps), all 16 columns of every row take part as in the
// reference (no structural shortcuts), and the closing "row that holds an exact 1 in column r goes to position r" pass
// (matrix.h:262-288) is replayed with predicated swaps.  ~4x the work of inverse8_sparse, ~1/20 of the index-driven generic
// routine; used for the inversions inverse8_sparse hands back.  Returns 0 = inverted (dst written), 1 = no pivot (dst untouched).
// Deliberately NOT inlined: inlined into the draw loop its 128 + 16 live values push the common path into scratch (measured
// 5.7 against 4.4 us per pair); the caller passes copies so that its own arrays stay in registers.
__host__ __device__ __attribute__((noinline)) inline int inverse8_pivot(const float* src, float* dst, float eps) {
    float t[8][16];
    bool used[8];
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        used[i] = false;
        sp::sfor<0, 16>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < 8) t[i][j] = src[i * 8 + j]; else t[i][j] = (j - 8 == i) ? 1.0f : 0.0f;
        });
    });
    bool failed = false;
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        int rowI = -1;
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (rowI < 0 && !used[j] && fabsf(t[j][i]) > eps) rowI = j;
        });
        if (rowI < 0) failed = true;
        float pr[16];
        float ei = 1.0f;
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j == rowI) { used[j] = true; ei = t[j][i]; }
        });
        sp::sfor<0, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            float v = 0.0f;
            sp::sfor<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; v = (j == rowI) ? t[j][c] : v; });
            pr[c] = v / ei;
        });
        sp::sfor<0, 8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float e2 = t[j][i];
            const bool is_p = (j == rowI);
            const bool upd = !is_p && !(fabsf(e2) < eps);
            const float ne = -e2;
            sp::sfor<0, 16>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const float prod = ne * pr[c];
                const float sum = t[j][c] + prod;
                t[j][c] = is_p ? pr[c] : (upd ? sum : t[j][c]);
            });
        });
    });
    if (failed) return 1;
    sp::sfor<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        int target = -1;
        sp::sfor<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (target < 0 && t[i][r] == 1.0f) target = i;
        });
        sp::sfor<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i != r) {
                const bool sw = (target == i);
                sp::sfor<0, 16>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    const float a = t[r][c], b = t[i][c];
                    t[r][c] = sw ? b : a; t[i][c] = sw ? a : b;
                });
            }
        });
    });
    sp::sfor<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        sp::sfor<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; dst[i * 8 + j] = t[i][8 + j]; });
    });
    return 0;
}

// The same sparse elimination with true divisions, out of line (its own registers): where the shared-reciprocal form leaves its guarded
// range (state 3) the result is formed again this way -- it is the host / reference form of the very same plan -- and only a pivot below the
// diagonal (state 2) still goes to the generic routine.  A draw that leaves the guard does so in every one of its 15 Gauss-Newton steps:
// through inverse8_pivot (~2000 ticks per call) such a draw held its wave for 30 000 ticks, a quarter of all pairs have one.
__host__ __device__ __attribute__((noinline)) inline int inverse8_sparse_exact(const float* src, float* dst, float eps) {
    return inverse8_sparse<false>(src, dst, eps);
}
HD int exact_call(const float* M, float* inv, float eps) {
    float in[64], out[64];
#pragma unroll
    for (int i = 0; i < 64; i++) { in[i] = M[i]; out[i] = inv[i]; }
    const int st = inverse8_sparse_exact(in, out, eps);
#pragma unroll
    for (int i = 0; i < 64; i++) inv[i] = out[i];
    return st;
}

HD void pivot_call(const float* M, float* inv, float eps) {
    float in[64], out[64];
#pragma unroll
    for (int i = 0; i < 64; i++) { in[i] = M[i]; out[i] = inv[i]; }
    inverse8_pivot(in, out, eps);
#pragma unroll
    for (int i = 0; i < 64; i++) inv[i] = out[i];
}

#if defined(__HIP_DEVICE_COMPILE__)
// InverseMatrix of order 8 (matrix.h:147-296) of ONE lane's matrix by the lanes of its wave that are active with it: the register form above
// holds a lane for ~50 us per call (its 144 live values live in scratch at the draw loop's register budget), and a draw whose J^T J needs a
// pivot below the diagonal needs one in each of its 15 Gauss-Newton steps -- one such draw kept its wave, and with it its image pair, for
// 0.7 ms (a rank's 63 pairs then wait for that one).  Here lane `src` puts its matrix into the wave's LDS scratch (wl: 192 floats), the active
// lanes (at least 32: the caller checks) share the 8 x 16 tableau's 128 entries and perform on each entry exactly the operations the
// sequential routine performs on it -- the scheme of ransac.hip inverse8_team with the wave in place of the workgroup: pivot search (first
// unused row whose entry in column i exceeds eps) by every lane on the same values, pivot row divided by the pivot, every other row whose
// entry in the pivot column is not below eps takes row + (-entry) * pivot row, and the closing pass moves the row that holds an exact 1 in
// column r to position r.  No pivot: inv stays as it is (the caller keeps the stale inverse).  LDS operations of one wave execute in order,
// so the lanes only have to wait for their own reads before the writes of a step.
// (inlined: as a call it costs the draw loop a stack frame and measured 2.7 against 1.5 us per pair)
__device__ __forceinline__ void inverse8_wave(const float* M, float* inv, float eps, float* wl, int src, unsigned long long active) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rank = __builtin_popcountll(active & ((1ull << lane) - 1ull)), na = __builtin_popcountll(active);      // na >= 32: at most 4 entries per lane
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    float* t = wl + 64;
    wsync();                                                  // earlier users of the scratch are done
    if (lane == src) {
#pragma unroll
        for (int k = 0; k < 64; k++) wl[k] = M[k];
    }
    wsync();
    for (int e = rank; e < 128; e += na) { const int j = e >> 4, c = e & 15; t[e] = c < 8 ? wl[j * 8 + c] : ((c - 8 == j) ? 1.0f : 0.0f); }
    wsync();
    unsigned used = 0;
    bool failed = false;
    for (int i = 0; i < 8; i++) {
        int rowI = -1;
        for (int jj = 0; jj < 8; jj++) if (rowI < 0 && !((used >> jj) & 1u) && fabsf(t[jj * 16 + i]) > eps) rowI = jj;
        if (rowI < 0) { failed = true; break; }               // matrix.h:206-222: no pivot, the routine gives up (the same verdict in every lane)
        used |= 1u << rowI;
        const float ei = t[rowI * 16 + i];
        float nv[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int e = rank + m * na;
            nv[m] = 0.0f;
            if (e < 128) {
                const int j = e >> 4, c = e & 15;
                const float prc = t[rowI * 16 + c] / ei;
                const float e2 = t[j * 16 + i], old = t[e];
                float v = old;
                if (j == rowI) v = prc;
                else if (!(fabsf(e2) < eps)) { const float ne = -e2; const float prod = ne * prc; v = old + prod; }
                nv[m] = v;
            }
        }
        wsync();                                              // every read of this step precedes its writes
#pragma unroll
        for (int m = 0; m < 4; m++) { const int e = rank + m * na; if (e < 128) t[e] = nv[m]; }
        wsync();
    }
    if (!failed) {
        for (int r = 0; r < 8; r++) {
            int target = -1;
            for (int ii = 0; ii < 8; ii++) if (target < 0 && t[ii * 16 + r] == 1.0f) target = ii;
            const bool sw = target >= 0 && target != r && rank < 16;      // lanes of rank 0 .. 15 swap one column each
            float x = 0.0f, y = 0.0f;
            if (sw) { x = t[r * 16 + rank]; y = t[target * 16 + rank]; }
            wsync();
            if (sw) { t[r * 16 + rank] = y; t[target * 16 + rank] = x; }
            wsync();
        }
        if (lane == src) {
#pragma unroll
            for (int k = 0; k < 64; k++) inv[k] = t[(k >> 3) * 16 + 8 + (k & 7)];
        }
    }
    wsync();
}
// the inversions inverse8_sparse hands back for a pivot search below the diagonal: with the wave's LDS scratch at hand and at least half of
// the wave active, one lane's matrix after the other by the wave together; else (host, or few active lanes) the register routine per lane
#ifndef MI355_PIVOT_WAVE_MAX
#define MI355_PIVOT_WAVE_MAX 6
#endif
constexpr int PIVOT_WAVE_MAX = MI355_PIVOT_WAVE_MAX;
__device__ __forceinline__ void pivot_dispatch(bool need, const float* M, float* inv, float eps, float* wl) {
    if (wl) {
        const unsigned long long active = __ballot(1);
        unsigned long long nm = __ballot(need);
        if (nm == 0) return;
        // one matrix by the wave takes a fraction of what the register routine takes a lane -- but that routine runs for all the lanes that need it
        // at once: a few matrices go to the wave (a pair of a strip survey: one draw in a thousand needs a pivot, and one lane held its pair for
        // 15 x 50 us), many stay with their lanes (pairs of unrelated images, most of a window survey, draw degenerate configurations by the dozen)
        if (__builtin_popcountll(active) >= 32 && __builtin_popcountll(nm) <= PIVOT_WAVE_MAX) { for (; nm; nm &= nm - 1ull) inverse8_wave(M, inv, eps, wl, __builtin_ctzll(nm), active); return; }
    }
    if (need) pivot_call(M, inv, eps);
}
#endif

// (J^T J)^-1 J^T with J's pattern (MulMatrix order: ascending k over the entries that are not structural zeros)
HD void invjt_sparse(const float* inv, const float* A, float* M) {
    sp::sfor<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        sp::sfor<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            float acc = 0.0f;
            sp::sfor<0, 8>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (sp::anz(c, k)) { const float pr = inv[r * 8 + k] * A[c * 8 + k]; acc = acc + pr; }
            });
            M[r * 8 + c] = acc;
        });
    });
}

// 4-point SolveHomographyMatrix + (when 0.01 < H[8] < 5) NonlinearLeastSquareProjection2, everything in registers.
// Returns false when an inversion needs the generic routine (caller falls back to solve_h4 / nlls4).
// POLISH = false stops after the 4-point solve: H[8] is its residual and *polished tells whether the polish WOULD run (the draw
// classification pass of csrc/ransac.hip; the arithmetic up to that point is the same instruction for instruction).
// wave_lds (device, optional): 192 floats of LDS private to the calling wave, for the inversions that need a pivot search (pivot_dispatch)
template <bool POLISH = true>
HD bool hypothesis4_fast(const float* p, float* H, int* polished = nullptr, float* wave_lds = nullptr) {
    (void)wave_lds;
    float A[64], M[64], inv[64], B[8];
#pragma unroll
    for (int i = 0; i < 64; i++) A[i] = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float x1 = p[4 * r], y1 = p[4 * r + 1], x2 = p[4 * r + 2], y2 = p[4 * r + 3];
        A[2 * r * 8 + 0] = x2; A[2 * r * 8 + 1] = y2; A[2 * r * 8 + 2] = 1.0f; A[2 * r * 8 + 6] = (-x1) * x2; A[2 * r * 8 + 7] = (-x1) * y2;
        A[(2 * r + 1) * 8 + 3] = x2; A[(2 * r + 1) * 8 + 4] = y2; A[(2 * r + 1) * 8 + 5] = 1.0f; A[(2 * r + 1) * 8 + 6] = (-y1) * x2; A[(2 * r + 1) * 8 + 7] = (-y1) * y2;
        B[2 * r] = x1; B[2 * r + 1] = y1;
    }
    // matrix.h:377: on failure the reference multiplies with an all-zero inverse.  A missing pivot (|.| <= 1e-20) is the
    // generic routine's business.
    if (!jtj_sparse(A, M)) return false;
#pragma unroll
    for (int i = 0; i < 64; i++) inv[i] = 0.0f;                   // a failed inversion leaves the zeros (matrix.h:377)
#if defined(__HIP_DEVICE_COMPILE__)
    {
        int st = inverse8_sparse<true>(M, inv, 1e-20f);
        if (st == 3) st = exact_call(M, inv, 1e-20f);            // division guard: the same plan with true divisions
        pivot_dispatch(st == 2, M, inv, 1e-20f, wave_lds);       // pivot search below the diagonal
    }
#else
    if (inverse8_sparse(M, inv, 1e-20f) == 2) pivot_call(M, inv, 1e-20f);
#endif
    invjt_sparse(inv, A, M);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) { const float pr = M[r * 8 + k] * B[k]; acc = acc + pr; }
        H[r] = acc;
    }
    double emax = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float fx, fy;
        apply_recip1(H, p[4 * i + 2], p[4 * i + 3], fx, fy);
        const double dx = (double)p[4 * i] - (double)fx, dy = (double)p[4 * i + 1] - (double)fy;
        const double d = sqrt(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    H[8] = (float)emax;
    if (!(H[8] < 5.0f && H[8] > 0.01f)) return true;             // no polish (mosaicimage.h:1864-1876)
    if (polished) *polished = 1;
    if constexpr (!POLISH) return true;
    // ---- Gauss-Newton polish, LeastSquare.h:353-531 (J is A's storage) ----
    float w[8], C[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = H[i];
    bool finished = false;
#pragma unroll
    for (int i = 0; i < 64; i++) inv[i] = 0.0f;
    for (int it = 0; it < 15 && !finished; it++) {
        auto jacobian = [&](auto fast) {
            constexpr bool FAST = decltype(fast)::value;
            DivGuard g; guard_init(g);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float x2 = p[4 * i], y2 = p[4 * i + 1], x1 = p[4 * i + 2], y1 = p[4 * i + 3];
                const float d = w[6] * x1 + w[7] * y1 + 1.0f;
                const float nx = w[0] * x1 + w[1] * y1 + w[2];
                const float ny = w[3] * x1 + w[4] * y1 + w[5];
                float* j = A + i * 16;
                j[3] = 0.0f; j[4] = 0.0f; j[5] = 0.0f; j[8] = 0.0f; j[9] = 0.0f; j[10] = 0.0f;
                if constexpr (FAST) {
                    const float dd = d * d, rd = rcp_nr(d, g), rdd = rcp_nr(dd, g);
                    j[0] = div_nr(x1, d, rd, g); j[1] = div_nr(y1, d, rd, g); j[2] = div_nr(1.0f, d, rd, g);
                    j[6] = div_nr((-x1) * nx, dd, rdd, g); j[7] = div_nr((-y1) * nx, dd, rdd, g);
                    j[11] = j[0]; j[12] = j[1]; j[13] = j[2];
                    j[14] = div_nr((-x1) * ny, dd, rdd, g); j[15] = div_nr((-y1) * ny, dd, rdd, g);
                    C[2 * i] = x2 - div_nr(nx, d, rd, g); C[2 * i + 1] = y2 - div_nr(ny, d, rd, g);
                } else {
                    j[0] = x1 / d; j[1] = y1 / d; j[2] = 1.0f / d;
                    j[6] = ((-x1) * nx) / (d * d); j[7] = ((-y1) * nx) / (d * d);
                    j[11] = x1 / d; j[12] = y1 / d; j[13] = 1.0f / d;
                    j[14] = ((-x1) * ny) / (d * d); j[15] = ((-y1) * ny) / (d * d);
                    C[2 * i] = x2 - nx / d; C[2 * i + 1] = y2 - ny / d;
                }
            }
            return guard_ok(g);
        };
#if defined(__HIP_DEVICE_COMPILE__)
        if (!jacobian(std::true_type{})) { jacobian(std::false_type{}); if (polished) *polished |= 2; }   // (rare) a division outside the guarded range: true divisions
        if (!jtj_sparse(A, M)) return false;
        int inv_state = inverse8_sparse<true>(M, inv, 1e-6f);    // 1: no pivot, the previous iteration's inverse stays (zeros before the first)
        if (inv_state == 3) { if (polished) *polished |= 4; inv_state = exact_call(M, inv, 1e-6f); }      // 3: division guard -> the same plan with true divisions
        pivot_dispatch(inv_state == 2, M, inv, 1e-6f, wave_lds);                                          // 2: pivot search below the diagonal
#else
        jacobian(std::false_type{});
        if (!jtj_sparse(A, M)) return false;
        if (inverse8_sparse(M, inv, 1e-6f) == 2) pivot_call(M, inv, 1e-6f);    // failure: the previous iteration's inverse stays (zeros before the first)
#endif
        invjt_sparse(inv, A, M);
        bool done = true;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) { const float pr = M[r * 8 + k] * C[k]; acc = acc + pr; }
            w[r] = w[r] + acc;
            if (!(fabsf(acc) < 1e-10f)) done = false;
        }
        finished = done;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) H[i] = w[i];
    float em = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float fx, fy;
        apply_recip1(H, p[4 * i + 2], p[4 * i + 3], fx, fy);
        const float dx = p[4 * i] - fx, dy = p[4 * i + 1] - fy;
        const float d = sqrtf(dx * dx + dy * dy);
        if (d > em) em = d;
    }
    H[8] = em;
    return true;
}

// the pixel expression of every warp (MosaicWithoutPos.cpp:2331-2334, MosaicImage.cpp:1715-1719):
// (uchar)( s00*(1-p)*(1-q) + s01*(1-p)*q + s10*p*(1-q) + s11*p*q ), terms ((s*a)*b), summed left to right
HD unsigned char bilin(float s00, float s01, float s10, float s11, float p, float q) {
    float omp = 1.0f - p, omq = 1.0f - q;
    float t0 = (s00 * omp) * omq;
    float t1 = (s01 * omp) * q;
    float t2 = (s10 * p) * omq;
    float t3 = (s11 * p) * q;
    float v = ((t0 + t1) + t2) + t3;
    return (unsigned char)(int)v;
}

}  // namespace hm
