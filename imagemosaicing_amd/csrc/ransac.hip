// csrc/ransac.hip -- K8: batched Ransac2D (mosaicimage.h:1729-2035) on gfx950, one workgroup per image pair.
//
// What is reproduced exactly:
//  * the glibc rand() stream after srand(seed) and the "redraw all four until pairwise distinct" rule
//    (mosaicimage.h:1777-1813): the stream is consumed in aligned groups of four, so the k-th draw is the
//    k-th group whose four values are distinct mod n -- the host builds that table once per (seed, n);
//  * every draw is an independent 4-point SolveHomographyMatrix (+ NonlinearLeastSquareProjection2 when
//    0.01 < H[8] < 5) in float32 with the reference's operation order (csrc/hmath.h): one lane per draw;
//  * the sequential bookkeeping of the loop -- a draw with H[8] > 5 is skipped without consuming a
//    hypothesis slot, at most sample_times accepted hypotheses, at most 4999 draws, first strict maximum of
//    the support wins, stop as soon as support/n > 0.99 -- is replayed in draw order over each chunk of 256
//    evaluated draws, so the winner is the one the sequential reference finds;
//  * support uses the reciprocal form (ApplyProjectMat2), the final inlier split the true-division form
//    (ApplyProjectMat3); inliers keep input order; the result is NLLS from the WINNING hypothesis over all
//    inliers (the inlier least-squares refit at :1958 only decides success and is not computed).
//
// CDNA4 mapping: points of the pair live in LDS (<= 16 KB), support counting is a per-lane loop over LDS
// broadcast reads (all lanes read the same point => conflict-free broadcast), inlier compaction uses wave
// ballots + popcounts, the final Gauss-Newton keeps J / J* in LDS and gives each of the 64 entries of
// J^T J to one lane of a wave so that the k-ordered accumulation of the reference is preserved per entry.
#include "common.h"
#include "hmath.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int RB = 256;            // threads per workgroup = draws per chunk
constexpr int RANSAC_LDS_POINTS = 4096;   // largest n whose points stay in LDS (64 KB)
constexpr int MAX_DRAWS = 4999;    // realSamTimes >= 5000 breaks before drawing (mosaicimage.h:1787-1792)

struct RansacArgs {
    const mi355_sfpoint* p1;       // [pair][stride]
    const mi355_sfpoint* p2;
    const int* n;                  // [pair]
    long long* dbg;                // optional: per pair 8 cycle stamps (MI355_RANSAC_DBG)
    const uint16_t* tables;        // concatenated draw tables (MAX_DRAWS x 4 each)
    int single_table;              // 1: `tables` IS the one table of this call's n (mi_ransac_big), whatever n is
    const int* table_of;           // [pair] table index, -1 = none (n < 4)
    int stride;
    float dist;
    int sample_times;
    int list_floats;               // floats of dynamic LDS in front of the point arrays: the list of accepted draws and their supports (2 x uint16 x min(sample_times, 4999)); also the uint16 offset of the supports
    int hb_off;                    // float offset of the lanes' best hypotheses (9 x RB floats) inside the dynamic LDS: behind the points and their float4 copy, where J* is kept later
    int min_keep;                  // pairs with at most this many inliers skip the closing refinement (-1: never): the caller rejects them anyway
    mi355_pair_result* out;        // [pair]
    // BIG variants (one pair with n > 400, mi355_ransac2d only): work arrays of the closing Gauss-Newton and the inlier lists in HBM
    float* big_ws;                 // 38 n floats: J (16 n), J* (16 n), C (2 n), compaction scratch (4 n)
    mi355_sfpoint* big_a;          // [n] inliers of image i
    mi355_sfpoint* big_b;          // [n] inliers of image j
    float* big_pts;                // MODE 2: x1, y1, x2, y2 (4 n floats) in HBM
    uint16_t* list_hbm;            // LISTG kernels only: [pair][2 x list_floats] uint16 -- the list of accepted draws and their supports in HBM (sample_times near 5000)
};

// the index-driven generic routines (hmath.h solve_h4 / nlls4) for the draws the register path hands back (a non-finite entry
// in the design matrix): a call, not inlined, so that the draw loop's registers are not shared with it.  Returns the skip flag.
__device__ __attribute__((noinline)) bool generic_hypothesis(const float* p, float* h, float* scr) {
    hm::solve_h4(p, h, scr);                               // mosaicimage.h:1863
    const bool skip = h[8] > 5.0f;                         // :1864-1867
    if (!skip && h[8] < 5.0f && h[8] > 0.01f) {            // :1868-1876
        float fine[9];
        hm::nlls4(p, h, fine, scr);
        for (int i = 0; i < 9; i++) h[i] = fine[i];
    }
    return skip;
}

// InverseMatrix of order 8 (matrix.h:147-296, hm::inverse_matrix) by the whole workgroup: thread (j, c) of the first 128 owns entry (j, c) of
// the 8 x 16 tableau in LDS and performs on it exactly the operations the sequential routine performs on that entry -- the pivot search
// (first unused row whose entry in column i exceeds eps) is evaluated by every thread on the same LDS values, the pivot row is divided by
// the pivot, every other row whose entry in the pivot column is not below eps takes row + (-entry) * pivot row, and the closing pass swaps
// the row holding an exact 1 in column r into position r.  A missing pivot leaves dst untouched (the caller keeps the stale inverse).
// One lane running the index-driven routine took ~55 us per call, 15 calls per accepted pair: as long as the whole hypothesis loop.
// All RB threads must call; uniform control flow.
__device__ __forceinline__ void inverse8_team(const float* src, float* dst, float eps, float* t, int tid) {
    const int j = (tid >> 4) & 7, c = tid & 15;
    if (tid < 128) t[tid] = c < 8 ? src[j * 8 + c] : ((c - 8 == j) ? 1.0f : 0.0f);
    __syncthreads();
    unsigned used = 0;
    for (int i = 0; i < 8; i++) {
        int rowI = -1;
        for (int jj = 0; jj < 8; jj++) if (rowI < 0 && !((used >> jj) & 1u) && fabsf(t[jj * 16 + i]) > eps) rowI = jj;
        if (rowI < 0) return;                                // matrix.h:206-222: no pivot, the routine gives up
        used |= 1u << rowI;
        const float ei = t[rowI * 16 + i];
        const float prc = t[rowI * 16 + c] / ei;
        const float e2 = t[j * 16 + i], old = t[j * 16 + c];
        __syncthreads();                                      // every read of this step precedes its writes
        if (tid < 128) {
            float nv = old;
            if (j == rowI) nv = prc;
            else if (!(fabsf(e2) < eps)) { const float ne = -e2; const float prod = ne * prc; nv = old + prod; }
            t[j * 16 + c] = nv;
        }
        __syncthreads();
    }
    for (int r = 0; r < 8; r++) {
        int target = -1;
        for (int ii = 0; ii < 8; ii++) if (target < 0 && t[ii * 16 + r] == 1.0f) target = ii;
        const bool sw = target >= 0 && target != r;
        float a = 0.0f, b = 0.0f;
        if (sw && tid < 16) { a = t[r * 16 + tid]; b = t[target * 16 + tid]; }
        __syncthreads();
        if (sw && tid < 16) { t[r * 16 + tid] = b; t[target * 16 + tid] = a; }
        __syncthreads();
    }
    if (tid < 64) dst[tid] = t[(tid >> 3) * 16 + 8 + (tid & 7)];
}

__device__ __forceinline__ int block_exclusive_scan_flags(bool flag, int tid, int* wave_tot /*LDS[RB/64+1]*/, int& total) {
    const unsigned long long m = __ballot(flag);
    const int lane = tid & 63, wv = tid >> 6;
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < RB / 64; i++) { int c = wave_tot[i]; if (i < wv) off += c; tot += c; }
    __syncthreads();
    total = tot;
    return off + before;
}

// the small shared arrays of a pair's workgroup (one object, so that the stages below can be functions)
struct alignas(16) RShared {
    unsigned long long mask[5][RB / 64];
    unsigned wkey[RB / 64];
    float bestH[9], firstH[9], w[8], dX[8], T1[64], T2[64], t[128];
    int   state[8];      // 0 t_acc, 1 maxSupport, 2 bestDraw, 3 firstAcc, 4 finished, 5 newBest, 6 newFirst, 7 done
    int   wtot[RB / 64 + 1];
    int   npol;
    int   next;          // next group of 64 list entries to hand to a wave
    int   fb;            // draws that needed the generic (private-memory) solve: diagnostic, reported in _pad
    float fbk[RB / 64][320];   // work arrays of the generic solve, one slot per wave
};

// ---- stage 1: does draw r hold a hypothesis slot? --------------------------------------------------------------------------------------
// A draw whose 4-point solve leaves a residual above 5 px is skipped without consuming a slot (mosaicimage.h:1864-1867): on unrelated image
// pairs that is 42 % of the draws, and 6.8 chunks of 256 draws were walked for the 1000 slots with those lanes idle through the 15
// Gauss-Newton iterations of their neighbours' polish (80 % of the kernel's time).  The solve alone is 1 / 60 of a polished draw: it runs
// for every draw first, the accepted ones are kept in draw order and then evaluated densely packed.
__device__ __forceinline__ bool classify_draw(const uint16_t* table, int r, const float* x1, const float* y1, const float* x2, const float* y2, RShared& sh, int tid) {
    float p[16], h[9];
    const uint16_t* s = table + 4 * r;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int k = s[i]; p[4 * i] = x1[k]; p[4 * i + 1] = y1[k]; p[4 * i + 2] = x2[k]; p[4 * i + 3] = y2[k]; }
    int pol = 0;
    const bool fast_ok = hm::hypothesis4_fast<false>(p, h, &pol, sh.fbk[tid >> 6]);
    bool skip = !pol && h[8] > 5.0f;
    for (unsigned long long need = __ballot(!fast_ok); need; need &= need - 1ull) {
        if ((tid & 63) == __builtin_ctzll(need)) {
            float pin[16], hout[9];
#pragma unroll
            for (int i = 0; i < 16; i++) pin[i] = p[i];
            skip = generic_hypothesis(pin, hout, sh.fbk[tid >> 6]);
        }
    }
    return !skip;
}

// ---- stage 2: hypothesis + support of draw r (mosaicimage.h:1863-1904) -------------------------------------------------------------------
template <bool BIG>
__device__ __forceinline__ int eval_draw(const RansacArgs& a, const uint16_t* table, int r, int n, float d2, const float* x1, const float* y1, const float* x2, const float* y2,
                                         const float4* pts, float* h, RShared& sh, int tid, long long* t_solve) {
    const long long c0 = wall_clock64();
    float p[16];
    const uint16_t* s = table + 4 * r;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int k = s[i]; p[4 * i] = x1[k]; p[4 * i + 1] = y1[k]; p[4 * i + 2] = x2[k]; p[4 * i + 3] = y2[k]; }
    // register-resident solve + polish (structural zeros skipped, failed inversions reproduced: hmath.h); the rare draws
    // whose inversion needs the reference's pivot search below the diagonal re-run the generic private-memory routines
    int pol = 0;
    const bool fast_ok = hm::hypothesis4_fast(p, h, &pol, sh.fbk[tid >> 6]);
    if (a.dbg && pol) atomicAdd(&sh.npol, 1 + ((pol & 2) ? (1 << 12) : 0) + ((pol & 4) ? (1 << 22) : 0));      // polished draws | << 12: a Jacobian redone with true divisions | << 22: an inversion
    // one lane of the wave at a time, its work arrays in the wave's LDS slot: a private array for these index-driven
    // routines costs the whole kernel registers and scratch set-up (measured 5.3 us per pair against 4.9 this way)
    for (unsigned long long need = __ballot(!fast_ok); need; need &= need - 1ull) {
        if ((tid & 63) == __builtin_ctzll(need)) {
            atomicAdd(&sh.fb, 1);                    // statistics only (reported in _pad)
            float pin[16], hout[9];
#pragma unroll
            for (int i = 0; i < 16; i++) pin[i] = p[i];
            (void)generic_hypothesis(pin, hout, sh.fbk[tid >> 6]);
#pragma unroll
            for (int i = 0; i < 9; i++) h[i] = hout[i];
        }
    }
    *t_solve += wall_clock64() - c0;
    // (a polished hypothesis is never skipped, whatever its residual after the polish: :1868-1876)
    int support = 0;
    if constexpr (BIG) {
        for (int i = 0; i < n; i++) {              // :1890-1904
            float bx, by;
            hm::apply_recip1(h, x2[i], y2[i], bx, by);
            const float dx = bx - x1[i], dy = by - y1[i];
            const float dd = dx * dx + dy * dy;
            if (dd < d2) support++;
        }
    } else {
        // the same expressions (ApplyProjectMat2, :1890-1904) with X and Y side by side in explicit pairs: every product and sum is
        // rounded separately as before (-ffp-contract=off), the pairs are packed instructions whatever the vectoriser's settings
        const f2 m03 = {h[0], h[3]}, m14 = {h[1], h[4]}, m25 = {h[2], h[5]};
        const float m6 = h[6], m7 = h[7];
#pragma unroll 4                                             // four independent points in flight: the chain of one (LDS read, 11-instruction division) is all latency
        for (int i = 0; i < n; i++) {
            const float4 q = pts[i];
            const float inv = 1.0f / (m6 * q.z + m7 * q.w + 1.0f);
            const f2 num = (m03 * q.z + m14 * q.w) + m25;
            const f2 b = num * inv;
            const f2 t1 = {q.x, q.y};
            const f2 d = b - t1;
            const f2 sq = d * d;
            const float dd = sq.x + sq.y;
            if (dd < d2) support++;
        }
    }
    return support;
}

// ---- stage 3: the loop's bookkeeping over the stored supports, in list order (mosaicimage.h:1864-1918) -------------------------------------
// A draw replaces the best one when its support is strictly larger (and ends the loop at once when that support exceeds 0.99 n); the slot
// limit is the list's length (list_cap <= sample_times).  E = the last draw the loop looks at; the winner is the first draw <= E holding the
// maximum, if that is positive.  All RB threads call; M = the winner's support (0: none), win = its list index.
__device__ __forceinline__ void replay_supports(const uint16_t* sup, int nlist, float invn, RShared& sh, int tid, int& M, int& win) {
    const int lane = tid & 63, wv = tid >> 6;
    int run = 0;                                          // maximum over the blocks before this one
    unsigned bestkey = 0;
    bool stopped = false;
    for (int b0 = 0; b0 < nlist && !stopped; b0 += RB) {
        const int i = b0 + tid;
        const bool valid = i < nlist;
        const int sv = valid ? (int)sup[i] : 0;
        int v = sv;                                        // inclusive maximum scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v = t > v ? t : v; }
        int ex = __shfl_up(v, 1, 64); if (lane == 0) ex = 0;
        if (lane == 63) sh.wtot[wv] = v;
        __syncthreads();
        int prev = run, blk = run;
        for (int w = 0; w < RB / 64; w++) { const int t = sh.wtot[w]; if (w < wv) prev = t > prev ? t : prev; blk = t > blk ? t : blk; }
        ex = ex > prev ? ex : prev;                        // maximum of every earlier draw (0 before the first: :1783)
        const unsigned long long m_r = __ballot(valid && sv > ex && (float)sv * invn > 0.99f);
        if (lane == 0) sh.mask[0][wv] = m_r;
        __syncthreads();
        int kr = RB;
        for (int w = 0; w < RB / 64; w++) { const unsigned long long m = sh.mask[0][w]; if (m && kr == RB) kr = 64 * w + (int)__builtin_ctzll(m); }
        int E = b0 + RB - 1;
        if (kr < RB) { E = b0 + kr; stopped = true; }
        unsigned key = (valid && i <= E) ? (((unsigned)sv << 13) | (unsigned)(8191 - i)) : 0u;      // i <= 4998 < 2^13, support < 2^16
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const unsigned t = __shfl_xor(key, o, 64); key = t > key ? t : key; }
        if (lane == 0) sh.wkey[wv] = key;
        __syncthreads();
        for (int w = 0; w < RB / 64; w++) bestkey = sh.wkey[w] > bestkey ? sh.wkey[w] : bestkey;
        run = blk;
        __syncthreads();                                   // wtot / mask / wkey are rewritten by the next block
    }
    M = (int)(bestkey >> 13); win = 8191 - (int)(bestkey & 8191u);
}

// the winner's hypothesis formed again by one lane (same operations, same bits): the lane that evaluated it went on to a larger support
// behind a stop, or (split form) its record is not at hand
__device__ __forceinline__ void recompute_hypothesis(const uint16_t* table, int r, const float* x1, const float* y1, const float* x2, const float* y2, RShared& sh) {
    float p[16];
    const uint16_t* sx = table + 4 * r;
    for (int i = 0; i < 4; i++) { const int k = sx[i]; p[4 * i] = x1[k]; p[4 * i + 1] = y1[k]; p[4 * i + 2] = x2[k]; p[4 * i + 3] = y2[k]; }
    int pol = 0;
    float hh[9];
    if (!hm::hypothesis4_fast(p, hh, &pol)) (void)generic_hypothesis(p, hh, sh.fbk[0]);
    for (int i = 0; i < 9; i++) sh.bestH[i] = hh[i];
}

// ---- stage 4: inlier split, true-division form (:1922-1944), order preserving compaction; returns the inlier count --------------------
template <bool BIG>
__device__ __forceinline__ int split_inliers(const RansacArgs& a, mi355_pair_result* out, const mi355_sfpoint* P1, const mi355_sfpoint* P2, int n, float d2, const float* W,
                                             const float* x1, const float* y1, const float* x2, const float* y2, float* C, RShared& sh, int tid) {
    int cnt = 0;
    for (int base = 0; base < n; base += RB) {
        const int i = base + tid;
        bool in = false;
        if (i < n) {
            float bx, by;
            hm::apply_div1(W, x2[i], y2[i], bx, by);
            const float dx = bx - x1[i], dy = by - y1[i];
            const float dd = dx * dx + dy * dy;
            in = dd < d2;
        }
        int tot;
        const int pos = cnt + block_exclusive_scan_flags(in, tid, sh.wtot, tot);
        if constexpr (BIG) { if (in) { a.big_a[pos] = P1[i]; a.big_b[pos] = P2[i]; } }
        else if (in && pos < MI355_MAX_SELECTED) { out->a[pos] = P1[i]; out->b[pos] = P2[i]; }
        if (in) { C[pos] = (float)i; }                      // remember source index (C reused later)
        cnt += tot;
    }
    __syncthreads();
    return cnt;
}

// inlier coordinates, compacted in place into the head of the point arrays (C holds the source indices)
template <bool BIG>
__device__ __forceinline__ void compact_inliers(int cnt, float* x1, float* y1, float* x2, float* y2, const float* C, float* CS, int tid) {
    if constexpr (BIG) {                                    // more than 2 per lane: through the HBM scratch
        for (int i = tid; i < cnt; i += RB) { const int s = (int)C[i]; CS[4 * i] = x1[s]; CS[4 * i + 1] = y1[s]; CS[4 * i + 2] = x2[s]; CS[4 * i + 3] = y2[s]; }
        __syncthreads();
        for (int i = tid; i < cnt; i += RB) { x1[i] = CS[4 * i]; y1[i] = CS[4 * i + 1]; x2[i] = CS[4 * i + 2]; y2[i] = CS[4 * i + 3]; }
    } else {                                                // read (<= 2 per lane since cnt <= 400), barrier, write
        float ix1[2], iy1[2], ix2[2], iy2[2];
        {
            int m = 0;
            for (int i = tid; i < cnt; i += RB) { const int s = (int)C[i]; ix1[m] = x1[s]; iy1[m] = y1[s]; ix2[m] = x2[s]; iy2[m] = y2[s]; m++; }
        }
        __syncthreads();
        {
            int m = 0;
            for (int i = tid; i < cnt; i += RB) { x1[i] = ix1[m]; y1[i] = iy1[m]; x2[i] = ix2[m]; y2[i] = iy2[m]; m++; }
        }
    }
}

// the Jacobian rows and residuals of the closing refinement's step (LeastSquare.h:404-432; naming there: 1 = source, 2 = target)
__device__ __forceinline__ void nlls_jacobian(int cnt, const float* x1, const float* y1, const float* x2, const float* y2, const float* w, float* J, float* C, int tid) {
    for (int i = tid; i < cnt; i += RB) {
        const float X2 = x1[i], Y2 = y1[i], X1 = x2[i], Y1 = y2[i];
        const float d = w[6] * X1 + w[7] * Y1 + 1.0f;
        const float nx = w[0] * X1 + w[1] * Y1 + w[2];
        const float ny = w[3] * X1 + w[4] * Y1 + w[5];
        float* j = J + i * 16;
        j[0] = X1 / d; j[1] = Y1 / d; j[2] = 1.0f / d; j[3] = 0.0f; j[4] = 0.0f; j[5] = 0.0f;
        j[6] = ((-X1) * nx) / (d * d); j[7] = ((-Y1) * nx) / (d * d);
        j[8] = 0.0f; j[9] = 0.0f; j[10] = 0.0f; j[11] = X1 / d; j[12] = Y1 / d; j[13] = 1.0f / d;
        j[14] = ((-X1) * ny) / (d * d); j[15] = ((-Y1) * ny) / (d * d);
        C[2 * i] = X2 - nx / d; C[2 * i + 1] = Y2 - ny / d;
    }
}

// ---- stage 5: NonlinearLeastSquareProjection2 over the inliers from the winning hypothesis (:1977-1986); w in sh.w ------------------------
__device__ __forceinline__ void nlls_closing(int cnt, const float* x1, const float* y1, const float* x2, const float* y2, float* J, float* JL, float* C, RShared& sh, int tid) {
    const int rows = 2 * cnt;
    for (int it = 0; it < 15; it++) {
        nlls_jacobian(cnt, x1, y1, x2, y2, sh.w, J, C, tid);
        __syncthreads();
        if (tid < 64) {                                     // J^T J, entry (r,c): k-ordered accumulation
            const int r = tid >> 3, c = tid & 7;
            float acc = 0.0f;
#pragma unroll 8                                             // the loads of eight steps in flight; the sum stays in k order
            for (int k = 0; k < rows; k++) { const float pr = J[k * 8 + r] * J[k * 8 + c]; acc = acc + pr; }
            sh.T1[tid] = acc;
        }
        __syncthreads();
        inverse8_team(sh.T1, sh.T2, 1e-6f, sh.t, tid);                          // LeastSquare.h:451 (stale T2 on failure)
        __syncthreads();
        for (int e = tid; e < 8 * rows; e += RB) {         // J* = (J^T J)^-1 J^T
            const int r = e / rows, k = e - r * rows;
            float acc = 0.0f;
#pragma unroll
            for (int m = 0; m < 8; m++) { const float pr = sh.T2[r * 8 + m] * J[k * 8 + m]; acc = acc + pr; }
            JL[e] = acc;
        }
        __syncthreads();
        if (tid < 8) {
            float acc = 0.0f;
#pragma unroll 8
            for (int k = 0; k < rows; k++) { const float pr = JL[tid * rows + k] * C[k]; acc = acc + pr; }
            sh.dX[tid] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            int done = 1;
            for (int i = 0; i < 8; i++) { sh.w[i] = sh.w[i] + sh.dX[i]; if (!(fabsf(sh.dX[i]) < 1e-10f)) done = 0; }
            sh.state[7] = done;
        }
        __syncthreads();
        if (sh.state[7]) break;
    }
}

// The same refinement for the split form (a pair's workgroup alone on its CU, LDS to spare): the two k-ordered sums of a step -- J^T J's 64
// entries over the 2 cnt Jacobian rows, the 8 entries of the update -- are chains of 2 cnt dependent additions whatever is done, but
// the chain's wave above also forms each product and reads its two factors (4 instructions per term, one wave issuing alone).  Here ALL
// threads form the products first (a thread per Jacobian row: 36 distinct products of its 8 entries; J^T J is symmetric and a product does
// not care about the order of its factors), into PQ[entry][k] with k contiguous, and the chain's wave only adds: one 16-byte LDS read per
// four terms.  Every product and every sum is the one of nlls_closing, in the same order.  PQ: 36 x pq_stride floats (>= 2 cnt, a multiple
// of 4 whose quarter is odd: the entries' rows then start in different banks); it also takes J* and the update's products later.
__device__ __forceinline__ void nlls_closing_wide(int cnt, const float* x1, const float* y1, const float* x2, const float* y2, float* J, float* PQ, int pq_stride, float* C, RShared& sh, int tid) {
    const int rows = 2 * cnt;
    float* JL = PQ;                                         // [8][pq_stride], after the J^T J chain is done with PQ
    float* P2 = PQ + 8 * (size_t)pq_stride;                 // [8][pq_stride]: J*[r][k] * C[k]
    for (int it = 0; it < 15; it++) {
        nlls_jacobian(cnt, x1, y1, x2, y2, sh.w, J, C, tid);
        __syncthreads();
        for (int k = tid; k < rows; k += RB) {              // products of row k: entry (r, c), r <= c, at index r * 8 - r (r - 1) / 2 + (c - r)
            float jr[8];
#pragma unroll
            for (int m = 0; m < 8; m++) jr[m] = J[k * 8 + m];
            int e = 0;
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = r; c < 8; c++) { PQ[(size_t)e * pq_stride + k] = jr[r] * jr[c]; e++; }
        }
        __syncthreads();
        if (tid < 64) {                                     // J^T J, entry (r,c): k-ordered accumulation of the products
            const int r = tid >> 3, c = tid & 7, lo = r < c ? r : c, hi = r < c ? c : r;
            const float* pq = PQ + (size_t)(lo * 8 - lo * (lo - 1) / 2 + (hi - lo)) * pq_stride;
            float acc = 0.0f;
            int k = 0;
#pragma unroll 4
            for (; k + 4 <= rows; k += 4) { const float4 q = *reinterpret_cast<const float4*>(pq + k); acc = acc + q.x; acc = acc + q.y; acc = acc + q.z; acc = acc + q.w; }
            for (; k < rows; k++) acc = acc + pq[k];
            sh.T1[tid] = acc;
        }
        __syncthreads();
        inverse8_team(sh.T1, sh.T2, 1e-6f, sh.t, tid);                          // LeastSquare.h:451 (stale T2 on failure)
        __syncthreads();
        for (int k = tid; k < rows; k += RB) {              // J*[r][k] = sum_m inv[r][m] J[k][m] in m order; and its product with C[k]
            float jr[8];
#pragma unroll
            for (int m = 0; m < 8; m++) jr[m] = J[k * 8 + m];
            const float ck = C[k];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float acc = 0.0f;
#pragma unroll
                for (int m = 0; m < 8; m++) { const float pr = sh.T2[r * 8 + m] * jr[m]; acc = acc + pr; }
                JL[(size_t)r * pq_stride + k] = acc;
                P2[(size_t)r * pq_stride + k] = acc * ck;
            }
        }
        __syncthreads();
        if (tid < 8) {
            const float* pq = P2 + (size_t)tid * pq_stride;
            float acc = 0.0f;
            int k = 0;
#pragma unroll 4
            for (; k + 4 <= rows; k += 4) { const float4 q = *reinterpret_cast<const float4*>(pq + k); acc = acc + q.x; acc = acc + q.y; acc = acc + q.z; acc = acc + q.w; }
            for (; k < rows; k++) acc = acc + pq[k];
            sh.dX[tid] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            int done = 1;
            for (int i = 0; i < 8; i++) { sh.w[i] = sh.w[i] + sh.dX[i]; if (!(fabsf(sh.dX[i]) < 1e-10f)) done = 0; }
            sh.state[7] = done;
        }
        __syncthreads();
        if (sh.state[7]) break;
    }
}

// motion[8] = max residual in float (LeastSquare.h:503-519; max is order independent), then the record's head
__device__ __forceinline__ void write_result(mi355_pair_result* out, int cnt, const float* x1, const float* y1, const float* x2, const float* y2, RShared& sh, int tid) {
    float emax = 0.0f;
    for (int i = tid; i < cnt; i += RB) {
        float fx, fy;
        hm::apply_recip1(sh.w, x2[i], y2[i], fx, fy);
        const float dx = x1[i] - fx, dy = y1[i] - fy;
        const float d = sqrtf(dx * dx + dy * dy);
        if (d > emax) emax = d;
    }
    unsigned bits = __float_as_uint(emax);                  // non-negative floats order like their bit patterns
    if (emax != emax) bits = 0;                             // NaN never wins `d > max` in the reference
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(bits, off); bits = o > bits ? o : bits; }
    if ((tid & 63) == 0) sh.wtot[tid >> 6] = (int)bits;
    __syncthreads();
    if (tid == 0) {
        unsigned m = 0;
        for (int i = 0; i < RB / 64; i++) { const unsigned v = (unsigned)sh.wtot[i]; m = v > m ? v : m; }
        for (int i = 0; i < 8; i++) out->H[i] = sh.w[i];
        out->H[8] = __uint_as_float(m);
        out->n_in = cnt;
        out->ok = 1;
        out->_pad = sh.fb;
    }
}

// MODE 0: the batched live path (n <= 400, everything in LDS); 1: one pair with 400 < n <= 4096 (points in LDS, Gauss-Newton work arrays in
// HBM); 2: one pair with more points than the LDS holds (points in HBM too: every lane of a wave reads the same point, one request)
// LISTG: the list of accepted draws and their supports live in HBM instead of in front of the points in LDS.  With sample_times near 5000 the
// list is 20 KB; on top of the 60.8 KB body (n up to 400) the workgroup would pass 80 KB and a CU (160 KB) would hold ONE workgroup instead of
// two.  The list is touched once per draw (a 2-byte read, a 2-byte write), nothing of the hot arithmetic moves.  The live path
// (sample_times 1000: 4 KB of list) keeps the LDS form, instantiated apart so that its list accesses stay ds_ instructions.
template <int MODE, bool LISTG = false>
__device__ __forceinline__ void ransac_body(const RansacArgs& a) {
    constexpr bool BIG = MODE >= 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // 16-byte aligned whatever the static LDS in front of it adds up to: the float4 point copies are read with ds_read_b128 (at 8 mod 16 the support loop took twice as long)
    __shared__ RShared sh;

    const int pair = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n[pair];
    mi355_pair_result* out = a.out + pair;
    const mi355_sfpoint* P1 = a.p1 + (size_t)pair * a.stride;
    const mi355_sfpoint* P2 = a.p2 + (size_t)pair * a.stride;

    if (tid < 9) out->H[tid] = 0.0f;
    if (tid == 0) { sh.fb = 0; sh.npol = 0; sh.next = 0; out->_pad = 0; }
    if (n < 4 || a.sample_times < 1) {                     // mosaicimage.h:1739-1761
        if (tid == 0) { out->n_in = 0; out->ok = 0; }
        return;
    }
    uint16_t* list = LISTG ? a.list_hbm + (size_t)pair * 2 * (size_t)a.list_floats : reinterpret_cast<uint16_t*>(lds);   // draws that hold a hypothesis slot, in draw order
    float* x1 = MODE == 2 ? a.big_pts : lds + (LISTG ? 0 : a.list_floats);      // targets (image i)
    float* y1 = x1 + n;
    float* x2 = y1 + n;         // sources (image j)
    float* y2 = x2 + n;
    float* J  = BIG ? a.big_ws : y2 + n;         // 2n x 8   (BIG: in HBM, the points alone fill the LDS)
    float* JL = J + 16 * n;     // 8 x 2n
    float* C  = JL + 16 * n;    // 2n
    float* CS = C + 2 * n;      // BIG: 4n floats of compaction scratch
    // The support loop reads every point once per hypothesis: (x1, y1, x2, y2) interleaved, one 16-byte broadcast read per point.  The
    // copy lives where the closing Gauss-Newton later keeps J (unused until then; BIG keeps J in HBM and reads the four arrays).
    float4* pts = reinterpret_cast<float4*>(y2 + n);
    for (int i = tid; i < n; i += RB) {
        x1[i] = P1[i].x; y1[i] = P1[i].y; x2[i] = P2[i].x; y2[i] = P2[i].y;
        if constexpr (!BIG) pts[i] = make_float4(P1[i].x, P1[i].y, P2[i].x, P2[i].y);
    }
    if (tid < 8) sh.state[tid] = (tid == 2 || tid == 3) ? -1 : 0;
    if (tid < 9) { sh.bestH[tid] = 0.0f; sh.firstH[tid] = 0.0f; }
    __syncthreads();

    const float d2 = a.dist * a.dist;                      // :1757
    const float invn = 1.0f / (float)n;                    // :1763
    const int sample_times = a.sample_times > 5000 ? 5000 : a.sample_times;
    const uint16_t* table = a.tables + (size_t)(a.single_table ? 0 : (a.table_of ? a.table_of[pair] : (n - 4))) * MAX_DRAWS * 4;

    float h[9];
    int my_li = -1;
    long long T0 = wall_clock64(); int nchunk = 0; long long Tsolve = 0, Tsup = 0;
    // ---- pass 1: which draws hold a hypothesis slot (classify_draw) ----
    int nlist = 0;
    const int list_cap = sample_times < MAX_DRAWS ? sample_times : MAX_DRAWS;
    for (int base = 0; base < MAX_DRAWS && nlist < list_cap; base += RB) {
        const int r = base + tid;
        bool accepted = false;
        if (r < MAX_DRAWS) accepted = classify_draw(table, r, x1, y1, x2, y2, sh, tid);
        int tot;
        const int pos = nlist + block_exclusive_scan_flags(accepted, tid, sh.wtot, tot);
        if (accepted && pos < list_cap) list[pos] = (uint16_t)r;
        nlist += tot;
    }
    nlist = nlist < list_cap ? nlist : list_cap;
    __syncthreads();
    long long Tclass = wall_clock64() - T0;
    // ---- pass 2: hypotheses + supports of the accepted draws; then the sequential loop (mosaicimage.h:1864-1918) replayed once over all of them ----
    // Round 4 (late): the waves no longer meet after every 256 draws.  A wave takes the next 64 draws of the list from a counter, solves, polishes,
    // counts, and takes the next ones; the supports go to LDS, and a lane keeps the hypothesis of ITS OWN first maximum in LDS (a lane's draws
    // come in list order, so the loop's winner -- the first draw holding the maximum of the draws the loop looks at -- is its lane's first
    // maximum).  With a barrier per chunk the four waves of a workgroup, whose solve times differ by a quarter from chunk to chunk, waited for
    // the slowest one four times per pair: 14 000 of 80 000 ticks.  The loop's two stop rules are applied afterwards on the stored supports;
    // draws behind a stop have then been evaluated for nothing (a stop by the 0.99 ratio is rare, the slot limit is the list's length).
    uint16_t* sup = list + a.list_floats;                      // support of list entry i (<= n <= 65535)
    float* hb = lds + a.hb_off;                                // [9][RB]: the lane's own best hypothesis
    {
        const int lane = tid & 63;
        const int nsub = (nlist + 63) >> 6;
        int mybest = -1;
        my_li = -1;
        for (;;) {
            long long c0 = wall_clock64();
            int sc = 0;
            if (lane == 0) sc = atomicAdd(&sh.next, 1);
            sc = __shfl(sc, 0, 64);
            if (sc >= nsub) break;
            nchunk++;
            const int li = sc * 64 + lane;
            if (li < nlist) {
                const int support = eval_draw<BIG>(a, table, list[li], n, d2, x1, y1, x2, y2, pts, h, sh, tid, &Tsolve);
                sup[li] = (uint16_t)support;
                if (support > mybest) {
                    mybest = support; my_li = li;
#pragma unroll
                    for (int i = 0; i < 9; i++) hb[i * RB + tid] = h[i];
                }
                if (li == 0) { for (int i = 0; i < 9; i++) sh.firstH[i] = h[i]; }      // the first accepted draw (:1878-1884)
            }
            Tsup += wall_clock64() - c0;
        }
    }
    __syncthreads();
    {
        int M, win;
        replay_supports(sup, nlist, invn, sh, tid, M, win);
        if (tid == 0) { sh.state[2] = M > 0 ? win : -1; sh.state[5] = 0; }
        __syncthreads();
        if (M > 0) {
            if (my_li == win) { for (int i = 0; i < 9; i++) sh.bestH[i] = hb[i * RB + tid]; sh.state[5] = 1; }
            __syncthreads();
            // the lane that evaluated the winner went on to a larger support behind a stop: the winner's hypothesis is formed again
            if (!sh.state[5] && tid == 0) recompute_hypothesis(table, list[win], x1, y1, x2, y2, sh);
        }
        __syncthreads();
    }
    long long T1 = wall_clock64();
    // winner: hyp[maxSupportIndex]; maxSupportIndex stays 0 when no support was ever positive (:1783) -> first accepted
    float W[9];
#pragma unroll
    for (int i = 0; i < 9; i++) W[i] = (sh.state[2] >= 0) ? sh.bestH[i] : sh.firstH[i];

    const int cnt = split_inliers<BIG>(a, out, P1, P2, n, d2, W, x1, y1, x2, y2, C, sh, tid);
    if (cnt < 4) {                                          // :1953-1961 refit fails below 4 -> no H; :2024-2032
        if (tid == 0) { out->n_in = cnt; out->ok = 0; }
        return;
    }
    if (cnt <= a.min_keep) {                                // match_pairs: MosaicWithoutPos.cpp:5201 drops the pair (n_in <= 30), its H is never looked at
        if (tid == 0) { out->n_in = cnt; out->ok = 0; }
        if (a.dbg && tid == 0) { long long* d = a.dbg + 8 * pair; d[0] = T1 - T0; d[3] = nchunk; d[4] = Tsolve; d[5] = Tsup; d[6] = Tclass; d[7] = sh.npol; }
        return;
    }
    compact_inliers<BIG>(cnt, x1, y1, x2, y2, C, CS, tid);
    if (tid < 8) sh.w[tid] = W[tid];
    if (tid < 64) sh.T2[tid] = 0.0f;
    __syncthreads();

    long long T2 = wall_clock64();
    nlls_closing(cnt, x1, y1, x2, y2, J, JL, C, sh, tid);
    if (a.dbg && tid == 0) { long long* d = a.dbg + 8 * pair; d[0] = T1 - T0; d[1] = T2 - T1; d[2] = wall_clock64() - T2; d[3] = nchunk; d[4] = Tsolve; d[5] = Tsup; d[6] = Tclass; d[7] = sh.npol; }
    write_result(out, cnt, x1, y1, x2, y2, sh, tid);
}

// 2 waves per SIMD (256 registers each): with 1 (512 registers, spills in AGPRs instead of scratch) the kernel measured 30 % slower
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_kernel(RansacArgs a) { ransac_body<0>(a); }
// the same with the draw list in HBM: sample_times so large that list + body would pass half a CU's LDS
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_listg_kernel(RansacArgs a) { ransac_body<0, true>(a); }
// Ransac2D accepts any n (mosaicimage.h:1729-1761); the live path never exceeds 396, stand-alone callers may: up to 4096 points in LDS
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_big_kernel(RansacArgs a) { ransac_body<1>(a); }
// ... and beyond 4096 (up to the 65 535 a 16-bit draw table can address) with the points in HBM
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_huge_kernel(RansacArgs a) { ransac_body<2>(a); }

// ---- few pairs: a pair's draws spread over several workgroups -------------------------------------------------------------------------------
// One workgroup per pair is ~1.3 ms of latency whatever the number of pairs (0.7 ms of draws, 0.5 ms of closing refinement): a rank that owns
// 63 pairs of a survey, or a caller of mi355_ransac2d, waits that long with most of the chip idle.  With fewer pairs than wave slots the same
// stages run as three launches (the launch boundaries are the synchronisation):
//   classify  S workgroups per pair classify all 20 chunks of 256 draws between them (ballot masks to HBM);
//   evaluate  S workgroups per pair rebuild the list of accepted draws from the masks and take groups of 64 of them from ONE counter per pair
//             (a wave's draws still come in list order), supports and the lanes' own first maxima go to HBM;
//   finish    one workgroup per pair replays the sequential loop over the stored supports (replay_supports, the same code), fetches the
//             winner's hypothesis from the lane that holds it, splits the inliers and runs the closing refinement in its wide form.
// Every hypothesis, support and sum is formed by the same code as in ransac_kernel, so the records are the same bytes.
constexpr int NCHUNK = (MAX_DRAWS + RB - 1) / RB;      // 20
constexpr int SPLIT_MAX = 8;                           // workgroups per pair at most
struct SplitBufs {
    unsigned long long* masks;   // [pair][NCHUNK][RB / 64]
    uint16_t* list;              // [pair][list_stride]
    uint16_t* sup;               // [pair][list_stride]
    int* nlist;                  // [pair]
    int* next;                   // [pair]: the evaluate pass's group counter (zero before the launch)
    float* rec;                  // [pair][S * RB][10]: list index of the lane's first maximum (int bits; -1 none), its hypothesis
    float* firstH;               // [pair][9]
    int* fb;                     // [pair]: draws that took the generic solve (the record's _pad)
    long long* dbg;              // optional (MI355_RANSAC_SPLIT_DBG): per evaluate wave {start, end, groups taken, time to the list}
    int S, list_stride;
    int nclass;                  // chunks of 256 draws the classify pass covers (the evaluate pass classifies the rest itself in the rare case that they are needed)
};

__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_split_classify(RansacArgs a, SplitBufs b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ RShared sh;
    const int pair = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const int n = a.n[pair];
    if (n < 4 || a.sample_times < 1) return;
    const mi355_sfpoint* P1 = a.p1 + (size_t)pair * a.stride;
    const mi355_sfpoint* P2 = a.p2 + (size_t)pair * a.stride;
    float* x1 = lds; float* y1 = x1 + n; float* x2 = y1 + n; float* y2 = x2 + n;
    for (int i = tid; i < n; i += RB) { x1[i] = P1[i].x; y1[i] = P1[i].y; x2[i] = P2[i].x; y2[i] = P2[i].y; }
    __syncthreads();
    const uint16_t* table = a.tables + (size_t)(a.single_table ? 0 : (a.table_of ? a.table_of[pair] : (n - 4))) * MAX_DRAWS * 4;
    for (int chunk = part; chunk < b.nclass; chunk += b.S) {
        const int r = chunk * RB + tid;
        bool accepted = false;
        if (r < MAX_DRAWS) accepted = classify_draw(table, r, x1, y1, x2, y2, sh, tid);
        const unsigned long long m = __ballot(accepted);
        if ((tid & 63) == 0) b.masks[((size_t)pair * NCHUNK + chunk) * (RB / 64) + (tid >> 6)] = m;
    }
}

__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_split_evaluate(RansacArgs a, SplitBufs b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ RShared sh;
    __shared__ int s_pref[NCHUNK * (RB / 64) + 1];
    const int pair = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const long long t_start = wall_clock64();
    const int n = a.n[pair];
    if (n < 4 || a.sample_times < 1) return;
    const mi355_sfpoint* P1 = a.p1 + (size_t)pair * a.stride;
    const mi355_sfpoint* P2 = a.p2 + (size_t)pair * a.stride;
    uint16_t* list = reinterpret_cast<uint16_t*>(lds);
    float* x1 = lds + a.list_floats; float* y1 = x1 + n; float* x2 = y1 + n; float* y2 = x2 + n;
    float4* pts = reinterpret_cast<float4*>(y2 + n);
    float* hb = reinterpret_cast<float*>(pts + n);             // [9][RB]
    for (int i = tid; i < n; i += RB) {
        x1[i] = P1[i].x; y1[i] = P1[i].y; x2[i] = P2[i].x; y2[i] = P2[i].y;
        pts[i] = make_float4(P1[i].x, P1[i].y, P2[i].x, P2[i].y);
    }
    if (tid == 0) { sh.fb = 0; sh.npol = 0; }
    // the list of accepted draws from the masks: (chunk, wave) blocks in draw order.  The classify pass covered the first nclass chunks (twice the
    // list's length in draws: enough unless fewer than half of the draws hold a slot); if those do not fill the list, this workgroup classifies the
    // remaining chunks itself -- every workgroup of the pair does, and arrives at the same masks
    constexpr int NB = NCHUNK * (RB / 64);                       // 80 blocks of 64 draws
    __shared__ unsigned long long s_masks[NB];
    __shared__ int s_cnt[NB];
    const int sample_times = a.sample_times > 5000 ? 5000 : a.sample_times;
    const int list_cap = sample_times < MAX_DRAWS ? sample_times : MAX_DRAWS;
    const uint16_t* table = a.tables + (size_t)(a.single_table ? 0 : (a.table_of ? a.table_of[pair] : (n - 4))) * MAX_DRAWS * 4;
    {
        const unsigned long long* masks = b.masks + (size_t)pair * NB;
        const int nb1 = b.nclass * (RB / 64);
        if (tid < NB) { const unsigned long long m = tid < nb1 ? masks[tid] : 0ull; s_masks[tid] = m; s_cnt[tid] = __popcll(m); }
        __syncthreads();
        int have = 0;
        for (int q = 0; q < nb1; q++) have += s_cnt[q];          // (the same sum in every thread)
        if (have < list_cap && b.nclass < NCHUNK) {
            for (int chunk = b.nclass; chunk < NCHUNK; chunk++) {
                const int r = chunk * RB + tid;
                bool accepted = false;
                if (r < MAX_DRAWS) accepted = classify_draw(table, r, x1, y1, x2, y2, sh, tid);
                const unsigned long long m = __ballot(accepted);
                if (lane == 0) { s_masks[chunk * (RB / 64) + (tid >> 6)] = m; s_cnt[chunk * (RB / 64) + (tid >> 6)] = __popcll(m); }
            }
            __syncthreads();
        }
        if (tid <= NB) { int acc = 0; for (int q = 0; q < tid; q++) acc += s_cnt[q]; s_pref[tid] = acc; }
    }
    __syncthreads();
    int nlist = s_pref[NB];
    nlist = nlist < list_cap ? nlist : list_cap;
    for (int chunk = 0; chunk < NCHUNK; chunk++) {
        const int q = chunk * (RB / 64) + (tid >> 6);
        if (s_pref[q] >= list_cap) break;                       // (later blocks only start further on)
        const unsigned long long m = s_masks[q];
        if ((m >> lane) & 1ull) { const int pos = s_pref[q] + __popcll(m & ((1ull << lane) - 1ull)); if (pos < list_cap) list[pos] = (uint16_t)(chunk * RB + tid); }
    }
    __syncthreads();
    if (part == 0) {
        for (int i = tid; i < nlist; i += RB) b.list[(size_t)pair * b.list_stride + i] = list[i];
        if (tid == 0) b.nlist[pair] = nlist;
    }
    const float d2 = a.dist * a.dist;
    const int nsub = (nlist + 63) >> 6;
    int mybest = -1, my_li = -1;
    long long Tsolve = 0;
    float h[9];
    int ngroups = 0;
    for (;;) {
        int sc = 0;
        if (lane == 0) sc = atomicAdd(&b.next[pair], 1);
        sc = __shfl(sc, 0, 64);
        if (sc >= nsub) break;
        ngroups++;
        const int li = sc * 64 + lane;
        if (li < nlist) {
            const int support = eval_draw<false>(a, table, list[li], n, d2, x1, y1, x2, y2, pts, h, sh, tid, &Tsolve);
            b.sup[(size_t)pair * b.list_stride + li] = (uint16_t)support;
            if (support > mybest) {
                mybest = support; my_li = li;
#pragma unroll
                for (int i = 0; i < 9; i++) hb[i * RB + tid] = h[i];
            }
            if (li == 0) { for (int i = 0; i < 9; i++) b.firstH[(size_t)pair * 9 + i] = h[i]; }
        }
    }
    if (b.dbg && lane == 0) { long long* d = b.dbg + (((size_t)pair * b.S + part) * (RB / 64) + (tid >> 6)) * 4; d[0] = t_start; d[1] = wall_clock64(); d[2] = ngroups; unsigned hw, xc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc)); d[3] = (long long)(((xc & 15u) << 16) | (hw & 0xffffu)); }
    float* rec = b.rec + ((size_t)pair * b.S * RB + (size_t)part * RB + tid) * 10;
    rec[0] = __int_as_float(my_li);
    if (my_li >= 0) { for (int i = 0; i < 9; i++) rec[1 + i] = hb[i * RB + tid]; }
    __syncthreads();
    if (tid == 0 && sh.fb) atomicAdd(&b.fb[pair], sh.fb);
}

__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_split_finish(RansacArgs a, SplitBufs b, int pq_stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ RShared sh;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n = a.n[pair];
    mi355_pair_result* out = a.out + pair;
    const mi355_sfpoint* P1 = a.p1 + (size_t)pair * a.stride;
    const mi355_sfpoint* P2 = a.p2 + (size_t)pair * a.stride;
    if (tid < 9) out->H[tid] = 0.0f;
    if (tid == 0) { sh.fb = b.fb[pair]; sh.npol = 0; out->_pad = 0; }
    if (n < 4 || a.sample_times < 1) {                     // mosaicimage.h:1739-1761
        if (tid == 0) { out->n_in = 0; out->ok = 0; }
        return;
    }
    const uint16_t* sup = b.sup + (size_t)pair * b.list_stride;      // read once by the replay: straight from HBM
    float* x1 = lds; float* y1 = x1 + n; float* x2 = y1 + n; float* y2 = x2 + n;
    float* J = y2 + n;                                     // 2n x 8
    float* C = J + 16 * n;                                 // 2n
    float* PQ = lds + ((22 * n + 3) & ~3);                 // 36 x pq_stride, 16-byte aligned
    const int nlist = b.nlist[pair];
    for (int i = tid; i < n; i += RB) { x1[i] = P1[i].x; y1[i] = P1[i].y; x2[i] = P2[i].x; y2[i] = P2[i].y; }
    if (tid < 8) sh.state[tid] = (tid == 2 || tid == 3) ? -1 : 0;
    if (tid < 9) { sh.bestH[tid] = 0.0f; sh.firstH[tid] = nlist > 0 ? b.firstH[(size_t)pair * 9 + tid] : 0.0f; }
    __syncthreads();
    const float d2 = a.dist * a.dist;                      // :1757
    const float invn = 1.0f / (float)n;                    // :1763
    const uint16_t* table = a.tables + (size_t)(a.single_table ? 0 : (a.table_of ? a.table_of[pair] : (n - 4))) * MAX_DRAWS * 4;
    {
        int M, win;
        replay_supports(sup, nlist, invn, sh, tid, M, win);
        if (tid == 0) { sh.state[2] = M > 0 ? win : -1; sh.state[5] = 0; }
        __syncthreads();
        if (M > 0) {
            const float* rec = b.rec + (size_t)pair * b.S * RB * 10;
            for (int q = tid; q < b.S * RB; q += RB)
                if (__float_as_int(rec[(size_t)q * 10]) == win) { for (int i = 0; i < 9; i++) sh.bestH[i] = rec[(size_t)q * 10 + 1 + i]; sh.state[5] = 1; }      // at most one lane evaluated it
            __syncthreads();
            if (!sh.state[5] && tid == 0) recompute_hypothesis(table, b.list[(size_t)pair * b.list_stride + win], x1, y1, x2, y2, sh);
        }
        __syncthreads();
    }
    float W[9];
#pragma unroll
    for (int i = 0; i < 9; i++) W[i] = (sh.state[2] >= 0) ? sh.bestH[i] : sh.firstH[i];
    const int cnt = split_inliers<false>(a, out, P1, P2, n, d2, W, x1, y1, x2, y2, C, sh, tid);
    if (cnt < 4 || cnt <= a.min_keep) {                    // :1953-1961 / :2024-2032; MosaicWithoutPos.cpp:5201 (see ransac_body)
        if (tid == 0) { out->n_in = cnt; out->ok = 0; }
        return;
    }
    compact_inliers<false>(cnt, x1, y1, x2, y2, C, nullptr, tid);
    if (tid < 8) sh.w[tid] = W[tid];
    if (tid < 64) sh.T2[tid] = 0.0f;
    __syncthreads();
    nlls_closing_wide(cnt, x1, y1, x2, y2, J, PQ, pq_stride, C, sh, tid);
    write_result(out, cnt, x1, y1, x2, y2, sh, tid);
}

}  // namespace

// ---- host: glibc rand() (TYPE_3 additive feedback, stdlib/random_r.c) and the draw table ------------------
namespace {
struct GlibcRand {
    int32_t r[34]; int f, b;
    void seed(uint32_t s) {
        if (s == 0) s = 1;
        r[0] = (int32_t)s;
        for (int i = 1; i < 31; i++) {
            const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
            long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            r[i] = (int32_t)word;
        }
        f = 3; b = 0;
        for (int i = 0; i < 310; i++) (void)next();
    }
    int next() {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return (int)(v >> 1);
    }
};
// The rand() stream depends only on the seed; what depends on n is how it is consumed (% n, redraw of all four until
// pairwise distinct).  For the all-n tables the raw stream is generated once and one workgroup per n walks it.
//
// The stream itself is produced on the device.  rand() is the lagged sum x_n = x_{n-3} + x_{n-31} (mod 2^32), output x_n >> 1: linear
// in the 31 last values, so the state after K steps is A^K times the state before, A a fixed 31 x 31 matrix over Z / 2^32.  The host
// seeds the generator (344 steps) and hands over the 31 values; workgroup j of the kernel jumps to the state after 992 j steps by
// applying the precomputed powers A^(992 * 2^b) for the set bits of j (at most 9 matrix-vector products of 961 terms), then one
// lane runs the 992 steps of its block with the 31 values in registers.  The host loop over 400 000 rand() calls plus the 1.6 MB
// upload cost 1.4-2.2 ms per new seed (every survey has its own seed); this takes ~0.1 ms.  Same integers by construction; every
// RANSAC golden case and soak compares results that depend on every draw.
constexpr int RAW_BLK = 31 * 32;                 // values per workgroup: 32 trips of the 31-step ring
constexpr int RAW_STREAM = 404 * RAW_BLK;        // 400 768 values: n = 4 consumes ~213 k for 4999 draws (9.4 % of the groups are distinct)
constexpr int RAW_BITS = 9;                      // 404 blocks < 2^9
struct RawSeed { uint32_t s[31]; };
__global__ __launch_bounds__(64) void raw_stream_kernel(const uint32_t* powers /* RAW_BITS x 31 x 31 */, RawSeed seed, int* raw, int nraw) {
    __shared__ uint32_t st[2][32];
    __shared__ int outv[RAW_BLK];
    const int j = blockIdx.x, l = threadIdx.x;
    if (l < 31) st[0][l] = seed.s[l];
    __syncthreads();
    int cur = 0;
    for (int b = 0; b < RAW_BITS; b++) {
        if (!((j >> b) & 1)) continue;                    // uniform
        if (l < 31) {
            const uint32_t* m = powers + ((size_t)b * 31 + l) * 31;
            uint32_t acc = 0;
            for (int k = 0; k < 31; k++) acc += m[k] * st[cur][k];
            st[cur ^ 1][l] = acc;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (l == 0) {
        uint32_t x[31];
#pragma unroll
        for (int k = 0; k < 31; k++) x[k] = st[cur][k];   // x[k] = x_{n-31+k}: slot k is the oldest value when step k runs
        for (int t = 0; t < RAW_BLK / 31; t++) {
#pragma unroll
            for (int k = 0; k < 31; k++) {
                x[k] = x[k] + x[(k + 28) % 31];           // x_n = x_{n-31} + x_{n-3}
                outv[t * 31 + k] = (int)(x[k] >> 1);
            }
        }
    }
    __syncthreads();
    for (int i = l; i < RAW_BLK; i += 64) { const int g = j * RAW_BLK + i; if (g < nraw) raw[g] = outv[i]; }
}
// powers of the step matrix: P[b] = A^(RAW_BLK * 2^b); S' = A S with S'[k] = S[k + 1] (k < 30), S'[30] = S[0] + S[28]
struct Mat31 { uint32_t m[31][31]; };
static void mat31_mul(const Mat31& a, const Mat31& b, Mat31& c) {
    for (int i = 0; i < 31; i++)
        for (int j = 0; j < 31; j++) { uint32_t acc = 0; for (int k = 0; k < 31; k++) acc += a.m[i][k] * b.m[k][j]; c.m[i][j] = acc; }
}
static const std::vector<uint32_t>& raw_stream_powers() {
    static const std::vector<uint32_t> P = [] {
        Mat31 A; memset(&A, 0, sizeof(A));
        for (int k = 0; k < 30; k++) A.m[k][k + 1] = 1;
        A.m[30][0] = 1; A.m[30][28] = 1;
        Mat31 R; memset(&R, 0, sizeof(R));
        for (int k = 0; k < 31; k++) R.m[k][k] = 1;
        Mat31 B = A, T;
        for (int e = RAW_BLK; e; e >>= 1) {               // R = A^RAW_BLK by binary powers
            if (e & 1) { mat31_mul(R, B, T); R = T; }
            mat31_mul(B, B, T); B = T;
        }
        std::vector<uint32_t> out((size_t)RAW_BITS * 31 * 31);
        for (int b = 0; b < RAW_BITS; b++) {
            memcpy(out.data() + (size_t)b * 961, &R, sizeof(R));
            mat31_mul(R, R, T); R = T;
        }
        return out;
    }();
    return P;
}
__global__ __launch_bounds__(256) void draw_tables_kernel(const int* raw, int nraw, uint16_t* tables, int n_lo, int* overflow) {
    // one workgroup per n: groups of four stream values are tested in parallel, the accepted ones (pairwise distinct
    // after % n) are compacted in stream order with a block-wide prefix count
    const int n = n_lo + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint16_t* out4 = tables + (size_t)blockIdx.x * MAX_DRAWS * 4;
    __shared__ int s_w[4];
    const int ngroups = nraw / 4;
    int done = 0;
    for (int g0 = 0; done < MAX_DRAWS; g0 += 256) {
        if (g0 >= ngroups) { if (tid == 0) *overflow = 1; return; }
        const int gi = g0 + tid;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        bool ok = false;
        if (gi < ngroups) {
            const int4 v = reinterpret_cast<const int4*>(raw)[gi];
            s0 = v.x % n; s1 = v.y % n; s2 = v.z % n; s3 = v.w % n;
            ok = !(s0 == s1 || s0 == s2 || s0 == s3 || s1 == s2 || s1 == s3 || s2 == s3);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        const int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wv] = __builtin_popcountll(m);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wv; q++) woff += s_w[q];
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const int pos = done + woff + before;
        if (ok && pos < MAX_DRAWS) { out4[4 * pos] = (uint16_t)s0; out4[4 * pos + 1] = (uint16_t)s1; out4[4 * pos + 2] = (uint16_t)s2; out4[4 * pos + 3] = (uint16_t)s3; }
        done += total;
        __syncthreads();
    }
}
}  // namespace

// mosaicimage.h:1777-1813: srand(seed); per draw: four rand()%n, all four redrawn until pairwise distinct
void mi_glibc_draw_table(uint32_t seed, int n, int max_draws, uint16_t* out4) {
    GlibcRand g;
    g.seed(seed);
    for (int d = 0; d < max_draws; d++) {
        int s[4];
        do { for (int i = 0; i < 4; i++) s[i] = g.next() % n; }
        while (s[0] == s[1] || s[0] == s[2] || s[0] == s[3] || s[1] == s[2] || s[1] == s[3] || s[2] == s[3]);
        for (int i = 0; i < 4; i++) out4[4 * d + i] = (uint16_t)s[i];
    }
}

// one pair with 400 < n <= MI355_RANSAC_BIG_MAX correspondences (host arrays in, host arrays out): mi355_ransac2d's large-n path
int mi_ransac_big(mi355_ctx* ctx, const mi355_sfpoint* p1, const mi355_sfpoint* p2, int n, float dist, int sample_times, uint32_t seed,
                  mi355_sfpoint* in1, mi355_sfpoint* in2, int* n_in, float* H, int* ok) {
    DevBuf& d1 = ctx->buf("rbig_p1"); DevBuf& d2 = ctx->buf("rbig_p2"); DevBuf& da = ctx->buf("rbig_a"); DevBuf& db = ctx->buf("rbig_b");
    DevBuf& dn = ctx->buf("r1_n"); DevBuf& dres = ctx->buf("pair_results"); DevBuf& dws = ctx->buf("rbig_ws"); DevBuf& dtab = ctx->buf("ransac_tables");
    const size_t pb = sizeof(mi355_sfpoint) * (size_t)n, one = (size_t)MAX_DRAWS * 4;
    MI_HIP(d1.reserve(pb)); MI_HIP(d2.reserve(pb)); MI_HIP(da.reserve(pb)); MI_HIP(db.reserve(pb));
    MI_HIP(dn.reserve(sizeof(int))); MI_HIP(dres.reserve(sizeof(mi355_pair_result))); MI_HIP(dws.reserve(sizeof(float) * 38 * (size_t)n));
    MI_HIP(dtab.reserve(one * sizeof(uint16_t)));
    std::vector<uint16_t> tab(one);
    mi_glibc_draw_table(seed, n, MAX_DRAWS, tab.data());
    MI_HIP(hipMemcpyAsync(d1.p, p1, pb, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(d2.p, p2, pb, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dn.p, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dtab.p, tab.data(), one * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemsetAsync(dres.p, 0, sizeof(mi355_pair_result), ctx->stream));
    RansacArgs a;
    memset(&a, 0, sizeof(a));
    a.p1 = d1.as<mi355_sfpoint>(); a.p2 = d2.as<mi355_sfpoint>(); a.n = dn.as<int>(); a.tables = dtab.as<uint16_t>(); a.table_of = nullptr;
    a.stride = n; a.dist = dist; a.sample_times = sample_times; a.min_keep = -1; a.out = dres.as<mi355_pair_result>();
    a.big_ws = dws.as<float>(); a.big_a = da.as<mi355_sfpoint>(); a.big_b = db.as<mi355_sfpoint>();
    a.single_table = 1;                                   // the one uploaded table, not table n - 4 of a set
    a.list_floats = (((sample_times < MAX_DRAWS ? (sample_times > 0 ? sample_times : 1) : MAX_DRAWS) + 7) / 8) * 8;      // uint16 list + uint16 supports
    if (n > RANSAC_LDS_POINTS) {                          // the points do not fit the LDS: HBM (ransac_huge_kernel)
        DevBuf& dpts = ctx->buf("rbig_pts");
        MI_HIP(dpts.reserve(sizeof(float) * 4 * (size_t)n));
        a.big_pts = dpts.as<float>();
        a.hb_off = a.list_floats;
        const size_t lds_bytes = ((size_t)a.list_floats + 9 * RB) * sizeof(float);
        hipLaunchKernelGGL(ransac_huge_kernel, dim3(1), dim3(RB), lds_bytes, ctx->stream, a);
    } else {
        a.hb_off = a.list_floats + 4 * n;
        const size_t lds_bytes = ((size_t)4 * n + a.list_floats + 9 * RB) * sizeof(float);
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(ransac_big_kernel, dim3(1), dim3(RB), lds_bytes, ctx->stream, a);
    }
    MI_HIP(hipGetLastError());
    mi355_pair_result r;
    MI_HIP(hipMemcpyAsync(&r, dres.p, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));            // `tab` goes out of scope; r has landed
    *n_in = r.n_in; *ok = r.ok;
    memcpy(H, r.H, sizeof(float) * 9);
    if (r.n_in > 0) {
        if (in1) MI_HIP(hipMemcpyAsync(in1, da.p, sizeof(mi355_sfpoint) * (size_t)r.n_in, hipMemcpyDeviceToHost, ctx->stream));
        if (in2) MI_HIP(hipMemcpyAsync(in2, db.p, sizeof(mi355_sfpoint) * (size_t)r.n_in, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MI355_OK;
}

// Draw tables of every n in [4, 400] for `seed`, indexed by n - 4 (397 x 40 KB = 15.9 MB): built once per seed on the ctx's side stream
// -- the device generates the rand() stream itself (raw_stream_kernel) -- and kept in one of four slots that are reused in turn, so that a
// caller whose every survey has its own seed (the reference seeds from the clock) neither allocates nor waits: mi_match_pairs_dev asks for the
// tables BEFORE it enqueues the matcher, the build runs beside it, and the RANSAC launch only waits for the slot's event on the device.  The
// "stream too short" flag of a build (never seen: the stream is twice what n = 4 needs) lands in pinned memory and is looked at when the slot is
// used again.  d_tables == NULL: prefetch only; otherwise the ctx stream is made to wait for the slot's event.
// The jump-ahead powers of the rand() recurrence: uploaded ONCE, with a blocking copy -- both users (the table build on the side stream, the
// diagnostic on the ctx stream) then find them complete whichever comes first; the two streams are not ordered with each other.
static int upload_raw_powers(mi355_ctx* ctx, DevBuf& dpow) {
    const std::vector<uint32_t>& P = raw_stream_powers();
    MI_HIP(dpow.reserve(P.size() * sizeof(uint32_t)));
    MI_HIP(hipMemcpy(dpow.p, P.data(), P.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return MI355_OK;
}

// the (never seen) "rand() stream too short" verdict of a finished build: the slot's tables are rebuilt on the host, the call that used them is void
static int check_draw_slot(mi355_ctx* ctx, int si) {
    mi355_ctx::DrawTables& t = ctx->draw_tables[si];
    if (!t.valid || !t.ready || !ctx->draw_flags[si]) return MI355_OK;
    const size_t one = (size_t)MAX_DRAWS * 4;
    const int ntab = MI355_MAX_SELECTED - 4 + 1;
    MI_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint16_t> tabs(one * ntab);
    for (int n = 4; n <= MI355_MAX_SELECTED; n++) mi_glibc_draw_table(t.seed, n, MAX_DRAWS, tabs.data() + one * (n - 4));
    MI_HIP(hipMemcpy(t.buf.p, tabs.data(), tabs.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    ctx->draw_flags[si] = 0;
    ctx->set_error("ransac: the draw tables of an earlier call (seed " + std::to_string(t.seed) + ") were incomplete (rand() stream too short); they have been rebuilt, results of that call are void");
    return MI355_ERR_FAILED;
}

int mi_ransac_tables(mi355_ctx* ctx, uint32_t seed, const uint16_t** d_tables) {
    const size_t one = (size_t)MAX_DRAWS * 4;
    const int ntab = MI355_MAX_SELECTED - 4 + 1;
    if (!ctx->aux_stream) {
        MI_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        MI_HIP(hipEventCreateWithFlags(&ctx->aux_ev, hipEventDisableTiming));
        MI_HIP(hipHostMalloc((void**)&ctx->draw_flags, 4 * sizeof(int), hipHostMallocDefault));
        for (int i = 0; i < 4; i++) ctx->draw_flags[i] = 0;
    }
    // every finished build's verdict is looked at on every call, whichever seed this call is about: a flag is never lost to a slot's reuse
    // (it used to be read only when the SAME seed came again, and was overwritten when the slot was taken for another one)
    for (int si = 0; si < 4; si++)
        if (ctx->draw_tables[si].valid && ctx->draw_tables[si].ready && hipEventQuery(ctx->draw_tables[si].ready) == hipSuccess) { const int rc = check_draw_slot(ctx, si); if (rc != MI355_OK) return rc; }
    mi355_ctx::DrawTables* slot = nullptr;
    for (auto& t : ctx->draw_tables) if (t.valid && t.seed == seed) slot = &t;
    if (!slot) {
        for (auto& t : ctx->draw_tables) if (!t.valid) { slot = &t; break; }
        if (!slot) { slot = &ctx->draw_tables[0]; for (auto& t : ctx->draw_tables) if (t.used < slot->used) slot = &t; }      // the slot used longest ago
        const int si = (int)(slot - ctx->draw_tables);
        if (slot->valid && slot->ready) {                // its last build may still be running: its verdict is read before the slot changes hands
            MI_HIP(hipEventSynchronize(slot->ready));
            const int rc = check_draw_slot(ctx, si); if (rc != MI355_OK) return rc;
        }
        if (!slot->ready) MI_HIP(hipEventCreateWithFlags(&slot->ready, hipEventDisableTiming));
        MI_HIP(slot->buf.reserve(one * ntab * sizeof(uint16_t)));
        MI_HIP(slot->raw.reserve((size_t)RAW_STREAM * sizeof(int) + 64));
        // launches that still read the slot's old tables were enqueued on the ctx stream: the build starts behind them
        MI_HIP(hipEventRecord(ctx->aux_ev, ctx->stream));
        MI_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev, 0));
        RawSeed rs;
        { GlibcRand g; g.seed(seed); for (int k = 0; k < 31; k++) rs.s[k] = (uint32_t)g.r[(g.f + k) % 31]; }     // x_{-31} .. x_{-1}: slot f is overwritten next
        int* d_flag = slot->raw.as<int>() + RAW_STREAM;
        DevBuf& dpow = ctx->buf("ransac_raw_powers");
        if (dpow.cap == 0) { const int rc = upload_raw_powers(ctx, dpow); if (rc != MI355_OK) return rc; }
        hipLaunchKernelGGL(raw_stream_kernel, dim3(RAW_STREAM / RAW_BLK), dim3(64), 0, ctx->aux_stream, dpow.as<uint32_t>(), rs, slot->raw.as<int>(), RAW_STREAM);
        MI_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), ctx->aux_stream));
        hipLaunchKernelGGL(draw_tables_kernel, dim3(ntab), dim3(256), 0, ctx->aux_stream, slot->raw.as<int>(), RAW_STREAM, slot->buf.as<uint16_t>(), 4, d_flag);
        MI_HIP(hipGetLastError());
        MI_HIP(hipMemcpyAsync(&ctx->draw_flags[si], d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->aux_stream));
        MI_HIP(hipEventRecord(slot->ready, ctx->aux_stream));
        slot->seed = seed; slot->valid = true;
    }
    slot->used = ++ctx->draw_clock;
    if (d_tables) { MI_HIP(hipStreamWaitEvent(ctx->stream, slot->ready, 0)); *d_tables = slot->buf.as<uint16_t>(); }      // (a prefetch does not make the ctx stream wait)
    return MI355_OK;
}

int mi_ransac_batch(mi355_ctx* ctx, const mi355_sfpoint* d_p1, const mi355_sfpoint* d_p2, const int* d_n, const int* h_n,
                    int n_pairs, int stride, float dist, int sample_times, uint32_t seed, mi355_pair_result* d_out, int min_keep) {
    if (n_pairs <= 0) return MI355_OK;
    const size_t one = (size_t)MAX_DRAWS * 4;
    const uint16_t* d_tables = nullptr;
    const int* d_table_of = nullptr;
    int nmax = MI355_MAX_SELECTED;
    if (h_n) {
        // host knows every n: one draw table per distinct n (the stream itself depends only on the seed)
        std::map<int, int> table_idx;
        nmax = 4;
        for (int i = 0; i < n_pairs; i++) {
            const int n = h_n[i];
            if (n > MI355_MAX_SELECTED) { ctx->set_error("ransac: more than 400 correspondences in one pair"); return MI355_ERR_ARG; }
            if (n >= 4 && !table_idx.count(n)) { const int id = (int)table_idx.size(); table_idx[n] = id; }
            if (n > nmax) nmax = n;
        }
        DevBuf& dtab = ctx->buf("ransac_tables");
        DevBuf& dof = ctx->buf("ransac_table_of");
        std::vector<uint16_t> tabs(one * (table_idx.empty() ? 1 : table_idx.size()));
        for (auto& kv : table_idx) mi_glibc_draw_table(seed, kv.first, MAX_DRAWS, tabs.data() + one * kv.second);
        std::vector<int> tof(n_pairs);
        for (int i = 0; i < n_pairs; i++) tof[i] = h_n[i] >= 4 ? table_idx[h_n[i]] : 0;
        MI_HIP(dtab.reserve(tabs.size() * sizeof(uint16_t)));
        MI_HIP(dof.reserve(sizeof(int) * n_pairs));
        MI_HIP(hipMemcpyAsync(dtab.p, tabs.data(), tabs.size() * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(hipMemcpyAsync(dof.p, tof.data(), sizeof(int) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));              // host vectors go out of scope
        d_tables = dtab.as<uint16_t>(); d_table_of = dof.as<int>();
    } else {
        // n is only known on the device (output of the selection kernel): the tables of every n in [4, 400] for this seed
        const int rc = mi_ransac_tables(ctx, seed, &d_tables);
        if (rc != MI355_OK) return rc;
    }
    RansacArgs a;
    memset(&a, 0, sizeof(a));
    a.p1 = d_p1; a.p2 = d_p2; a.n = d_n; a.tables = d_tables; a.table_of = d_table_of;
    a.dbg = nullptr; a.min_keep = min_keep;
    static const bool dbg_on = getenv("MI355_RANSAC_DBG") != nullptr;
    DevBuf& ddbg = ctx->buf("ransac_dbg");
    if (dbg_on) { MI_HIP(ddbg.reserve((size_t)n_pairs * 64)); MI_HIP(hipMemsetAsync(ddbg.p, 0, (size_t)n_pairs * 64, ctx->stream)); a.dbg = ddbg.as<long long>(); }
    a.stride = stride; a.dist = dist; a.sample_times = sample_times; a.out = d_out;
    a.list_floats = (((sample_times < MAX_DRAWS ? (sample_times > 0 ? sample_times : 1) : MAX_DRAWS) + 7) / 8) * 8;      // uint16 list + uint16 supports
    // the lanes' best hypotheses (9 x RB floats) lie at the end of the allocation, where J* and C are kept after the draw loop (points + their
    // float4 copy take the first 8 n floats)
    const size_t body_floats = (size_t)38 * nmax > (size_t)8 * nmax + 9 * RB ? (size_t)38 * nmax : (size_t)8 * nmax + 9 * RB;
    // two workgroups per CU need at most 80 KB each, static LDS (RShared, ~2 KB) included: beyond that the draw list moves to HBM
    const bool listg = (body_floats + a.list_floats) * sizeof(float) + 2048 > 80 * 1024;
    a.hb_off = (int)((listg ? 0 : a.list_floats) + body_floats - 9 * RB);
    const size_t lds_bytes = (body_floats + (listg ? 0 : a.list_floats)) * sizeof(float);
    if (listg) {
        DevBuf& dl = ctx->buf("ransac_list_hbm");
        MI_HIP(dl.reserve(sizeof(uint16_t) * 2 * (size_t)a.list_floats * (size_t)n_pairs));
        a.list_hbm = dl.as<uint16_t>();
    }
    if (lds_bytes > 48 * 1024) {
        MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(listg ? ransac_listg_kernel : ransac_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    // records start from zero: the inlier slots beyond n_in, H / ok of pairs that stop early and the padding are then the same bytes
    // on every run and every rank (records are compared and all-gathered as bytes)
    MI_HIP(hipMemsetAsync(d_out, 0, sizeof(mi355_pair_result) * (size_t)n_pairs, ctx->stream));
    // few pairs (a rank's share of a strip survey, a single mi355_ransac2d call): several workgroups per pair, three launches
    int S = ctx->ransac_split;                          // option "ransac_split": -1 = by the number of pairs, 0 = never, k = k workgroups per pair
    if (S < 0) { S = (2 * ctx->num_cu) / n_pairs; if (S < 2) S = 0; if (S > 4) S = 4; }      // 4 x 4 waves take the 16 groups of a 1000-draw list at once; more only idle
    if (S > SPLIT_MAX) S = SPLIT_MAX;
    if (S >= 1 && !dbg_on && !listg) {
        SplitBufs b;
        memset(&b, 0, sizeof(b));
        b.S = S; b.list_stride = (a.list_floats + 7) & ~7;
        {
            const int cap = sample_times < MAX_DRAWS ? (sample_times > 0 ? sample_times : 1) : MAX_DRAWS;
            b.nclass = (2 * cap + RB - 1) / RB;
            if (b.nclass < S) b.nclass = S;
            if (b.nclass > NCHUNK) b.nclass = NCHUNK;
        }
        const size_t o_masks = 0, o_list = o_masks + sizeof(unsigned long long) * NCHUNK * (RB / 64) * (size_t)n_pairs, o_sup = o_list + sizeof(uint16_t) * b.list_stride * (size_t)n_pairs,
                     o_nlist = o_sup + sizeof(uint16_t) * b.list_stride * (size_t)n_pairs, o_next = o_nlist + sizeof(int) * (size_t)n_pairs, o_first = o_next + sizeof(int) * (size_t)n_pairs,
                     o_fb = o_first + sizeof(float) * 9 * (size_t)n_pairs, o_rec = (o_fb + sizeof(int) * (size_t)n_pairs + 15) & ~(size_t)15, total = o_rec + sizeof(float) * 10 * S * RB * (size_t)n_pairs;
        DevBuf& dsp = ctx->buf("ransac_split");
        MI_HIP(dsp.reserve(total));
        uint8_t* base = dsp.as<uint8_t>();
        b.masks = reinterpret_cast<unsigned long long*>(base + o_masks); b.list = reinterpret_cast<uint16_t*>(base + o_list); b.sup = reinterpret_cast<uint16_t*>(base + o_sup);
        b.nlist = reinterpret_cast<int*>(base + o_nlist); b.next = reinterpret_cast<int*>(base + o_next); b.firstH = reinterpret_cast<float*>(base + o_first); b.fb = reinterpret_cast<int*>(base + o_fb);
        b.rec = reinterpret_cast<float*>(base + o_rec);
        MI_HIP(hipMemsetAsync(base + o_nlist, 0, o_rec - o_nlist, ctx->stream));      // list lengths, group counters, first hypotheses
        static const bool split_dbg = getenv("MI355_RANSAC_SPLIT_DBG") != nullptr;
        if (split_dbg) { DevBuf& dd = ctx->buf("ransac_split_dbg"); MI_HIP(dd.reserve((size_t)n_pairs * S * (RB / 64) * 32)); MI_HIP(hipMemsetAsync(dd.p, 0, (size_t)n_pairs * S * (RB / 64) * 32, ctx->stream)); b.dbg = dd.as<long long>(); }
        ProfScope ps(ctx, "ransac", (double)n_pairs * (24.0 * nmax + sizeof(mi355_pair_result)));
        const size_t lds_c = sizeof(float) * 4 * (size_t)nmax;
        const size_t lds_e = sizeof(float) * ((size_t)a.list_floats + 8 * (size_t)nmax + 9 * RB);
        int q4 = (2 * nmax + 3) / 4; if (!(q4 & 1)) q4++;                             // rows of the product table: a multiple of 4 floats whose quarter is odd
        const int pq_stride = 4 * q4;
        const size_t lds_f = sizeof(float) * (22 * (size_t)nmax + 4 + 36 * (size_t)pq_stride);      // points, J, C, the product table
        if (lds_e > 48 * 1024) MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_split_evaluate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e));
        if (lds_f > 48 * 1024) MI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_split_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
        hipLaunchKernelGGL(ransac_split_classify, dim3(S, n_pairs), dim3(RB), lds_c, ctx->stream, a, b);
        hipLaunchKernelGGL(ransac_split_evaluate, dim3(S, n_pairs), dim3(RB), lds_e, ctx->stream, a, b);
        hipLaunchKernelGGL(ransac_split_finish, dim3(n_pairs), dim3(RB), lds_f, ctx->stream, a, b, pq_stride);
        if (split_dbg) {
            std::vector<long long> hd((size_t)n_pairs * S * (RB / 64) * 4);
            MI_HIP(hipMemcpyAsync(hd.data(), b.dbg, hd.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            MI_HIP(hipStreamSynchronize(ctx->stream));
            long long t0 = -1, t1 = 0;
            for (size_t q = 0; q < hd.size(); q += 4) { if (hd[q] == 0) continue; if (t0 < 0 || hd[q] < t0) t0 = hd[q]; if (hd[q + 1] > t1) t1 = hd[q + 1]; }
            // busy waves: duration by XCD, and by the number of busy waves that share their CU
            std::map<int, std::pair<double, int>> byx, bycu;
            std::map<long long, int> cu_busy;
            for (size_t q = 0; q < hd.size(); q += 4) if (hd[q + 2] > 0) { const long long id = hd[q + 3]; cu_busy[(id >> 16) * 4096 + ((id >> 8) & 0xff)]++; }      // (xcd, se | cu)
            for (size_t q = 0; q < hd.size(); q += 4) {
                if (hd[q + 2] == 0) continue;
                const long long id = hd[q + 3]; const double dur = (double)(hd[q + 1] - hd[q]);
                auto& a1 = byx[(int)(id >> 16)]; a1.first += dur; a1.second++;
                auto& a2 = bycu[cu_busy[(id >> 16) * 4096 + ((id >> 8) & 0xff)]]; a2.first += dur; a2.second++;
            }
            fprintf(stderr, "[ransac split dbg, 100 MHz ticks] %d pairs x %d workgroups, evaluate pass %lld ticks; busy waves by XCD (n, mean ticks):", n_pairs, S, t1 - t0);
            for (auto& kv : byx) fprintf(stderr, " x%d:%d,%.0f", kv.first, kv.second.second, kv.second.first / kv.second.second);
            fprintf(stderr, " | by busy waves on the same CU:");
            for (auto& kv : bycu) fprintf(stderr, " %d:%d,%.0f", kv.first, kv.second.second, kv.second.first / kv.second.second);
            int dh[16] = {0}; double dmax = 0; long long slow_id = 0; long long slow_q = 0;
            for (size_t q = 0; q < hd.size(); q += 4) if (hd[q + 2] > 0) { const double dur = (double)(hd[q + 1] - hd[q]); int bk = (int)(dur / 10000.0); dh[bk > 15 ? 15 : bk]++; if (dur > dmax) { dmax = dur; slow_id = hd[q + 3]; slow_q = (long long)(q / 4); } }
            fprintf(stderr, " | durations in steps of 10000 ticks:");
            for (int i = 0; i < 16; i++) fprintf(stderr, " %d", dh[i]);
            fprintf(stderr, " | slowest: wave %lld (pair %lld part %lld) on xcd %lld hw_id 0x%llx, %.0f ticks", slow_q, slow_q / (S * 4), (slow_q / 4) % S, slow_id >> 16, slow_id & 0xffff, dmax);
            fprintf(stderr, "\n");
        }
    } else {
        ProfScope ps(ctx, "ransac", (double)n_pairs * (24.0 * nmax + sizeof(mi355_pair_result)));
        if (listg) hipLaunchKernelGGL(ransac_listg_kernel, dim3(n_pairs), dim3(RB), lds_bytes, ctx->stream, a);
        else hipLaunchKernelGGL(ransac_kernel, dim3(n_pairs), dim3(RB), lds_bytes, ctx->stream, a);
    }
    MI_HIP(hipGetLastError());
    if (dbg_on) {
        std::vector<long long> hd((size_t)n_pairs * 8);
        MI_HIP(hipMemcpyAsync(hd.data(), ddbg.p, hd.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        MI_HIP(hipStreamSynchronize(ctx->stream));
        double acc[8] = {0};
        double redo_j = 0, redo_i = 0;
        for (int i = 0; i < n_pairs; i++) {
            for (int k = 0; k < 7; k++) acc[k] += (double)hd[(size_t)i * 8 + k];
            const long long v = hd[(size_t)i * 8 + 7];
            acc[7] += (double)(v & 0xfff); redo_j += (double)((v >> 12) & 0x3ff); redo_i += (double)(v >> 22);
        }
        fprintf(stderr, "[ransac dbg, avg per pair, 100 MHz ticks] loop %.0f split %.0f nlls %.0f | chunks %.1f solve %.0f solve+support %.0f classify %.0f | polished draws %.1f, with a Jacobian / an inversion outside the division guard %.3f / %.3f\n", acc[0] / n_pairs, acc[1] / n_pairs, acc[2] / n_pairs, acc[3] / n_pairs, acc[4] / n_pairs, acc[5] / n_pairs, acc[6] / n_pairs, acc[7] / n_pairs, redo_j / n_pairs, redo_i / n_pairs);
    }
    return MI355_OK;
}

// diagnostic (tests/test_gpu_parity.py): the guarded shared-reciprocal division of hmath.h against the compiler's correctly rounded a / b,
// element by element: q_fast = hm::div_nr(a, b, hm::rcp_nr(b)), ok = the guard's verdict for this one quotient, q_true = a / b
namespace {
__global__ __launch_bounds__(256) void div_check_kernel(const float* a, const float* b, int n, float* q_fast, float* q_true, int* ok) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    hm::DivGuard g; hm::guard_init(g);
    const float r = hm::rcp_nr(b[i], g);
    q_fast[i] = hm::div_nr(a[i], b[i], r, g);
    ok[i] = hm::guard_ok(g) ? 1 : 0;
    q_true[i] = a[i] / b[i];
}
}  // namespace
extern "C" int mi355_debug_div(mi355_ctx* ctx, const float* a, const float* b, int n, float* q_fast, float* q_true, int32_t* ok) {
    if (!a || !b || !q_fast || !q_true || !ok || n <= 0) return MI355_ERR_ARG;
    LOCKED_PROLOGUE
    DevBuf& d = ctx->buf("debug_div");
    MI_HIP(d.reserve((size_t)n * 5 * sizeof(float)));
    float* da = d.as<float>(); float* db = da + n; float* dqf = db + n; float* dqt = dqf + n; int* dok = reinterpret_cast<int*>(dqt + n);
    MI_HIP(hipMemcpyAsync(da, a, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(db, b, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(div_check_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, da, db, n, dqf, dqt, dok);
    MI_HIP(hipGetLastError());
    MI_HIP(hipMemcpyAsync(q_fast, dqf, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(q_true, dqt, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(ok, dok, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

// diagnostic (tests/test_gpu_parity.py): InverseMatrix of order 8 with the pivot search, lane per matrix: the register routine (inverse8_pivot)
// and what the draw loop uses when a wave's LDS scratch is at hand (pivot_dispatch -> inverse8_wave for >= 32 active lanes).  Both start from
// the caller's `init` in the output (a matrix without a pivot leaves it untouched).
namespace {
// (every kernel of this file that reaches the out-of-line solvers carries the same waves-per-SIMD bound: a caller without one widens the
// callees' register budget, and with it the draw loop's kernels lose their second wave per SIMD -- 2.5 instead of 1.5 us per pair)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void inverse8_check_kernel(const float* mats, int n, float eps, const float* init, float* out_reg, float* out_wave) {
    __shared__ float s_wl[4][192];
    (void)s_wl;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;                                      // the last wave is partly active: what the draw loop's last group looks like
    float M[64], a[64], b[64];
#pragma unroll
    for (int k = 0; k < 64; k++) { M[k] = mats[(size_t)i * 64 + k]; a[k] = init[k]; b[k] = init[k]; }
    hm::pivot_call(M, a, eps);
#if defined(__HIP_DEVICE_COMPILE__)
    hm::pivot_dispatch(true, M, b, eps, s_wl[threadIdx.x >> 6]);
#endif
#pragma unroll
    for (int k = 0; k < 64; k++) { out_reg[(size_t)i * 64 + k] = a[k]; out_wave[(size_t)i * 64 + k] = b[k]; }
}
}  // namespace
extern "C" int mi355_debug_inverse8(mi355_ctx* ctx, const float* mats, int n, float eps, const float* init64, float* out_reg, float* out_wave) {
    if (!mats || !init64 || !out_reg || !out_wave || n <= 0) return MI355_ERR_ARG;
    LOCKED_PROLOGUE
    DevBuf& d = ctx->buf("debug_inv8");
    const size_t mb = sizeof(float) * 64 * (size_t)n;
    MI_HIP(d.reserve(3 * mb + 256));
    float* dm = d.as<float>(); float* da = dm + 64 * (size_t)n; float* db = da + 64 * (size_t)n; float* di = db + 64 * (size_t)n;
    MI_HIP(hipMemcpyAsync(dm, mats, mb, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(di, init64, sizeof(float) * 64, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(inverse8_check_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dm, n, eps, di, da, db);
    MI_HIP(hipGetLastError());
    MI_HIP(hipMemcpyAsync(out_reg, da, mb, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(out_wave, db, mb, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

// diagnostic (tests/test_gpu_parity.py): the device-generated rand() stream of a seed, for comparison with libc's
extern "C" int mi355_debug_rand_stream(mi355_ctx* ctx, uint32_t seed, int32_t* out, int n) {
    if (!out || n < 0 || n > RAW_STREAM) return MI355_ERR_ARG;
    LOCKED_PROLOGUE
    RawSeed rs;
    { GlibcRand g; g.seed(seed); for (int k = 0; k < 31; k++) rs.s[k] = (uint32_t)g.r[(g.f + k) % 31]; }
    DevBuf& draw = ctx->buf("ransac_raw_stream");
    MI_HIP(draw.reserve((size_t)RAW_STREAM * sizeof(int) + 64));
    DevBuf& dpow = ctx->buf("ransac_raw_powers");
    if (dpow.cap == 0) { const int rc = upload_raw_powers(ctx, dpow); if (rc != MI355_OK) return rc; }
    hipLaunchKernelGGL(raw_stream_kernel, dim3(RAW_STREAM / RAW_BLK), dim3(64), 0, ctx->stream, dpow.as<uint32_t>(), rs, draw.as<int>(), RAW_STREAM);
    MI_HIP(hipMemcpyAsync(out, draw.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

