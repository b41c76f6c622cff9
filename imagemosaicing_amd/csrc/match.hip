// csrc/match.hip -- K6 exact brute-force descriptor matching on the matrix cores + K7 sort / grid selection,
// and the per-pair orchestration match -> select -> RANSAC (gfx950).
//
// Replaces the j-loop body of GetMatchedPairsOneToAllSIFTThread, MosaicWithoutPos.cpp:5084-5232:
//   cv::FlannBasedMatcher().match (:5108-5110, approximate 1-NN)  -> exact 1-NN / 2-NN (what FLANN approximates)
//   std::sort(matches) (:5111)                                    -> total order (squared distance, queryIdx)
//   SelectMatchPairs grid walk (:4977-5028, call :5146-5153)      -> same walk, 64 matches per step with ballots
//   Ransac2D (:5169)                                              -> csrc/ransac.hip
//   accept when inliers > 30 (:5049, :5201)
//
// The distances are exact integers on the matrix cores: SIFT descriptors are the integers 0..255 (OpenCV stores
// saturate_cast<uchar> values in a float Mat); moved to v - 128 they are int8, and v_mfma_i32_32x32x32_i8 accumulates
// S = sum (q - 128)(t - 128) in int32 without rounding.  |q - t|^2 = |q'|^2 - (2 S - |t'|^2) for the shifted vectors q', t', so
// the nearest train row maximises x2 = 2 S - |t'|^2 and the arg-max (ties -> lowest train index) equals a CPU integer brute force.
// Round 2 did the same sum in bf16 (exact too, 8 significand bits); the int8 pipe is twice as fast (scratch/mfma_bench.hip, MI355X,
// descriptor-like operands: bf16 32x32x16 1.95 PFLOP/s, i8 32x32x32 3.9 POP/s -- both clock lower on toggling data than the
// 2.4 / 4.9 they reach on zeros) and its operands are half the bytes in HBM, L2 and LDS.
//
// Kernel shape (v_mfma_i32_32x32x32_i8): the TRAIN tile is the A operand (rows) and the QUERY tile the B operand (columns), so
// after the MFMA lane l holds 16 train rows of ONE query (column l & 31): the running best is lane-local (no cross-lane traffic
// in the loop) and the two half-waves are merged once at the end.  A wave owns 64 queries (two B operands): every A fragment and
// every row constant read from LDS feeds two MFMAs.
#include "common.h"
#include <algorithm>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// Pointers that arrive inside a structure read from memory (PairDesc) are generic to the compiler: it emits flat_load, and a flat
// load in flight turns every LDS wait into a wait for ALL memory.  Cast to the global address space: global_load, counted apart.
#define GLOBAL_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

constexpr int KSTRIDE = 2048;        // per-pair stride of the nn arrays (>= nfeatures rounded up); also the chunk size of the large-pair path
constexpr int BIG_KP_MAX = MI355_SIFT_KEEPALL_MAX;   // keypoints per image the large-pair path takes (keep-all frames)
constexpr int QTILE = 512;           // queries per workgroup (8 waves x 64): every staged train tile serves 512 queries
constexpr int BF_NT = 512;           // threads per workgroup
constexpr int ROWPAD = 256;          // descriptor matrices are padded to a multiple of this many rows (zeros)

struct PairDesc {
    const int8_t* s8_i; const int* n8_i; const float2* xy_i; int n_i; int npad_i;
    const int8_t* s8_j; const int* n8_j; const float2* xy_j; int n_j; int npad_j;
    int img_i, img_j, width, height;
};

// Work-to-XCD placement: workgroup ids go round-robin over the 8 XCDs (each with its own L2).  XCD x owns a contiguous eighth of
// the (pair, query tile) list, so the query tiles of a pair run side by side under ONE L2 and the train set crosses the fabric once.
__device__ __forceinline__ int xcd_owned(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ int med3_i32(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Where the time of the round-2 kernel went (scratch/match_time.py, 1740 pairs of 2000 x 2000; bf16, 32 queries per wave):
//   the eight MFMAs per tile alone, operands in registers       3.07 ms   (the pipe on these operands: 94 % of what mfma_bench gets)
//   + LDS operand reads                                         3.16 ms
//   + staging and one workgroup barrier per tile                4.09 ms
//   + per-element top-2 (fma, compare, select, med3, max)       5.96 ms   (0.36 of the nominal bf16 peak)
// and re-reading the rows of the best group after the loop to find the row costs more L2 traffic than the loop itself (tried:
// +1.2 ms).  Hence:
//   * the row position rides in the low bits of the compared value: z = 16 x2 + (15 - e) for slot e of the lane's 16 rows of the
//     tile (|x2| < 2^23, so z fits int32); ONE v_lshl_add_u32 per element builds it from the accumulator and a per-row constant
//     k = -16 |t'|^2 + 15 - e kept in LDS, a v_max3 tree finds the tile's maximum with its slot, and the running best needs four
//     more instructions per TILE (compare against best | 15 so that an equal distance in a later tile does not replace an earlier
//     row): 28 VALU per 16 elements instead of 96;
//   * the second-best distance is only needed by the optional ratio test and by mi355_bf_match: template parameter (two more
//     instructions per element, a multiset top-2 of z, whose order statistics map onto those of x2);
//   * three LDS stages: the first half of the NEXT tile's operands and its row constants are read before the barrier, so the
//     MFMAs of the next iteration start straight after it; the loop body has no branches.
// This kernel, same job: 2.50 ms = 2.1 POP/s (0.55 of the 3.9 the pipe sustains on these operands).  Its parts: MFMAs + LDS reads
// alone 1.72 ms, + staging and barrier 1.89, + epilogue 2.64, epilogues issued under the other query block's MFMAs 2.50.  What is
// left is VALU issue: scratch/valu_bench.hip puts v_lshl_add_u32, v_max3_i32 / _f32, v_med3, v_cmp and v_cvt at 4.3 cycles per
// wave64 instruction (half rate; only fma / add / or / mul run at 2.5), so the 58 epilogue instructions of a wave's tile take as
// long (250 cycles) as its eight MFMAs (256), and the four waves of a SIMD share both pipes.
template <bool SECOND>
__global__ __launch_bounds__(BF_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void bf_match_kernel(const PairDesc* pairs, int nqt, int* nn_idx, int* nn_d2, int* nn_2nd) {
    constexpr int NOROW = -(1 << 30);
    __shared__ __attribute__((aligned(16))) int s_k[KSTRIDE + 32];   // k of the train rows, NOROW for the padding rows
    constexpr int APITCH = 128 + 16;                      // bytes per staged row: 36 dwords keep the 16 lanes of a b128 group on distinct banks
    constexpr int STAGE = 32 * APITCH;
    __shared__ __attribute__((aligned(16))) int8_t s_a[3 * STAGE];
    const int work = xcd_owned(blockIdx.x, gridDim.x), pair = work / nqt;
    const PairDesc pd = pairs[pair];
    const int q_base = (work - pair * nqt) * QTILE;
    if (q_base >= pd.n_i) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    const int npad = pd.npad_j;
    for (int i = tid; i < npad + 32; i += BF_NT) {
        const int e = 4 * ((i & 31) >> 3) + (i & 3);      // slot of row i in its lane: C/D layout of the 32x32 MFMA, rows 8g + 4hi + r in acc[4g + r]
        s_k[i] = i < pd.n_j ? -16 * GLOBAL_PTR(int, pd.n8_j)[i] + 15 - e : NOROW;
    }
    int q[2];
    i32x4 bq[2][4];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        q[c] = q_base + wave * 64 + c * 32 + col;         // this lane's queries (rows beyond n_i are zero padding)
        const int q_ld = q[c] < pd.npad_i ? q[c] : pd.npad_i - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bq[c][ks] = *GLOBAL_PTR(i32x4, pd.s8_i + (size_t)q_ld * 128 + ks * 32 + hi * 16);
    }
    // The 32 x 128 train tile (4 KB) is the A operand of all eight waves: fetched once per workgroup into LDS, every thread 8 B.
    const int ld_off = (tid >> 4) * APITCH + (tid & 15) * 8;
    const int8_t* gsrc = pd.s8_j + (size_t)(tid >> 4) * 128 + (tid & 15) * 8;
    u32x2 pf;
    auto fetch = [&](int t0) { pf = *GLOBAL_PTR(u32x2, gsrc + (size_t)t0 * 128); };
    auto stage = [&](int buf, const u32x2& v) { *reinterpret_cast<u32x2*>(&s_a[buf * STAGE + ld_off]) = v; };
    fetch(0); stage(0, pf);
    fetch(32); stage(1, pf);                              // npad is a multiple of 256
    fetch(64);                                            // staged during iteration 0; every later fetch has a whole iteration to land
    __syncthreads();
    const int a_off = col * APITCH + hi * 16;
    i32x4 fa[4];
    i32x16 kA, kB;                                        // row constants of the even / odd tiles
    auto read_lo = [&](int buf, int t0, i32x16& kk) {     // operands ks 0..1 of the tile in `buf` and its row constants
#pragma unroll
        for (int ks = 0; ks < 2; ks++) fa[ks] = *reinterpret_cast<const i32x4*>(&s_a[buf * STAGE + a_off + ks * 32]);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const i32x4 c = *reinterpret_cast<const i32x4*>(&s_k[t0 + 8 * g + 4 * hi]);
            kk[4 * g] = c.x; kk[4 * g + 1] = c.y; kk[4 * g + 2] = c.z; kk[4 * g + 3] = c.w;
        }
    };
    constexpr int IMIN = (int)0x80000000;
    int bz[2] = {IMIN, IMIN}, bthr[2] = {IMIN | 15, IMIN | 15}, bt[2] = {0, 0};
    int b2[2] = {IMIN, IMIN}, s2[2] = {IMIN, IMIN};       // SECOND: multiset top-2 of z
    i32x16 acc[2];
    // epilogue of one 32 x 32 block: z = 32 S + k, the tile's maximum with its slot, the running best
    auto epilogue = [&](int c, const i32x16& kk, int tile) {
        int z[16];
#pragma unroll
        for (int e = 0; e < 16; e++) z[e] = (acc[c][e] << 5) + kk[e];
        if (SECOND) {
#pragma unroll
            for (int e = 0; e < 16; e++) { s2[c] = med3_i32(b2[c], z[e], s2[c]); b2[c] = max(b2[c], z[e]); }
        }
        int m4[4];
#pragma unroll
        for (int g = 0; g < 4; g++) m4[g] = max(max(max(z[4 * g], z[4 * g + 1]), z[4 * g + 2]), z[4 * g + 3]);
        const int tm = max(max(max(m4[0], m4[1]), m4[2]), m4[3]);
        const bool win = tm > bthr[c];                    // strictly nearer than the best so far (the position bits masked out)
        bz[c] = win ? tm : bz[c];
        bt[c] = win ? tile : bt[c];
        bthr[c] = bz[c] | 15;
    };
    auto chain = [&](int c) {
        acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0], bq[c][0], i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ks++) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[ks], bq[c][ks], acc[c], 0, 0, 0);
    };
    // Within a wave the matrix pipe and the VALU work side by side: the epilogue of query block 1 of the PREVIOUS tile is issued among the
    // MFMAs of block 0, the epilogue of block 0 among the MFMAs of block 1 (each needs only the other block's accumulator to be busy).
    int cur = 0;
    auto iteration = [&](int t0, int tile, i32x16& kcur, i32x16& kother) {   // kother: constants of tile - 1 on entry, of tile + 1 on exit
        const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
        const u32x2 pf_stage = pf;                        // tile t + 2, fetched one iteration ago
        fetch(t0 + 96 < npad ? t0 + 96 : 0);              // past the end the loop fetches / stages / reads tiles nobody uses
#pragma unroll
        for (int ks = 2; ks < 4; ks++) fa[ks] = *reinterpret_cast<const i32x4*>(&s_a[cur * STAGE + a_off + ks * 32]);
        chain(0);
        epilogue(1, kother, tile - 1);
        chain(1);
        read_lo(nxt, t0 + 32, kother);                    // issued before the barrier: the next iteration starts with its MFMAs
        epilogue(0, kcur, tile);
        stage(nx2, pf_stage);
        __syncthreads();
        cur = nxt;
    };
    read_lo(0, 0, kA);
#pragma unroll
    for (int e = 0; e < 16; e++) { acc[1][e] = 0; kB[e] = NOROW; }     // the "previous tile" of the first iteration: nothing
    int tile = 0;
    for (int t0 = 0; t0 < npad; t0 += 64, tile += 2) {    // npad is a multiple of 256
        iteration(t0, tile, kA, kB);
        iteration(t0 + 32, tile + 1, kB, kA);
    }
    epilogue(1, kB, tile - 1);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        int best = 0x7fffffff, second = 0x7fffffff, bi = -1;
        const int e = 15 - (bz[c] & 15), row = bt[c] * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
        const bool q_ok = q[c] < pd.n_i;
        const int nq = q_ok ? GLOBAL_PTR(int, pd.n8_i)[q[c]] : 0;
        if (q_ok && bz[c] != IMIN && bt[c] >= 0 && row < pd.n_j) { bi = row; best = nq - (bz[c] >> 4); }      // bt < 0: the lane's 16 rows never held a train row (n_j <= 4 / n_j == 0)
        if (SECOND && q_ok && s2[c] > NOROW / 2) second = nq - (s2[c] >> 4);
        // merge the two half-waves (same query, disjoint train rows); ties -> lowest train index
        const int ob = __shfl_xor(best, 32), os = __shfl_xor(second, 32), oi = __shfl_xor(bi, 32);
        const bool other_wins = (ob < best) || (ob == best && oi >= 0 && (bi < 0 || oi < bi));
        const int nb = other_wins ? ob : best, ni = other_wins ? oi : bi;
        if (hi == 0 && q_ok) {
            const size_t o = (size_t)pair * KSTRIDE + q[c];
            nn_idx[o] = ni; nn_d2[o] = nb;
            if (SECOND) {
                const int loser_best = other_wins ? best : ob;
                const int min_sec = os < second ? os : second;
                nn_2nd[o] = loser_best < min_sec ? loser_best : min_sec;
            }
        }
    }
}

// ---- K7: sort by (d2, queryIdx) + grid walk -----------------------------------------------------------------
struct SelectParams { int max_selected; double fraction; int gx, gy; float ratio2; };

// The grid walk of SelectMatchPairs by ONE wave, 64 sorted matches per step (MosaicWithoutPos.cpp:4977-5028): key_at(i) = the i-th key of the
// sorted list ((d2 << 32) | queryIdx), idx / d2 / d2nd the pair's 1-NN arrays indexed by query.  Shared by select_kernel (keys in LDS) and
// select_big_kernel (keys in HBM); inlined into both, so the LDS form keeps its ds_ reads.
template <class KeyAt>
__device__ __forceinline__ void grid_walk(const PairDesc& pd, int M, KeyAt key_at, const int* idx, const int* d2v, const int* d2nd, const SelectParams& sp, int* s_label,
                                          mi355_sfpoint* o1, mi355_sfpoint* o2, int* nsel_out, int lane) {
    const int nGrids = sp.gx * sp.gy;
    const double lim = sp.fraction * (double)M;                        // Min(400, 0.3*M) evaluated in double (:5146-5147)
    const int nMatch = (int)((double)sp.max_selected < lim ? (double)sp.max_selected : lim);
    const int perGrid = (int)((float)nMatch / (float)nGrids);                    // :4990
    const int stepX = pd.width / sp.gx, stepY = pd.height / sp.gy;               // :4994-4995
    if (lane < 64) s_label[lane] = 0;
    int count = 0;
    for (int base = 0; base < M; base += 64) {
        const int i = base + lane;
        bool valid = i < M;
        int q = 0, t = 0, cell = 0; float x = 0.0f, y = 0.0f;
        if (valid) {
            q = (int)(unsigned)(key_at(i) & 0xffffffffull);
            t = idx[q];
            const float2 pxy = pd.xy_i[q];
            x = pxy.x; y = pxy.y;
            const int nX = (int)(x / (float)stepX), nY = (int)(y / (float)stepY);      // :5008-5009
            cell = sp.gx * nY + nX;                       // aliases into the next row when nX == gridX, like the reference
            if (cell < 0) cell = 0;
            if (cell >= nGrids) cell = nGrids - 1;        // the reference would index label[] out of bounds here
            if (sp.ratio2 > 0.0f) {                       // optional Lowe ratio test (north_star), squared distances
                const float d1 = (float)d2v[q], dd2 = (float)d2nd[q];
                if (!(d1 < sp.ratio2 * dd2)) valid = false;
            }
        }
        bool keep = false;
        for (int cc = 0; cc < nGrids; cc++) {
            const unsigned long long m = __ballot(valid && cell == cc);
            if (m == 0) continue;
            const int lab = s_label[cc];
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (valid && cell == cc && lab + rank < perGrid) keep = true;
            int add = __popcll(m);
            const int room = perGrid - lab;
            if (add > room) add = room > 0 ? room : 0;
            if (lane == 0) s_label[cc] = lab + add;
        }
        const unsigned long long km = __ballot(keep);
        const int pos = count + __popcll(km & ((1ull << lane) - 1ull));
        if (keep && pos < MI355_MAX_SELECTED) {
            const float2 p2 = pd.xy_j[t];
            o1[pos].x = x; o1[pos].y = y; o1[pos].id = q;
            o2[pos].x = p2.x; o2[pos].y = p2.y; o2[pos].id = t;
        }
        count += __popcll(km);
    }
    if (lane == 0) *nsel_out = count < MI355_MAX_SELECTED ? count : MI355_MAX_SELECTED;
}

__global__ __launch_bounds__(256) void select_kernel(const PairDesc* pairs, const int* nn_idx, const int* nn_d2, const int* nn_2nd,
                                                     SelectParams sp, mi355_sfpoint* sel1, mi355_sfpoint* sel2, int* nsel,
                                                     unsigned long long* sorted_keys /* optional [pair][KSTRIDE] */) {
    __shared__ unsigned long long key[KSTRIDE];
    __shared__ int s_label[64];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[pair];
    const int M = (pd.n_j > 0) ? pd.n_i : 0;                       // one match per query descriptor
    const size_t o = (size_t)pair * KSTRIDE;
    for (int i = tid; i < KSTRIDE; i += 256)
        key[i] = (i < M) ? (((unsigned long long)(unsigned)nn_d2[o + i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    // bitonic sort of 2048 64-bit keys in LDS
    for (int k = 2; k <= KSTRIDE; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < KSTRIDE; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], b = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (sorted_keys) for (int i = tid; i < KSTRIDE; i += 256) sorted_keys[o + i] = key[i];
    if (tid >= 64) return;
    grid_walk(pd, M, [&](int i) { return key[i]; }, nn_idx + o, nn_d2 + o, nn_2nd + o, sp, s_label, sel1 + (size_t)pair * MI355_MAX_SELECTED, sel2 + (size_t)pair * MI355_MAX_SELECTED, nsel + pair, tid);
}

// ---- more than 2048 keypoints per image ---------------------------------------------------------------------------------------------
// The reference's committed run kept EVERY keypoint (nfeatures = 0: ids up to 3130 in matchPairs.match) and its j-loop takes any M
// (MosaicWithoutPos.cpp:5108-5153: nMatch = Min(400, 0.3 M)).  The live path (nfeatures 2000) never gets here; a pair with a larger image
// goes through the same kernels in pieces: bf_match_kernel on (query chunk, train chunk) sub-pairs of <= 2048 x 2048 -- every sub-pair is a
// PairDesc whose pointers start at its chunks --, merge_chunks_kernel takes each query's nearest over the train chunks (ties -> the lowest
// train index: the earliest chunk, and inside a chunk the kernel already keeps the lowest), select_big_kernel sorts the M keys
// (d2, queryIdx) in HBM -- one workgroup, bitonic over the next power of two -- and walks the grid with the very code of select_kernel.
struct BigPairDev { int n_i, n_j, tc_n, sub0; long long off, koff; int mpad, _pad; };

__global__ __launch_bounds__(256) void merge_chunks_kernel(const BigPairDev* bp, const int* nn_idx, const int* nn_d2, const int* nn_2nd, int second,
                                                           int* m_idx, int* m_d2, int* m_2nd) {
    const BigPairDev b = bp[blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= b.n_i) return;
    const int qc = q / KSTRIDE, ql = q - qc * KSTRIDE;
    int best_d = 0x7fffffff, best_i = -1, sec = 0x7fffffff;
    for (int tc = 0; tc < b.tc_n; tc++) {
        const size_t o = (size_t)(b.sub0 + qc * b.tc_n + tc) * KSTRIDE + ql;
        const int i = nn_idx[o], d = nn_d2[o];
        if (i >= 0) {
            if (d < best_d) { sec = sec < best_d ? sec : best_d; best_d = d; best_i = i + tc * KSTRIDE; }
            else sec = sec < d ? sec : d;
        }
        if (second) { const int s2 = nn_2nd[o]; sec = sec < s2 ? sec : s2; }
    }
    m_idx[b.off + q] = best_i; m_d2[b.off + q] = best_d; m_2nd[b.off + q] = sec;
}

__global__ __launch_bounds__(1024) void select_big_kernel(const PairDesc* pairs, const BigPairDev* bp, const int* m_idx, const int* m_d2, const int* m_2nd,
                                                          SelectParams sp, unsigned long long* keys, mi355_sfpoint* sel1, mi355_sfpoint* sel2, int* nsel) {
    __shared__ int s_label[64];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const PairDesc pd = pairs[pair];
    const BigPairDev b = bp[pair];
    const int M = (pd.n_j > 0) ? pd.n_i : 0;
    unsigned long long* key = keys + b.koff;
    const int* d2 = m_d2 + b.off;
    for (int i = tid; i < b.mpad; i += 1024)
        key[i] = (i < M) ? (((unsigned long long)(unsigned)d2[i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    // bitonic sort in HBM by the one workgroup (all its waves share the CU's cache: a barrier orders the passes)
    for (int k = 2; k <= b.mpad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < b.mpad; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], c = key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { key[i] = c; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (tid >= 64) return;
    grid_walk(pd, M, [&](int i) { return key[i]; }, m_idx + b.off, d2, m_2nd + b.off, sp, s_label, sel1 + (size_t)pair * MI355_MAX_SELECTED, sel2 + (size_t)pair * MI355_MAX_SELECTED, nsel + pair, tid);
}

__global__ void finalize_kernel(const PairDesc* pairs, const int* nsel, int n_pairs, int min_inliers, mi355_pair_result* out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    out[p].i = pairs[p].img_i; out[p].j = pairs[p].img_j;
    out[p].n_selected = nsel[p];
    out[p].accepted = out[p].n_in > min_inliers ? 1 : 0;          // MosaicWithoutPos.cpp:5201
}

// ---- features ------------------------------------------------------------------------------------------------
// 8 lanes per descriptor row (16 bytes each), 32 rows per workgroup: a workgroup per row was 2048 workgroups of 128 threads per image,
// dispatch-bound (7 us per image; 3.5 ms for the 500 images a rank installs after the feature exchange)
__global__ __launch_bounds__(256) void finish_features_kernel(const mi355_keypoint* kp, const uint8_t* d8, int n, const int* d_n, int npad,
                                                              float2* xy, int8_t* s8, int* n8) {
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (row >= npad) return;
    if (d_n) n = *d_n;                                       // count still on the device (asynchronous SIFT)
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);     // padding rows: zeros after the shift
    if (row < n) v = *reinterpret_cast<const uint4*>(d8 + (size_t)row * 128 + part * 16);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int b = 0; b < 4; b++) { const int t = (int)((w[q] >> (8 * b)) & 0xffu) - 128; s += t * t; }
    // the matcher's operand: the descriptor moved to int8 (u - 128 = u ^ 0x80 as a byte)
    *reinterpret_cast<uint4*>(s8 + (size_t)row * 128 + part * 16) = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0) {
        n8[row] = s;
        if (row < n) xy[row] = make_float2(kp[row].x, kp[row].y);
    }
}

// the same for all frames of a SIFT batch in one launch (blockIdx.y = frame): 32 launches of 5 us at the tail of every batch were 160 us
// during which the batch's stream held a slot of the pipeline for nothing
struct FinishBatch {
    const mi355_keypoint* kp[MI355_SIFT_BATCH_MAX]; const uint8_t* d8[MI355_SIFT_BATCH_MAX];
    float2* xy[MI355_SIFT_BATCH_MAX]; int8_t* s8[MI355_SIFT_BATCH_MAX]; int* n8[MI355_SIFT_BATCH_MAX];
};
__global__ __launch_bounds__(256) void finish_features_batch_kernel(FinishBatch fb, const int* d_n, int n_stride, int npad) {
    const int k = blockIdx.y;
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (row >= npad) return;
    const int n = d_n[(size_t)k * n_stride];
    const uint8_t* d8 = fb.d8[k];
    uint4 v = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    if (row < n) v = *reinterpret_cast<const uint4*>(d8 + (size_t)row * 128 + part * 16);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    int s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int b = 0; b < 4; b++) { const int t = (int)((w[q] >> (8 * b)) & 0xffu) - 128; s += t * t; }
    *reinterpret_cast<uint4*>(fb.s8[k] + (size_t)row * 128 + part * 16) = make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0) {
        fb.n8[k][row] = s;
        if (row < n) fb.xy[k][row] = make_float2(fb.kp[k][row].x, fb.kp[k][row].y);
    }
}

__global__ void desc_f32_to_u8_kernel(const float* f, uint8_t* d8, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float v = f[i];
    v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    d8[i] = (uint8_t)(int)(v + 0.5f);
}

__global__ void desc_u8_to_f32_kernel(const uint8_t* d8, float* f, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) f[i] = (float)d8[i];
}

int build_pair_table(mi355_ctx* ctx, const int32_t* pairs, int n_pairs, std::vector<PairDesc>& pd) {
    {                                                       // wait for the frames of THESE pairs only: later batches keep running
        std::vector<int> ids(pairs, pairs + 2 * (size_t)n_pairs);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        int rc = mi_resolve_features_of(ctx, ids.data(), (int)ids.size());
        if (rc != MI355_OK) return rc;
    }
    pd.resize(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        const int i = pairs[2 * p], j = pairs[2 * p + 1];
        auto fi = ctx->feats.find(i), fj = ctx->feats.find(j);
        if (fi == ctx->feats.end() || fj == ctx->feats.end()) { ctx->set_error("match_pairs: no resident features for image " + std::to_string(fi == ctx->feats.end() ? i : j)); return MI355_ERR_ARG; }
        const Features &a = fi->second, &b = fj->second;
        if (a.n > BIG_KP_MAX || b.n > BIG_KP_MAX) { ctx->set_error("match_pairs: more than 32768 keypoints per image"); return MI355_ERR_ARG; }
        if ((a.n > KSTRIDE && a.npad < a.n) || (b.n > KSTRIDE && b.npad < b.n)) { ctx->set_error("match_pairs: the matcher's operands of a large image are incomplete"); return MI355_ERR_ARG; }
        PairDesc& d = pd[p];
        d.s8_i = a.s8.as<int8_t>(); d.n8_i = a.n8.as<int>(); d.xy_i = a.xy.as<float2>(); d.n_i = a.n; d.npad_i = a.npad;
        d.s8_j = b.s8.as<int8_t>(); d.n8_j = b.n8.as<int>(); d.xy_j = b.xy.as<float2>(); d.n_j = b.n; d.npad_j = b.npad;
        d.img_i = i; d.img_j = j; d.width = a.w; d.height = a.h;        // the cell comes from the image-i point (:5002-5009)
    }
    return MI355_OK;
}

}  // namespace

int mi_finish_features(mi355_ctx* ctx, Features& f, const int* d_n, hipStream_t st) {
    if (!st) st = ctx->stream;
    const int nmax = d_n ? KSTRIDE : f.n;
    f.npad = ((nmax + ROWPAD - 1) / ROWPAD) * ROWPAD;
    if (f.npad == 0) f.npad = ROWPAD;
    MI_HIP(f.xy.reserve(sizeof(float2) * (size_t)(nmax > 0 ? nmax : 1)));
    MI_HIP(f.s8.reserve((size_t)128 * (size_t)f.npad));
    MI_HIP(f.n8.reserve(sizeof(int) * (size_t)f.npad));
    ProfScope ps(ctx, "features", (double)f.npad * 128 * 2, st);
    hipLaunchKernelGGL(finish_features_kernel, dim3((f.npad + 31) / 32), dim3(256), 0, st,
                       f.kp.as<mi355_keypoint>(), f.d8.as<uint8_t>(), f.n, d_n, f.npad, f.xy.as<float2>(), f.s8.as<int8_t>(), f.n8.as<int>());
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

// the matcher's operands of all frames of a SIFT batch; the keypoint counts are still on the device (d_n[k * n_stride])
int mi_finish_features_batch(mi355_ctx* ctx, Features* const* fs, int nf, const int* d_n, int n_stride, hipStream_t st, int max_rows) {
    if (nf <= 0) return MI355_OK;
    if (nf > MI355_SIFT_BATCH_MAX) return MI355_ERR_ARG;
    if (max_rows < KSTRIDE) max_rows = KSTRIDE;          // keep-all frames (nfeatures <= 0) carry up to MI355_SIFT_KEEPALL_MAX rows: the large-pair path reads them all
    FinishBatch fb;
    memset(&fb, 0, sizeof(fb));
    const int npad = ((max_rows + ROWPAD - 1) / ROWPAD) * ROWPAD;
    for (int k = 0; k < nf; k++) {
        Features& f = *fs[k];
        f.npad = npad;
        MI_HIP(f.xy.reserve(sizeof(float2) * (size_t)max_rows));
        MI_HIP(f.s8.reserve((size_t)128 * (size_t)npad));
        MI_HIP(f.n8.reserve(sizeof(int) * (size_t)npad));
        fb.kp[k] = f.kp.as<mi355_keypoint>(); fb.d8[k] = f.d8.as<uint8_t>();
        fb.xy[k] = f.xy.as<float2>(); fb.s8[k] = f.s8.as<int8_t>(); fb.n8[k] = f.n8.as<int>();
    }
    ProfScope ps(ctx, "features", (double)npad * 128 * 2 * nf, st);
    hipLaunchKernelGGL(finish_features_batch_kernel, dim3((npad + 31) / 32, nf), dim3(256), 0, st, fb, d_n, n_stride, npad);
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

int mi_set_features(mi355_ctx* ctx, int img_id, const mi355_keypoint* kp, const float* desc, int n, int w, int h) {
    if (n < 0 || n > BIG_KP_MAX || (n > 0 && (!kp || !desc)) || w <= 0 || h <= 0) { ctx->set_error("set_features: bad arguments (n must be <= 32768)"); return MI355_ERR_ARG; }
    (void)mi_resolve_features(ctx);
    Features& f = ctx->feats[img_id];
    f.n = n; f.w = w; f.h = h; f.pending = false;
    const size_t cnt = (size_t)n * 128;
    MI_HIP(f.kp.reserve(sizeof(mi355_keypoint) * (size_t)(n > 0 ? n : 1)));
    MI_HIP(f.d8.reserve(cnt > 0 ? cnt : 1));
    if (n > 0) {
        DevBuf& tmp = ctx->buf("desc_f32");
        MI_HIP(tmp.reserve(cnt * sizeof(float)));
        MI_HIP(hipMemcpyAsync(f.kp.p, kp, sizeof(mi355_keypoint) * n, hipMemcpyHostToDevice, ctx->stream));
        MI_HIP(hipMemcpyAsync(tmp.p, desc, cnt * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(desc_f32_to_u8_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, tmp.as<float>(), f.d8.as<uint8_t>(), cnt);
    }
    int rc = mi_finish_features(ctx, f);
    if (rc != MI355_OK) return rc;
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

extern "C" int mi355_get_features(mi355_ctx* ctx, int img_id, mi355_keypoint* kp, float* desc128, int max_kp, int* n_kp) {
    LOCKED_PROLOGUE
    { int rc = mi_resolve_features(ctx); if (rc != MI355_OK) return rc; }
    auto it = ctx->feats.find(img_id);
    if (it == ctx->feats.end()) { ctx->set_error("get_features: unknown image id"); return MI355_ERR_ARG; }
    Features& f = it->second;
    if (n_kp) *n_kp = f.n;
    const int n = f.n < max_kp ? f.n : max_kp;
    if (n <= 0) return MI355_OK;
    if (kp) MI_HIP(hipMemcpyAsync(kp, f.kp.p, sizeof(mi355_keypoint) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (desc128) {
        const size_t cnt = (size_t)n * 128;
        DevBuf& tmp = ctx->buf("desc_f32");
        MI_HIP(tmp.reserve(cnt * sizeof(float)));
        hipLaunchKernelGGL(desc_u8_to_f32_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, f.d8.as<uint8_t>(), tmp.as<float>(), cnt);
        MI_HIP(hipMemcpyAsync(desc128, tmp.p, cnt * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    MI_HIP(hipStreamSynchronize(ctx->stream));
    return MI355_OK;
}

static int run_match_select(mi355_ctx* ctx, const std::vector<PairDesc>& pd, int n_pairs, bool want_sorted_keys) {
    DevBuf& dpd = ctx->buf("pair_desc");
    DevBuf& didx = ctx->buf("nn_idx");
    DevBuf& dd2 = ctx->buf("nn_d2");
    DevBuf& d2nd = ctx->buf("nn_2nd");
    DevBuf& ds1 = ctx->buf("sel1");
    DevBuf& ds2 = ctx->buf("sel2");
    DevBuf& dns = ctx->buf("nsel");
    DevBuf& dkeys = ctx->buf("sorted_keys");
    const size_t nn = (size_t)n_pairs * KSTRIDE;
    MI_HIP(dpd.reserve(sizeof(PairDesc) * n_pairs));
    MI_HIP(didx.reserve(nn * 4)); MI_HIP(dd2.reserve(nn * 4)); MI_HIP(d2nd.reserve(nn * 4));
    MI_HIP(ds1.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)n_pairs));
    MI_HIP(ds2.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)n_pairs));
    MI_HIP(dns.reserve(sizeof(int) * n_pairs));
    if (want_sorted_keys) MI_HIP(dkeys.reserve(nn * 8));
    MI_HIP(hipMemcpyAsync(dpd.p, pd.data(), sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    int max_ni = 1;
    double flops_bytes = 0.0;
    for (int p = 0; p < n_pairs; p++) { if (pd[p].n_i > max_ni) max_ni = pd[p].n_i; flops_bytes += (double)(pd[p].npad_i + pd[p].npad_j) * 128.0 + 8.0 * pd[p].n_i; }
    {
        ProfScope ps(ctx, "match", flops_bytes);
        const int nqt = (max_ni + QTILE - 1) / QTILE;
        const bool second = want_sorted_keys || ctx->p.ratio > 0.0f;      // mi355_bf_match reports it, the ratio test reads it
        if (second) hipLaunchKernelGGL(bf_match_kernel<true>, dim3((unsigned)nqt * (unsigned)n_pairs), dim3(BF_NT), 0, ctx->stream,
                                       dpd.as<PairDesc>(), nqt, didx.as<int>(), dd2.as<int>(), d2nd.as<int>());
        else hipLaunchKernelGGL(bf_match_kernel<false>, dim3((unsigned)nqt * (unsigned)n_pairs), dim3(BF_NT), 0, ctx->stream,
                                dpd.as<PairDesc>(), nqt, didx.as<int>(), dd2.as<int>(), d2nd.as<int>());
    }
    SelectParams sp;
    sp.max_selected = ctx->p.max_selected; sp.fraction = ctx->p.select_fraction; sp.gx = ctx->p.grid_x; sp.gy = ctx->p.grid_y;
    sp.ratio2 = ctx->p.ratio > 0.0f ? ctx->p.ratio * ctx->p.ratio : 0.0f;
    {
        ProfScope ps(ctx, "select", (double)n_pairs * (KSTRIDE * 12.0 + 9600.0));
        hipLaunchKernelGGL(select_kernel, dim3(n_pairs), dim3(256), 0, ctx->stream, dpd.as<PairDesc>(), didx.as<int>(), dd2.as<int>(), d2nd.as<int>(),
                           sp, ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>(),
                           want_sorted_keys ? dkeys.as<unsigned long long>() : (unsigned long long*)nullptr);
    }
    MI_HIP(hipGetLastError());
    return MI355_OK;
}

// The large-pair form of run_match_select (an image of the pair has more than 2048 keypoints): same outputs -- sel1 / sel2 / nsel by pair of the
// run, "pair_desc" for finalize_kernel -- through the sub-pair decomposition described at merge_chunks_kernel.  The merged 1-NN arrays stay in
// "nnb_idx" / "nnb_d2" / "nnb_2nd" at off[p] and the sorted keys in "sorted_keys_big" at koff[p] (mi_bf_match reads them back).
static int run_match_select_big(mi355_ctx* ctx, const std::vector<PairDesc>& pd, int n_pairs, bool want_second, std::vector<BigPairDev>* layout_out = nullptr) {
    std::vector<BigPairDev> bp(n_pairs);
    std::vector<PairDesc> sub;
    long long off = 0, koff = 0;
    int max_ni = 1;
    for (int p = 0; p < n_pairs; p++) {
        const PairDesc& d = pd[p];
        const int qc_n = d.n_i > 0 ? (d.n_i + KSTRIDE - 1) / KSTRIDE : 1, tc_n = d.n_j > 0 ? (d.n_j + KSTRIDE - 1) / KSTRIDE : 1;
        BigPairDev& b = bp[p];
        b.n_i = d.n_i; b.n_j = d.n_j; b.tc_n = tc_n; b.sub0 = (int)sub.size(); b.off = off; b.koff = koff; b._pad = 0;
        int mp = 64; while (mp < d.n_i) mp <<= 1;
        b.mpad = mp;
        off += d.n_i > 0 ? d.n_i : 1; koff += mp;
        if (d.n_i > max_ni) max_ni = d.n_i;
        for (int qc = 0; qc < qc_n; qc++)
            for (int tc = 0; tc < tc_n; tc++) {
                PairDesc s = d;
                const int q0 = qc * KSTRIDE, t0 = tc * KSTRIDE;
                s.s8_i = d.s8_i + (size_t)q0 * 128; s.n8_i = d.n8_i + q0; s.xy_i = d.xy_i + q0;
                s.n_i = d.n_i - q0 < KSTRIDE ? (d.n_i - q0 > 0 ? d.n_i - q0 : 0) : KSTRIDE;
                s.npad_i = ((s.n_i + ROWPAD - 1) / ROWPAD) * ROWPAD; if (s.npad_i == 0) s.npad_i = ROWPAD;
                s.s8_j = d.s8_j + (size_t)t0 * 128; s.n8_j = d.n8_j + t0; s.xy_j = d.xy_j + t0;
                s.n_j = d.n_j - t0 < KSTRIDE ? (d.n_j - t0 > 0 ? d.n_j - t0 : 0) : KSTRIDE;
                s.npad_j = ((s.n_j + ROWPAD - 1) / ROWPAD) * ROWPAD; if (s.npad_j == 0) s.npad_j = ROWPAD;
                sub.push_back(s);
            }
    }
    const int nsub = (int)sub.size();
    DevBuf& dpd = ctx->buf("pair_desc"); DevBuf& dsub = ctx->buf("pair_desc_sub"); DevBuf& dbp = ctx->buf("big_pairs");
    DevBuf& didx = ctx->buf("nn_idx"); DevBuf& dd2 = ctx->buf("nn_d2"); DevBuf& d2nd = ctx->buf("nn_2nd");
    DevBuf& midx = ctx->buf("nnb_idx"); DevBuf& md2 = ctx->buf("nnb_d2"); DevBuf& m2nd = ctx->buf("nnb_2nd"); DevBuf& dkeys = ctx->buf("sorted_keys_big");
    DevBuf& ds1 = ctx->buf("sel1"); DevBuf& ds2 = ctx->buf("sel2"); DevBuf& dns = ctx->buf("nsel");
    const size_t nn = (size_t)nsub * KSTRIDE;
    MI_HIP(dpd.reserve(sizeof(PairDesc) * n_pairs)); MI_HIP(dsub.reserve(sizeof(PairDesc) * nsub)); MI_HIP(dbp.reserve(sizeof(BigPairDev) * n_pairs));
    MI_HIP(didx.reserve(nn * 4)); MI_HIP(dd2.reserve(nn * 4)); MI_HIP(d2nd.reserve(nn * 4));
    MI_HIP(midx.reserve((size_t)off * 4)); MI_HIP(md2.reserve((size_t)off * 4)); MI_HIP(m2nd.reserve((size_t)off * 4)); MI_HIP(dkeys.reserve((size_t)koff * 8));
    MI_HIP(ds1.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)n_pairs));
    MI_HIP(ds2.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED * (size_t)n_pairs));
    MI_HIP(dns.reserve(sizeof(int) * n_pairs));
    MI_HIP(hipMemcpyAsync(dpd.p, pd.data(), sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dsub.p, sub.data(), sizeof(PairDesc) * nsub, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dbp.p, bp.data(), sizeof(BigPairDev) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));           // the host vectors are locals
    const bool second = want_second || ctx->p.ratio > 0.0f;
    {
        double bytes = 0.0;
        for (const PairDesc& q : sub) bytes += (double)(q.npad_i + q.npad_j) * 128.0 + 8.0 * q.n_i;
        ProfScope ps(ctx, "match", bytes);
        const int nqt = KSTRIDE / QTILE;
        if (second) hipLaunchKernelGGL(bf_match_kernel<true>, dim3((unsigned)nqt * (unsigned)nsub), dim3(BF_NT), 0, ctx->stream, dsub.as<PairDesc>(), nqt, didx.as<int>(), dd2.as<int>(), d2nd.as<int>());
        else hipLaunchKernelGGL(bf_match_kernel<false>, dim3((unsigned)nqt * (unsigned)nsub), dim3(BF_NT), 0, ctx->stream, dsub.as<PairDesc>(), nqt, didx.as<int>(), dd2.as<int>(), d2nd.as<int>());
        hipLaunchKernelGGL(merge_chunks_kernel, dim3((max_ni + 255) / 256, n_pairs), dim3(256), 0, ctx->stream, dbp.as<BigPairDev>(), didx.as<int>(), dd2.as<int>(), d2nd.as<int>(), second ? 1 : 0,
                           midx.as<int>(), md2.as<int>(), m2nd.as<int>());
    }
    SelectParams sp;
    sp.max_selected = ctx->p.max_selected; sp.fraction = ctx->p.select_fraction; sp.gx = ctx->p.grid_x; sp.gy = ctx->p.grid_y;
    sp.ratio2 = ctx->p.ratio > 0.0f ? ctx->p.ratio * ctx->p.ratio : 0.0f;
    {
        ProfScope ps(ctx, "select", (double)koff * 8.0);
        hipLaunchKernelGGL(select_big_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, dpd.as<PairDesc>(), dbp.as<BigPairDev>(), midx.as<int>(), md2.as<int>(), m2nd.as<int>(), sp,
                           dkeys.as<unsigned long long>(), ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>());
    }
    MI_HIP(hipGetLastError());
    if (layout_out) *layout_out = bp;
    return MI355_OK;
}

static bool pair_is_big(const PairDesc& d) { return d.n_i > KSTRIDE || d.n_j > KSTRIDE; }

int mi_match_pairs_dev(mi355_ctx* ctx, const int32_t* pairs, int n_pairs, float dist, uint32_t seed, mi355_pair_result* d_out) {
    if (n_pairs <= 0) return MI355_OK;
    if (!pairs || !d_out) return MI355_ERR_ARG;
    if (ctx->p.grid_x * ctx->p.grid_y > 64 || ctx->p.grid_x < 1 || ctx->p.grid_y < 1 || ctx->p.max_selected > MI355_MAX_SELECTED) { ctx->set_error("match_pairs: grid > 64 cells or max_selected > 400"); return MI355_ERR_ARG; }
    const int BATCH = 32768;                                  // bounds the nn workspaces (32768 x 2048 x 12 B = 768 MiB of the 288 GB); every batch boundary drains the stream
    std::vector<PairDesc> pd;
    { const int rc = mi_ransac_tables(ctx, seed, nullptr); if (rc != MI355_OK) return rc; }      // a new seed's draw tables are built beside the matcher
    for (int b0 = 0; b0 < n_pairs; b0 += BATCH) {
        const int nb = (n_pairs - b0) < BATCH ? (n_pairs - b0) : BATCH;
        int rc = build_pair_table(ctx, pairs + 2 * b0, nb, pd);
        if (rc != MI355_OK) return rc;
        // maximal runs of pairs of one kind: the live path (every image <= 2048 keypoints) is ONE run per batch, exactly as before; a pair with
        // a larger image (keep-all frames) goes through the large-pair form, at most 256 of them at a time (each may hold 256 sub-pairs)
        for (int r0 = 0; r0 < nb;) {
            const bool big = pair_is_big(pd[r0]);
            int r1 = r0 + 1;
            while (r1 < nb && pair_is_big(pd[r1]) == big && (!big || r1 - r0 < 256)) r1++;
            const int nr = r1 - r0;
            if (r0 == 0 && nr == nb && !big) rc = run_match_select(ctx, pd, nb, false);
            else {
                const std::vector<PairDesc> run(pd.begin() + r0, pd.begin() + r1);
                rc = big ? run_match_select_big(ctx, run, nr, false) : run_match_select(ctx, run, nr, false);
            }
            if (rc != MI355_OK) return rc;
            rc = mi_ransac_batch(ctx, ctx->buf("sel1").as<mi355_sfpoint>(), ctx->buf("sel2").as<mi355_sfpoint>(), ctx->buf("nsel").as<int>(), nullptr,
                                 nr, MI355_MAX_SELECTED, dist, ctx->p.sample_times, seed, d_out + b0 + r0, ctx->p.min_inliers);
            if (rc != MI355_OK) return rc;
            hipLaunchKernelGGL(finalize_kernel, dim3((nr + 255) / 256), dim3(256), 0, ctx->stream, ctx->buf("pair_desc").as<PairDesc>(), ctx->buf("nsel").as<int>(),
                               nr, ctx->p.min_inliers, d_out + b0 + r0);
            MI_HIP(hipGetLastError());
            if (r1 < nb || b0 + BATCH < n_pairs) MI_HIP(hipStreamSynchronize(ctx->stream));     // workspaces are reused by the next run / batch
            r0 = r1;
        }
    }
    return MI355_OK;
}

int mi_bf_match(mi355_ctx* ctx, int img_i, int img_j, int sorted, mi355_dmatch* matches, int32_t* d2, int32_t* second, int maxm, int* nm) {
    const int32_t pr[2] = {img_i, img_j};
    std::vector<PairDesc> pd;
    int rc = build_pair_table(ctx, pr, 1, pd);
    if (rc != MI355_OK) return rc;
    const bool big = pair_is_big(pd[0]);
    rc = big ? run_match_select_big(ctx, pd, 1, true) : run_match_select(ctx, pd, 1, true);
    if (rc != MI355_OK) return rc;
    const int M = pd[0].n_j > 0 ? pd[0].n_i : 0;
    const size_t K = big ? (size_t)(M > 0 ? M : 1) : (size_t)KSTRIDE;        // the large-pair form keeps one pair at offset 0 of its merged arrays
    std::vector<int> idx(K), dd(K), d2nd(K);
    std::vector<unsigned long long> keys(K);
    MI_HIP(hipMemcpyAsync(idx.data(), ctx->buf(big ? "nnb_idx" : "nn_idx").p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(dd.data(), ctx->buf(big ? "nnb_d2" : "nn_d2").p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(d2nd.data(), ctx->buf(big ? "nnb_2nd" : "nn_2nd").p, K * 4, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(keys.data(), ctx->buf(big ? "sorted_keys_big" : "sorted_keys").p, K * 8, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    const int n = M < maxm ? M : maxm;
    for (int k = 0; k < n; k++) {
        const int q = sorted ? (int)(keys[k] & 0xffffffffull) : k;
        if (matches) { matches[k].queryIdx = q; matches[k].trainIdx = idx[q]; matches[k].imgIdx = 0; matches[k].distance = sqrtf((float)dd[q]); }
        if (d2) d2[k] = dd[q];
        if (second) second[k] = d2nd[q];
    }
    if (nm) *nm = M;
    return MI355_OK;
}

// stand-alone SelectMatchPairs: host arrays in, the same select kernel on one synthetic "pair"
int mi_select_grid(mi355_ctx* ctx, const mi355_dmatch* sorted, int n, const float* kp1, int nk1, const float* kp2, int nk2,
                   int nMatch, int width, int height, int gx, int gy, mi355_sfpoint* v1, mi355_sfpoint* v2, int* n_out) {
    if (n < 0 || n > BIG_KP_MAX || !kp1 || !kp2 || !v1 || !v2 || !n_out || gx < 1 || gy < 1 || gx * gy > 64 || width < gx || height < gy) { ctx->set_error("select_grid: bad arguments (at most 32768 matches)"); return MI355_ERR_ARG; }
    // The kernel sorts by (d2, queryIdx); feed it ranks so that the given order is kept: d2 := position.
    // Queries are remapped to 0..n-1 in the given order (x/y/id carried through).
    const bool big = n > KSTRIDE;                           // more matches than select_kernel's LDS list holds: the large-pair form (keys sorted in HBM)
    const size_t K = big ? (size_t)n : (size_t)KSTRIDE;
    std::vector<float2> xy1(n > 0 ? n : 1), xy2(n > 0 ? n : 1);
    std::vector<int> idx(K, 0), dd(K, 0), d2nd(K, 0);
    for (int k = 0; k < n; k++) {
        const int q = sorted[k].queryIdx, t = sorted[k].trainIdx;
        if (q < 0 || q >= nk1 || t < 0 || t >= nk2) { ctx->set_error("select_grid: match index out of range"); return MI355_ERR_ARG; }
        xy1[k] = make_float2(kp1[2 * q], kp1[2 * q + 1]);
        xy2[k] = make_float2(kp2[2 * t], kp2[2 * t + 1]);
        idx[k] = k; dd[k] = k;
    }
    DevBuf& dx1 = ctx->buf("sg_xy1"); DevBuf& dx2 = ctx->buf("sg_xy2");
    DevBuf& dpd = ctx->buf("pair_desc"); DevBuf& didx = ctx->buf(big ? "nnb_idx" : "nn_idx"); DevBuf& dd2 = ctx->buf(big ? "nnb_d2" : "nn_d2"); DevBuf& d2n = ctx->buf(big ? "nnb_2nd" : "nn_2nd");
    DevBuf& ds1 = ctx->buf("sel1"); DevBuf& ds2 = ctx->buf("sel2"); DevBuf& dns = ctx->buf("nsel");
    MI_HIP(dx1.reserve(sizeof(float2) * xy1.size())); MI_HIP(dx2.reserve(sizeof(float2) * xy2.size()));
    MI_HIP(dpd.reserve(sizeof(PairDesc))); MI_HIP(didx.reserve(K * 4)); MI_HIP(dd2.reserve(K * 4)); MI_HIP(d2n.reserve(K * 4));
    MI_HIP(ds1.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED)); MI_HIP(ds2.reserve(sizeof(mi355_sfpoint) * MI355_MAX_SELECTED)); MI_HIP(dns.reserve(sizeof(int)));
    PairDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.xy_i = dx1.as<float2>(); pd.xy_j = dx2.as<float2>(); pd.n_i = n; pd.n_j = n > 0 ? 1 : 0; pd.width = width; pd.height = height;
    MI_HIP(hipMemcpyAsync(dx1.p, xy1.data(), sizeof(float2) * xy1.size(), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dx2.p, xy2.data(), sizeof(float2) * xy2.size(), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dpd.p, &pd, sizeof(pd), hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(didx.p, idx.data(), K * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(dd2.p, dd.data(), K * 4, hipMemcpyHostToDevice, ctx->stream));
    MI_HIP(hipMemcpyAsync(d2n.p, d2nd.data(), K * 4, hipMemcpyHostToDevice, ctx->stream));
    // nMatch is given by the caller here: choose (max_selected, fraction) that reproduce it: min(nMatch, 1.0*M)
    SelectParams sp;
    sp.max_selected = nMatch; sp.fraction = 1e9; sp.gx = gx; sp.gy = gy; sp.ratio2 = 0.0f;      // Min(nMatch, huge) = the caller's nMatch
    BigPairDev bp;
    memset(&bp, 0, sizeof(bp));
    if (big) {
        bp.n_i = n; bp.n_j = 1; bp.tc_n = 1; bp.mpad = 64; while (bp.mpad < n) bp.mpad <<= 1;
        DevBuf& dbp = ctx->buf("big_pairs"); DevBuf& dkeys = ctx->buf("sorted_keys_big");
        MI_HIP(dbp.reserve(sizeof(bp))); MI_HIP(dkeys.reserve((size_t)bp.mpad * 8));
        MI_HIP(hipMemcpyAsync(dbp.p, &bp, sizeof(bp), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(select_big_kernel, dim3(1), dim3(1024), 0, ctx->stream, dpd.as<PairDesc>(), dbp.as<BigPairDev>(), didx.as<int>(), dd2.as<int>(), d2n.as<int>(), sp,
                           dkeys.as<unsigned long long>(), ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>());
    } else {
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(256), 0, ctx->stream, dpd.as<PairDesc>(), didx.as<int>(), dd2.as<int>(), d2n.as<int>(),
                           sp, ds1.as<mi355_sfpoint>(), ds2.as<mi355_sfpoint>(), dns.as<int>(), (unsigned long long*)nullptr);
    }
    int cnt = 0;
    std::vector<mi355_sfpoint> h1(MI355_MAX_SELECTED), h2(MI355_MAX_SELECTED);
    MI_HIP(hipMemcpyAsync(&cnt, dns.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(h1.data(), ds1.p, sizeof(mi355_sfpoint) * MI355_MAX_SELECTED, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipMemcpyAsync(h2.data(), ds2.p, sizeof(mi355_sfpoint) * MI355_MAX_SELECTED, hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < cnt; k++) {
        const int pos = h1[k].id;                       // position in the caller's sorted list
        v1[k].x = h1[k].x; v1[k].y = h1[k].y; v1[k].id = sorted[pos].queryIdx;
        v2[k].x = h2[k].x; v2[k].y = h2[k].y; v2[k].id = sorted[pos].trainIdx;
    }
    *n_out = cnt;
    return MI355_OK;
}

// diagnostic (scratch/match_time.py): workgroups of bf_match_kernel the runtime keeps resident per CU
extern "C" int mi355_debug_match_occupancy() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bf_match_kernel<false>, BF_NT, 0) != hipSuccess) return -1;
    return nb;
}
