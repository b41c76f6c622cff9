// csrc/host_io.cpp -- the formats and the caller either side of the hot path.
// Host-only code (no device work): byte-compatible readers/writers of the reference's files and the
// linear system of the global affine alignment.
//
//   matchPairs.match   WriteMatchPairs / LoadMatchPairs     MosaicWithoutPos.cpp:4736-4749, 4774-4797
//   matchPairs.txt     WriteMatchPairs_ASC2                 MosaicWithoutPos.cpp:4751-4772
//   tran0.txt          OutTransform                         MosaicWithoutPos.cpp:2798-2818
//   keypoint_%d.key    WriteSurfKeyPoints/LoadSurfKeyPoints MosaicWithoutPos.cpp:4682-4734
//   BundleAdjustmentSparse (affine, image 0 fixed)          MosaicWithoutPos.cpp:6971-7202
#include "../../include/mi355_mosaic.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <string>
#include <fstream>
#include <atomic>
#include <thread>
#include <chrono>
#include <vector>

static_assert(sizeof(mi355_match_point_pairs) == 40, "MatchPointPairs must be 40 bytes (matchPairs.match record)");
static_assert(sizeof(mi355_keypoint) == 28, "cv::KeyPoint must be 28 bytes (keypoint_%d.key record)");
static_assert(sizeof(mi355_pair_result) == 9664, "pair result record size");

extern "C" int mi355_write_match_pairs(const char* path, const mi355_match_point_pairs* v, int n) {
    if (!path || n < 0 || (n > 0 && !v)) return MI355_ERR_ARG;
    if (n == 0) return MI355_OK;                    // the reference writes nothing for an empty vector (:4739)
    FILE* f = fopen(path, "wb");
    if (!f) return MI355_ERR_FAILED;
    const int32_t cnt = n;
    bool ok = fwrite(&cnt, sizeof(cnt), 1, f) == 1 && fwrite(v, sizeof(*v), (size_t)n, f) == (size_t)n;
    ok = (fclose(f) == 0) && ok;
    return ok ? MI355_OK : MI355_ERR_FAILED;
}

extern "C" int mi355_load_match_pairs(const char* path, mi355_match_point_pairs** v, int* n) {
    if (!path || !v || !n) return MI355_ERR_ARG;
    *v = nullptr; *n = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return MI355_ERR_ARG;                   // LoadMatchPairs returns -1 when the count cannot be read (:4781-4782)
    int32_t cnt = 0;
    if (fread(&cnt, sizeof(cnt), 1, f) != 1 || cnt < 0) { fclose(f); return MI355_ERR_ARG; }
    mi355_match_point_pairs* buf = (mi355_match_point_pairs*)malloc(sizeof(*buf) * (size_t)(cnt > 0 ? cnt : 1));
    if (!buf) { fclose(f); return MI355_ERR_NOMEM; }
    const size_t got = fread(buf, sizeof(*buf), (size_t)cnt, f);
    fclose(f);
    if (got != (size_t)cnt) { free(buf); return MI355_ERR_FAILED; }
    *v = buf; *n = cnt;
    return MI355_OK;
}

// operator<< of a float on a default ofstream: %g with 6 significant digits
extern "C" int mi355_write_match_pairs_txt(const char* path, const mi355_match_point_pairs* v, int n) {
    if (!path || n < 0 || (n > 0 && !v)) return MI355_ERR_ARG;
    std::ofstream out(path, std::ios::trunc);
    if (!out) return MI355_ERR_FAILED;
    for (int i = 0; i < n; i++)
        out << v[i].ptA_i << " " << v[i].ptA.x << " " << v[i].ptA.y << " " << v[i].ptA_Fixed << " "
            << v[i].ptB_i << " " << v[i].ptB.x << " " << v[i].ptB.y << " " << v[i].ptB_Fixed << std::endl;
    out.close();
    return out.fail() ? MI355_ERR_FAILED : MI355_OK;
}

extern "C" int mi355_write_transforms(const char* path, const mi355_image_transform* t, int n) {
    if (!path || n < 0 || (n > 0 && !t)) return MI355_ERR_ARG;
    std::ofstream out(path, std::ios::trunc);
    if (!out) return MI355_ERR_FAILED;
    for (int i = 1; i < n; i++) {                   // image 0 is the fixed reference and is not written (:2804)
        for (int j = 0; j < 8; j++) out << t[i].m[j] << " ";
        out << t[i].fixed;
        out << std::endl;
    }
    out.close();
    return out.fail() ? MI355_ERR_FAILED : MI355_OK;
}

// ImportTransform (MosaicWithoutPos.cpp:2820-2843): a count, then 9 floats per transform; fixed = 1 for the first, 0 for the rest.
extern "C" int mi355_load_transforms(const char* path, mi355_image_transform** t, int* n) {
    if (!path || !t || !n) return MI355_ERR_ARG;
    *t = nullptr; *n = 0;
    std::ifstream in(path);
    if (!in) return MI355_ERR_FAILED;
    int cnt = 0;
    in >> cnt;
    if (!in || cnt < 0 || cnt > 10000000) return MI355_ERR_FAILED;
    mi355_image_transform* v = (mi355_image_transform*)malloc(sizeof(mi355_image_transform) * (size_t)(cnt > 0 ? cnt : 1));
    if (!v) return MI355_ERR_NOMEM;
    for (int i = 0; i < cnt; i++) {
        for (int j = 0; j < 9; j++) in >> v[i].m[j];
        v[i].fixed = (i == 0) ? 1 : 0;
    }
    if (!in && cnt > 0) { free(v); return MI355_ERR_FAILED; }
    *t = v; *n = cnt;
    return MI355_OK;
}

// The file OutTransform writes (tran0.txt, :2798-2818): one row per image 1..n-1, "m0 .. m7 fixed".  Image 0 (the fixed reference,
// identity) is not in the file and is put back in front, so that n_images transforms come out; m8 = 1.
extern "C" int mi355_load_tran0(const char* path, mi355_image_transform** t, int* n) {
    if (!path || !t || !n) return MI355_ERR_ARG;
    *t = nullptr; *n = 0;
    std::ifstream in(path);
    if (!in) return MI355_ERR_FAILED;
    std::vector<mi355_image_transform> v(1);
    for (int j = 0; j < 9; j++) v[0].m[j] = (j % 4 == 0) ? 1.0f : 0.0f;
    v[0].fixed = 1;
    for (;;) {
        mi355_image_transform r;
        bool ok = true;
        for (int j = 0; j < 8 && ok; j++) ok = (bool)(in >> r.m[j]);
        if (!ok) break;
        if (!(in >> r.fixed)) return MI355_ERR_FAILED;               // a torn row
        r.m[8] = 1.0f;
        v.push_back(r);
    }
    mi355_image_transform* out = (mi355_image_transform*)malloc(sizeof(mi355_image_transform) * v.size());
    if (!out) return MI355_ERR_NOMEM;
    memcpy(out, v.data(), sizeof(mi355_image_transform) * v.size());
    *t = out; *n = (int)v.size();
    return MI355_OK;
}

extern "C" int mi355_write_keypoints(const char* path, const mi355_keypoint* kp, int n) {
    if (!path || n < 0 || (n > 0 && !kp)) return MI355_ERR_ARG;
    if (n == 0) return MI355_OK;                    // :4691 only written when non-empty
    FILE* f = fopen(path, "wb");
    if (!f) return MI355_ERR_FAILED;
    const int32_t cnt = n;
    bool ok = fwrite(&cnt, sizeof(cnt), 1, f) == 1 && fwrite(kp, sizeof(*kp), (size_t)n, f) == (size_t)n;
    ok = (fclose(f) == 0) && ok;
    return ok ? MI355_OK : MI355_ERR_FAILED;
}

extern "C" int mi355_load_keypoints(const char* path, mi355_keypoint** kp, int* n) {
    if (!path || !kp || !n) return MI355_ERR_ARG;
    *kp = nullptr; *n = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return MI355_ERR_ARG;
    int32_t cnt = 0;
    if (fread(&cnt, sizeof(cnt), 1, f) != 1 || cnt < 0) { fclose(f); return MI355_ERR_ARG; }
    mi355_keypoint* buf = (mi355_keypoint*)malloc(sizeof(*buf) * (size_t)(cnt > 0 ? cnt : 1));
    if (!buf) { fclose(f); return MI355_ERR_NOMEM; }
    const size_t got = fread(buf, sizeof(*buf), (size_t)cnt, f);
    fclose(f);
    if (got != (size_t)cnt) { free(buf); return MI355_ERR_FAILED; }
    *kp = buf; *n = cnt;
    return MI355_OK;
}

// MosaicWithoutPos.cpp:5201-5221: accepted pairs only, inlier order, ptA <- image i, ptB <- image j
extern "C" int mi355_results_to_match_pairs(const mi355_pair_result* r, int n_pairs, const int32_t* fixed_flags,
                                            mi355_match_point_pairs** v, int* n) {
    if (n_pairs < 0 || (n_pairs > 0 && !r) || !v || !n) return MI355_ERR_ARG;
    size_t total = 0;
    for (int p = 0; p < n_pairs; p++) if (r[p].accepted) total += (size_t)r[p].n_in;
    mi355_match_point_pairs* out = (mi355_match_point_pairs*)malloc(sizeof(*out) * (total > 0 ? total : 1));
    if (!out) return MI355_ERR_NOMEM;
    size_t k = 0;
    for (int p = 0; p < n_pairs; p++) {
        if (!r[p].accepted) continue;
        for (int q = 0; q < r[p].n_in; q++, k++) {
            out[k].ptA = r[p].a[q]; out[k].ptA_i = r[p].i; out[k].ptA_Fixed = fixed_flags ? fixed_flags[r[p].i] : 0;
            out[k].ptB = r[p].b[q]; out[k].ptB_i = r[p].j; out[k].ptB_Fixed = fixed_flags ? fixed_flags[r[p].j] : 0;
        }
    }
    *v = out; *n = (int)total;
    return MI355_OK;
}

// discriptor_%d.xml: cv::FileStorage << "descriptor" << Mat (CV_32F), as OpenCV 2.4's XML emitter lays it out (persistence.cpp, from memory of its
// rules -- the reference commits no such file, so the bytes are unpinned): a scalar of a sequence goes on the current line behind one blank
// unless the line would pass column 71 (then a new line, indented 4); floats that are integers print as "12.", the others as "%.8e"; the closing
// tags follow the last number on its line.
extern "C" int mi355_write_descriptors_xml(const char* path, const float* desc, int n_rows, int n_cols) {
    if (!path || n_rows < 0 || n_cols < 0 || ((size_t)n_rows * n_cols > 0 && !desc)) return MI355_ERR_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return MI355_ERR_FAILED;
    fprintf(f, "<?xml version=\"1.0\"?>\n<opencv_storage>\n<descriptor type_id=\"opencv-matrix\">\n  <rows>%d</rows>\n  <cols>%d</cols>\n  <dt>f</dt>\n  <data>", n_rows, n_cols);
    std::string line;                                   // the current line of the data block (empty: straight behind "<data>")
    bool first = true;
    const size_t total = (size_t)n_rows * (size_t)n_cols;
    for (size_t k = 0; k < total; k++) {
        char buf[64];
        const float v = desc[k];
        uint32_t bits; memcpy(&bits, &v, 4);
        if ((bits & 0x7f800000u) != 0x7f800000u) {
            const int iv = (int)lrintf(v);
            if ((float)iv == v) snprintf(buf, sizeof buf, "%d.", iv); else snprintf(buf, sizeof buf, "%.8e", v);
        } else if (bits & 0x7fffffu) snprintf(buf, sizeof buf, ".Nan");
        else snprintf(buf, sizeof buf, (bits >> 31) ? "-.Inf" : ".Inf");
        const size_t len = strlen(buf);
        if (first || (line.size() + len > 71 && line.size() + len - 4 > 10)) {
            if (!first) fputs(line.c_str(), f);
            fputc('\n', f);
            line.assign(4, ' ');
            first = false;
        } else line.push_back(' ');
        line += buf;
    }
    fputs(line.c_str(), f);
    fputs("</data></descriptor>\n</opencv_storage>\n", f);
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? MI355_OK : MI355_ERR_FAILED;
}
extern "C" int mi355_load_descriptors_xml(const char* path, float** desc, int* n_rows, int* n_cols) {
    if (!path || !desc || !n_rows || !n_cols) return MI355_ERR_ARG;
    *desc = nullptr; *n_rows = *n_cols = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return MI355_ERR_FAILED;
    std::string t;
    { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) t.append(buf, n); }
    fclose(f);
    auto field = [&t](const char* open_tag, const char* close_tag, size_t from, size_t& a, size_t& b) {
        a = t.find(open_tag, from);
        if (a == std::string::npos) return false;
        a += strlen(open_tag);
        b = t.find(close_tag, a);
        return b != std::string::npos;
    };
    size_t d0 = t.find("<descriptor");
    if (d0 == std::string::npos) return MI355_ERR_FAILED;
    size_t a, b;
    if (!field("<rows>", "</rows>", d0, a, b)) return MI355_ERR_FAILED;
    const long rows = strtol(t.c_str() + a, nullptr, 10);
    if (!field("<cols>", "</cols>", d0, a, b)) return MI355_ERR_FAILED;
    const long cols = strtol(t.c_str() + a, nullptr, 10);
    if (!field("<dt>", "</dt>", d0, a, b) || t.compare(a, b - a, "f") != 0) return MI355_ERR_FAILED;      // CV_32F, one channel: what the reference writes
    if (rows < 0 || cols < 0 || rows > (1 << 24) || cols > (1 << 16)) return MI355_ERR_FAILED;
    if (!field("<data>", "</data>", d0, a, b)) return MI355_ERR_FAILED;
    const size_t total = (size_t)rows * (size_t)cols;
    float* out = (float*)malloc(sizeof(float) * (total ? total : 1));
    if (!out) return MI355_ERR_NOMEM;
    const char* p = t.c_str() + a; const char* end = t.c_str() + b;
    size_t k = 0;
    while (k < total) {
        while (p < end && isspace((unsigned char)*p)) p++;
        if (p >= end) break;
        float v;
        if (!strncmp(p, ".Nan", 4)) { v = NAN; p += 4; }
        else if (!strncmp(p, ".Inf", 4)) { v = INFINITY; p += 4; }
        else if (!strncmp(p, "-.Inf", 5)) { v = -INFINITY; p += 5; }
        else { char* q = nullptr; v = strtof(p, &q); if (q == p) break; p = q; }
        out[k++] = v;
    }
    if (k != total) { free(out); return MI355_ERR_FAILED; }
    *desc = out; *n_rows = (int)rows; *n_cols = (int)cols;
    return MI355_OK;
}

// Global affine alignment.  Every correspondence (A in image a, B in image b) contributes the two equations
//   T_a(A) - T_b(B) = 0,  T_k(x,y) = (m0 x + m1 y + m2, m3 x + m4 y + m5),
// with T_k = identity for fixed images (their terms move to the right-hand side) -- the system
// BundleAdjustmentSparse builds as a sparse 2P x 6(N-F) matrix and solves through CHOLMOD's normal
// equations (MosaicWithoutPos.cpp:6971-7202, test_cholmod.cpp:180-262).  Here the 6(N-F) normal matrix
// is accumulated directly in double (x- and y-rows decouple into two identical 3(N-F) systems) and
// factorised by dense Cholesky.  Known answer: tests/golden/matchPairs.txt -> tran0.txt.
// The work is in two places: the second moments of the correspondences (21 products per point; C5: 28 M points) and the banded
// Cholesky factorisation (C5: 6000 rows x half bandwidth 545).  Both run on a few host threads in a way that keeps every sum's
// order: a thread owns whole image pairs (their moments are summed per pair first, then added to the blocks in pair order by one
// thread, as the serial code did) and whole matrix entries (the products of an entry are subtracted one by one, ascending k, panel
// after panel) -- the result does not depend on the number of threads or on the panel width.
namespace {
int host_threads() {
    static const int n = [] {
        const char* e = getenv("MI355_HOST_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (!e) {
            // several ranks of one node call the alignment at the same moment (the host step is replicated): share the cores between them,
            // a team of spinning threads per rank on oversubscribed cores is what the yield below exists for
            const char* lw = getenv("LOCAL_WORLD_SIZE");
            const int ranks = lw ? atoi(lw) : 1;
            if (ranks > 1) v /= ranks;
        }
        if (v > 32) v = 32;
        return v < 1 ? 1 : v;
    }();
    return n;
}
inline void cpu_relax() {      // a waiting thread must not take issue slots from the hardware thread next to it (it may be the one that works)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
}
template <class F> void parallel_chunks(size_t n, int threads, F&& f) {          // f(begin, end) on contiguous chunks
    if (threads <= 1 || n < 2) { f((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 1; t < threads; t++) { const size_t lo = per * t, hi = lo + per < n ? lo + per : n; if (lo < hi) th.emplace_back([&f, lo, hi] { f(lo, hi); }); }
    f((size_t)0, per < n ? per : n);
    for (auto& x : th) x.join();
}
// Phase (C) of the banded Cholesky below for one row: s[j] -= l_k * pt_k[j] for the panel's columns k in ascending order, j over the row's
// trailing entries.  pt_k = the panel's column k laid out along j (a transposed copy made in phase (B)), so the j loop runs over contiguous
// doubles and the compiler vectorises it; every entry still takes its products one by one in k order, each product rounded, then the
// difference (this file is compiled with -ffp-contract=off): the same bits as the scalar form, for any vector width.  Clones for the
// host's vector unit are picked when the library is loaded (the build machine need not be the machine that runs).
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__) || defined(__SANITIZE_THREAD__) || defined(__SANITIZE_ADDRESS__)      // (an ifunc resolver runs before a sanitizer's runtime is up)
#define MI355_SIMD_CLONES
#else
#define MI355_SIMD_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif
MI355_SIMD_CLONES static void chol_row_update(double* __restrict s, int n, const double* __restrict l, int nk, const double* __restrict pt, size_t pt_stride) {
    // four columns of the panel per walk over the row: an entry is loaded and stored once for four subtractions (one column per walk made the
    // loop memory-bound: C5-sized factorisation 58 -> 38 ms on one thread of the box); per entry the same subtractions in the same order
    int k = 0;
    for (; k + 4 <= nk; k += 4) {
        const double l0 = l[k], l1 = l[k + 1], l2 = l[k + 2], l3 = l[k + 3];
        const double* __restrict p0 = pt + (size_t)k * pt_stride;
        const double* __restrict p1 = p0 + pt_stride;
        const double* __restrict p2 = p1 + pt_stride;
        const double* __restrict p3 = p2 + pt_stride;
        for (int j = 0; j < n; j++) {
            double v = s[j];
            v -= l0 * p0[j]; v -= l1 * p1[j]; v -= l2 * p2[j]; v -= l3 * p3[j];
            s[j] = v;
        }
    }
    for (; k < nk; k++) {
        const double lk = l[k];
        const double* __restrict p = pt + (size_t)k * pt_stride;
        for (int j = 0; j < n; j++) s[j] -= lk * p[j];
    }
}
// Phase (B) for one row below the panel's diagonal block, right-looking: entry j is final once the columns before it have left it, is divided by
// the diagonal, and leaves the row's later entries at once -- s[q] -= L(i, j) * L(j + 1 + q, j), the column of the diagonal block laid out along
// q (dt) -- so the walk is over contiguous doubles; an entry still takes its products one by one in ascending k.  (Columns left of a later
// entry's own envelope contribute products with an exact zero.)
MI355_SIMD_CLONES static void chol_row_forward(double* __restrict r /* entries ja .. jb - 1 of the row */, int n, const double* __restrict diag /* L(j, j) */,
                                               const double* __restrict dt /* dt[j * dts + q] = L(ja + j + 1 + q, ja + j) */, size_t dts, double* __restrict pt /* column of the transposed panel copy */, size_t pts) {
    for (int j = 0; j < n; j++) {
        const double sv = r[j] / diag[j];
        r[j] = sv;
        pt[(size_t)j * pts] = sv;
        const double* __restrict d = dt + (size_t)j * dts;
        double* __restrict s = r + j + 1;
        const int m = n - j - 1;
        for (int q = 0; q < m; q++) s[q] -= sv * d[q];
    }
}

struct PairGroup { int a, b; size_t first; int count; };                          // correspondences `first .. first + count` belong to images (a, b)
struct Moments { double aa[6], ab[9], bb[6]; };         // second moments of one image pair's correspondences: sum ca ca^T (lower triangle), sum ca cb^T, sum cb cb^T; ca = (xa, ya, 1)

// The system from the image pairs' second moments (on the host: align_core below; formed on the device: mi355_pair_moments_dev).  The right-hand
// sides need no sums of their own: with image a fixed the equations' constant terms are -A, and sum cb (-xa) = -(sum ca cb^T) row 0 -- the same
// products, negated; with b fixed sum ca xb = column 0 of that matrix.
int align_from_moments(const std::vector<PairGroup>& groups, const std::vector<Moments>& mom, size_t npoints_hint, int n_images, const int32_t* fixed, mi355_image_transform* out) {
    static const bool align_dbg = getenv("MI355_ALIGN_DBG") != nullptr;
    const auto tdbg0 = std::chrono::steady_clock::now();
    auto tdbg = [&](const char* what) { if (align_dbg) fprintf(stderr, "[align] %-10s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tdbg0).count()); };
    std::vector<int> col(n_images, -1);
    int nf = 0;
    for (int k = 0; k < n_images; k++) {
        const bool fx = fixed ? fixed[k] != 0 : (k == 0);
        if (!fx) col[k] = nf++;
    }
    for (int k = 0; k < n_images; k++) {
        for (int i = 0; i < 9; i++) out[k].m[i] = 0.0f;
        out[k].m[0] = out[k].m[4] = out[k].m[8] = 1.0f;
        out[k].fixed = col[k] < 0 ? 1 : 0;
    }
    if (nf == 0) return MI355_OK;
    const int D = 3 * nf;
    // The normal matrix is banded: with the free images in index order, N(i,j) != 0 only for |block(i) - block(j)| <= max |a - b| over
    // the pairs (adjacent-pair strips: 1 block; the reference's window: 181 blocks), and the Cholesky factor keeps that band.  Only the
    // lower band is stored: entry (i, j), i - bw <= j <= i, at Nb[i * W + (j - i + bw)] (a dense D x D matrix was 18 MB to clear per
    // call at 500 images, 288 MB at 2000).
    int bwb = 0;
    (void)npoints_hint;
    for (const PairGroup& g : groups) {
        if (g.a < 0 || g.a >= n_images || g.b < 0 || g.b >= n_images) return MI355_ERR_ARG;
        const int oa = col[g.a], ob = col[g.b];
        if (oa >= 0 && ob >= 0) { const int d = oa > ob ? oa - ob : ob - oa; if (d > bwb) bwb = d; }
    }
    const int bw = 3 * bwb + 2;                      // half bandwidth in scalar rows
    const size_t W = (size_t)bw + 1;
    // The band (26 MB at C5) lives in a buffer the calling thread keeps between calls: a fresh vector paid a page fault per 4 KB on every call
    // (4-13 ms of the C5 alignment, more than a third of the factorisation it precedes); cleared here, in parallel for the big ones.
    static thread_local std::vector<double> Nb_keep;
    if (Nb_keep.size() < (size_t)D * W) { std::vector<double>().swap(Nb_keep); Nb_keep.resize((size_t)D * W); }
    std::vector<double>& Nb = Nb_keep;
    parallel_chunks((size_t)D * W, (size_t)D * W > ((size_t)1 << 21) ? (host_threads() < 8 ? host_threads() : 8) : 1, [&](size_t lo, size_t hi) { memset(Nb.data() + lo, 0, (hi - lo) * sizeof(double)); });
    std::vector<double> bx(D, 0.0), by(D, 0.0);
    auto NL = [&](int i, int j) -> double& { return Nb[(size_t)i * W + (size_t)(j - i + bw)]; };     // i >= j >= i - bw
    // rows: coefficients [xa ya 1] on image a's columns, -[xb yb 1] on image b's columns
    tdbg("setup");
    tdbg("moments");
    for (size_t gi = 0; gi < groups.size(); gi++) {
        const PairGroup& g = groups[gi];
        const int oa = col[g.a], ob = col[g.b];
        if (oa < 0 && ob < 0) continue;
        const Moments& m = mom[gi];
        double vax[3] = {0.0, 0.0, 0.0}, vay[3] = {0.0, 0.0, 0.0}, vbx[3] = {0.0, 0.0, 0.0}, vby[3] = {0.0, 0.0, 0.0};
        if (oa < 0) for (int i = 0; i < 3; i++) { vbx[i] = 0.0 - m.ab[i]; vby[i] = 0.0 - m.ab[3 + i]; }      // a fixed: sum cb[i] (-xa), sum cb[i] (-ya)
        if (ob < 0) for (int i = 0; i < 3; i++) { vax[i] = m.ab[3 * i]; vay[i] = m.ab[3 * i + 1]; }            // b fixed: sum ca[i] xb, sum ca[i] yb
        int t = 0;
        if (oa >= 0) for (int i = 0; i < 3; i++) { bx[3 * oa + i] += vax[i]; by[3 * oa + i] += vay[i]; for (int j = 0; j <= i; j++) NL(3 * oa + i, 3 * oa + j) += m.aa[t++]; }
        t = 0;
        if (ob >= 0) for (int i = 0; i < 3; i++) { bx[3 * ob + i] -= vbx[i]; by[3 * ob + i] -= vby[i]; for (int j = 0; j <= i; j++) NL(3 * ob + i, 3 * ob + j) += m.bb[t++]; }
        if (oa >= 0 && ob >= 0 && oa != ob) {
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    // N(a_i, b_j) = -Mab[i][j] = N(b_j, a_i): store the entry of the lower triangle
                    if (oa > ob) NL(3 * oa + i, 3 * ob + j) -= m.ab[3 * i + j]; else NL(3 * ob + j, 3 * oa + i) -= m.ab[3 * i + j];
                }
        }
    }
    // images without any correspondence would make N singular: pin them to identity (the driver flags them
    // invalid through Select_Connected_Matched_Images before calling, MosaicWithoutPos.cpp:4503-4523)
    for (int k = 0; k < n_images; k++) {
        const int o = col[k];
        if (o < 0) continue;
        if (NL(3 * o + 2, 3 * o + 2) == 0.0) {
            for (int i = 0; i < 3; i++) NL(3 * o + i, 3 * o + i) = 1.0;
            bx[3 * o + 0] = 1.0; by[3 * o + 1] = 1.0;
        }
    }
    // Cholesky N = L L^T (lower), in place, inside the band, in panels of 64 columns.  Every entry (i, j) still receives the same
    // operations in the same order as in the column-by-column form -- N(i, j) minus L(i, k) L(j, k) for ascending k, each product
    // subtracted on its own, then the division by the diagonal -- so the factor has the same bits whatever the panel width and the
    // number of threads; what changes is who waits for whom: a team walking the columns together met at one barrier per column
    // (1497 at C4, 6000 at C5: half of the time), and one dot product per entry is a single dependent chain of subtractions.  Per
    // panel: (A) one thread factors the 64 x 64 diagonal block, (B) the rows below it are divided up between the threads, (C) the
    // panel's products are subtracted from the trailing band, again by rows, four independent entries at a time.  Three barriers per
    // panel.
    tdbg("assembled");
    // The band's width is set by the ONE pair with the largest index distance (a window survey accepts a few pairs up to 181 images apart
    // among ~2000 between neighbours <= 51 apart): most rows are much shorter than the band.  fst[i] = first column of row i that can be
    // non-zero; the factor's fill-in stays inside each row's own envelope (profile Cholesky), so everything left of fst[i] is skipped --
    // those entries are exact zeros, skipping them changes no value -- and the work is sum (i - fst[i])^2 instead of D * bw^2 (C4: 12 x less).
    std::vector<int> fst(D);
    for (int o = 0; o < nf; o++) for (int t = 0; t < 3; t++) fst[3 * o + t] = 3 * o;
    for (const PairGroup& g : groups) {
        const int oa = col[g.a], ob = col[g.b];
        if (oa < 0 || ob < 0 || oa == ob) continue;
        const int hi = oa > ob ? oa : ob, lo = oa > ob ? ob : oa;
        for (int t = 0; t < 3; t++) if (3 * lo < fst[3 * hi + t]) fst[3 * hi + t] = 3 * lo;
    }
    double work = 0.0;
    for (int i = 0; i < D; i++) work += (double)(i - fst[i]) * (double)(i - fst[i]);
    // one thread per 2.5e7 multiply-subtracts, up to 16 (measured on the GPU box's host, C5's size, 5e8: 34 ms on one thread, 20 on four, 14 on eight,
    // 12 on sixteen, 11.5 on thirty-two; C4's 3.6e7 stay on one thread: 1.3 ms, starting a team costs about that).  The team scales only since rows
    // keep their owner and eight neighbouring rows share one (see `mine` below): before, two threads took twice as long as one.
    int team = (int)(work / 2.5e7);
    if (team > 16) team = 16;
    if (team > host_threads()) team = host_threads();
    if (team < 1) team = 1;
    constexpr int PW = 64;
    const size_t pts = ((size_t)bw + PW + 7) & ~(size_t)7;       // a panel's columns, transposed: PT[k - p0][i - p1] = L(i, k) for the rows i below the block
    std::vector<double> PT_store((size_t)PW * pts + 8, 0.0), DT((size_t)PW * (PW + 1), 0.0), DG(PW, 1.0);      // DT[(k - p0) * (PW + 1) + (j - k - 1)] = L(j, k) of the diagonal block, j > k
    double* const PT = PT_store.data() + ((64 - (reinterpret_cast<uintptr_t>(PT_store.data()) & 63)) & 63) / sizeof(double);      // on a cache-line boundary: eight rows' entries of a column share a line, and eight rows share an owner
    std::atomic<int> arrived{0}, generation{0}, failed{0};
    auto rowp = [&](int i) -> double* { return Nb.data() + (ptrdiff_t)i * (ptrdiff_t)W + (ptrdiff_t)(bw - i); };     // rowp(i)[j] = N(i, j), i - bw <= j <= i
    auto worker = [&](int tid) {
        int gen = 0;
        auto barrier = [&]() {
            if (team <= 1) return;
            gen++;
            if (arrived.fetch_add(1) + 1 == team) { arrived.store(0); generation.store(gen); }
            else { int spins = 0; while (generation.load(std::memory_order_acquire) < gen) { cpu_relax(); if (++spins > 4096) { std::this_thread::yield(); spins = 4000; } } }   // a phase takes microseconds: spin, but yield when the host has fewer cores than threads
        };
        for (int p0 = 0; p0 < D; p0 += PW) {
            const int p1 = p0 + PW < D ? p0 + PW : D;
            if (tid == 0) {                                       // (A) the diagonal block
                for (int j = p0; j < p1; j++) {
                    double* rj = rowp(j);
                    double d = rj[j];
                    for (int k = (fst[j] > p0 ? fst[j] : p0); k < j; k++) d -= rj[k] * rj[k];
                    if (!(d > 0.0)) { failed.store(1); d = 1.0; }        // keep walking so that the team stays in step; the caller sees `failed`
                    d = std::sqrt(d);
                    rj[j] = d;
                    for (int i = j + 1; i < p1 && i <= j + bw; i++) {    // rows of the block that reach column j (half bandwidth below the panel width: adjacent-pair strips)
                        if (fst[i] > j) continue;
                        double* ri = rowp(i);
                        double sv = ri[j];
                        const int ka = fst[i] > fst[j] ? fst[i] : fst[j];
                        for (int k = (ka > p0 ? ka : p0); k < j; k++) sv -= ri[k] * rj[k];
                        ri[j] = sv / d;
                    }
                }
            }
            if (tid == 0) {                                       // the block's columns laid out along the rows (phase B walks them), zeros outside a row's envelope, and its diagonal
                for (int k = p0; k < p1; k++) {
                    DG[k - p0] = rowp(k)[k];
                    for (int j = k + 1; j < p1; j++) DT[(size_t)(k - p0) * (PW + 1) + (size_t)(j - k - 1)] = (fst[j] <= k && j - k <= bw) ? rowp(j)[k] : 0.0;
                }
            }
            barrier();
            const int i_end = p1 - 1 + bw < D - 1 ? p1 - 1 + bw : D - 1;             // last row that holds an entry in a column of the panel
            // rows dealt out one by one: row i of the trailing band has i - p1 + 1 entries to update, contiguous chunks would give the last
            // thread twice the mean
            // rows belong to threads in groups of eight, the same thread in every panel ((i / 8) mod team): a row's entries stay in that core's
            // cache from panel to panel, and the eight doubles of a line of the transposed panel copy have one writer (dealt row by row from the
            // panel's first row on, a row changed hands with every panel and neighbouring rows wrote into the same lines: two threads took twice
            // as long as one)
            auto mine = [&](int i) { return ((i >> 3) % team) == tid; };
            for (int i = p1; i <= i_end; i++) {       // (B) the panel's columns of the rows below the block
                if (!mine(i)) continue;
                double* ri = rowp(i);
                const int j0 = fst[i] > p0 ? fst[i] : p0;
                for (int j = p0; j < j0 && j < p1; j++) PT[(size_t)(j - p0) * pts + (size_t)(i - p1)] = 0.0;      // left of the row's envelope (or the row does not reach the panel at all)
                if (j0 < p1) chol_row_forward(ri + j0, p1 - j0, DG.data() + (j0 - p0), DT.data() + (size_t)(j0 - p0) * (PW + 1), (size_t)PW + 1, PT + (size_t)(j0 - p0) * pts + (size_t)(i - p1), pts);
            }
            barrier();
            // (C) the panel's products leave the trailing band, tile by tile of 64 trailing columns: the tile's part of the transposed panel copy
            // (64 x 64 doubles) stays in the first-level cache while the rows pass (row by row over the whole band it was streamed from the
            // second level once per row: 16 GB at C5)
            constexpr int JT = 64;
            for (int jt = p1; jt <= i_end; jt += JT) {
                for (int i = (jt > p1 ? jt : p1); i <= i_end; i++) {
                    if (!mine(i) || fst[i] >= p1) continue;          // above the tile / the row holds nothing in the panel's columns
                    double* ri = rowp(i);
                    const int k0 = fst[i] > p0 ? fst[i] : p0;
                    const int j0 = fst[i] > p1 ? fst[i] : p1;
                    const int ja = j0 > jt ? j0 : jt, jb = i < jt + JT - 1 ? i : jt + JT - 1;
                    if (ja > jb) continue;
                    // the transposed copy holds L(j, k) for every row j of the trailing band, zeros left of row j's envelope
                    chol_row_update(ri + ja, jb - ja + 1, ri + k0, p1 - k0, PT + (size_t)(k0 - p0) * pts + (size_t)(ja - p1), pts);
                }
            }
            barrier();
        }
    };
    if (team > 1) {
        std::vector<std::thread> th;
        for (int t = 1; t < team; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto& x : th) x.join();
    } else worker(0);
    tdbg("cholesky");
    if (failed.load()) return MI355_ERR_FAILED;
    // L y = b, L^T x = y for both right-hand sides in one walk (two independent chains of subtractions; the same operations per side as before)
    for (int i = 0; i < D; i++) {
        const int k0 = fst[i];
        const double* ri = rowp(i);
        double sx = bx[i], sy = by[i];
        for (int k = k0; k < i; k++) { const double l = ri[k]; sx -= l * bx[k]; sy -= l * by[k]; }
        bx[i] = sx / ri[i]; by[i] = sy / ri[i];
    }
    // L^T x = y, row-oriented: once x[i] is final, row i of L (contiguous) takes its products out of the entries above it.  An entry y[k] thus
    // loses L(i, k) x[i] for DESCENDING i -- a fixed order, whatever the thread count.  (Until round 5 the walk went down column i of L for every
    // i: one entry per row of the band, W doubles apart -- 3.3 M cache misses at C5, as long as half the factorisation on 16 threads.)
    for (int i = D - 1; i >= 0; i--) {
        const double* ri = rowp(i);
        const double xi = bx[i] / ri[i], yi = by[i] / ri[i];
        bx[i] = xi; by[i] = yi;
        for (int k = fst[i]; k < i; k++) { const double l = ri[k]; bx[k] -= l * xi; by[k] -= l * yi; }
    }
    tdbg("solved");
    for (int k = 0; k < n_images; k++) {
        const int o = col[k];
        if (o < 0) continue;
        out[k].m[0] = (float)bx[3 * o]; out[k].m[1] = (float)bx[3 * o + 1]; out[k].m[2] = (float)bx[3 * o + 2];
        out[k].m[3] = (float)by[3 * o]; out[k].m[4] = (float)by[3 * o + 1]; out[k].m[5] = (float)by[3 * o + 2];
    }
    return MI355_OK;
}

// the second moments of a pair's correspondences, summed in correspondence order (21 products per point; mi355_pair_moments_dev forms the same sums)
template <class XY>
inline void pair_moments_host(const PairGroup& g, XY& xy, Moments& m) {
    double Maa[3][3] = {{0}}, Mab[3][3] = {{0}}, Mbb[3][3] = {{0}};
    for (int k = 0; k < g.count; k++) {
        float fxa, fya, fxb, fyb;
        xy(g.first + (size_t)k, fxa, fya, fxb, fyb);
        const double ca[3] = {fxa, fya, 1.0}, cb[3] = {fxb, fyb, 1.0};
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j <= i; j++) { Maa[i][j] += ca[i] * ca[j]; Mbb[i][j] += cb[i] * cb[j]; }
            for (int j = 0; j < 3; j++) Mab[i][j] += ca[i] * cb[j];
        }
    }
    int t = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++, t++) { m.aa[t] = Maa[i][j]; m.bb[t] = Mbb[i][j]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.ab[3 * i + j] = Mab[i][j];
}
// xy(k, xa, ya, xb, yb): the k-th correspondence.  A thread owns whole image pairs (their moments are summed per pair first, then added to the
// blocks in pair order by one thread): the result does not depend on the number of threads.
template <class XY>
int align_core(const std::vector<PairGroup>& groups, XY&& xy, int n_images, const int32_t* fixed, mi355_image_transform* out) {
    size_t npoints = 0;
    for (const PairGroup& g : groups) npoints += (size_t)g.count;
    std::vector<Moments> mom(groups.size());
    parallel_chunks(groups.size(), npoints > 60000 ? (host_threads() < 8 ? host_threads() : 8) : 1, [&](size_t g0, size_t g1) {      // 3.2 ms on one thread at C4, 0.7 on eight; more threads only add their start-up
        for (size_t gi = g0; gi < g1; gi++) pair_moments_host(groups[gi], xy, mom[gi]);
    });
    return align_from_moments(groups, mom, npoints, n_images, fixed, out);
}
}  // namespace

extern "C" int mi355_global_affine_align(const mi355_match_point_pairs* v, int n, int n_images, const int32_t* fixed,
                                         mi355_image_transform* out) {
    if (n < 0 || n_images <= 0 || (n > 0 && !v) || !out) return MI355_ERR_ARG;
    std::vector<PairGroup> groups;                      // the records of one image pair are consecutive (mi355_results_to_match_pairs)
    for (int p = 0; p < n;) {
        int q = p;
        while (q < n && v[q].ptA_i == v[p].ptA_i && v[q].ptB_i == v[p].ptB_i) q++;
        groups.push_back(PairGroup{v[p].ptA_i, v[p].ptB_i, (size_t)p, q - p});
        p = q;
    }
    return align_core(groups, [v](size_t k, float& xa, float& ya, float& xb, float& yb) { xa = v[k].ptA.x; ya = v[k].ptA.y; xb = v[k].ptB.x; yb = v[k].ptB.y; },
                      n_images, fixed, out);
}

// The same system straight from the pair records (no m_vecMatchPairs copy: at C5 that vector is 28 M records = 1.1 GB): accepted pairs
// only, and with `label` only the pairs whose two images carry a non-zero label (the driver drops the others, :4512-4523).
extern "C" int mi355_global_affine_align_results(const mi355_pair_result* r, int n_pairs, int n_images, const int32_t* fixed,
                                                 const int32_t* label, mi355_image_transform* out) {
    if (n_pairs < 0 || n_images <= 0 || (n_pairs > 0 && !r) || !out) return MI355_ERR_ARG;
    std::vector<PairGroup> groups;
    for (int p = 0; p < n_pairs; p++) {
        if (!r[p].accepted || r[p].n_in <= 0) continue;
        if (r[p].i < 0 || r[p].i >= n_images || r[p].j < 0 || r[p].j >= n_images || r[p].n_in > MI355_MAX_SELECTED) return MI355_ERR_ARG;
        if (label && !(label[r[p].i] && label[r[p].j])) continue;
        groups.push_back(PairGroup{r[p].i, r[p].j, (size_t)p * MI355_MAX_SELECTED, r[p].n_in});
    }
    return align_core(groups, [r](size_t k, float& xa, float& ya, float& xb, float& yb) {
                          const mi355_pair_result& e = r[k / MI355_MAX_SELECTED]; const size_t q = k % MI355_MAX_SELECTED;
                          xa = e.a[q].x; ya = e.a[q].y; xb = e.b[q].x; yb = e.b[q].y; },
                      n_images, fixed, out);
}

// ... and from the pairs' second moments (formed on the device by mi355_pair_moments_dev, or here): the same system, the same bits
static_assert(sizeof(mi355_pair_moments) == 184 && sizeof(Moments) == 21 * sizeof(double), "moment record layout");
extern "C" int mi355_pair_moments_host(const mi355_pair_result* r, int n, mi355_pair_moments* out) {
    if (n < 0 || (n > 0 && (!r || !out))) return MI355_ERR_ARG;
    for (int p = 0; p < n; p++) {
        mi355_pair_moments& o = out[p];
        memset(&o, 0, sizeof(o));
        o.i = r[p].i; o.j = r[p].j;
        if (!r[p].accepted || r[p].n_in <= 0) continue;
        if (r[p].n_in > MI355_MAX_SELECTED) return MI355_ERR_ARG;
        o.n_in = r[p].n_in;
        const mi355_pair_result& e = r[p];
        auto xy = [&e](size_t k, float& xa, float& ya, float& xb, float& yb) { xa = e.a[k].x; ya = e.a[k].y; xb = e.b[k].x; yb = e.b[k].y; };
        Moments m;
        pair_moments_host(PairGroup{e.i, e.j, 0, e.n_in}, xy, m);
        memcpy(o.aa, m.aa, sizeof(m.aa)); memcpy(o.ab, m.ab, sizeof(m.ab)); memcpy(o.bb, m.bb, sizeof(m.bb));
    }
    return MI355_OK;
}
extern "C" int mi355_global_affine_align_moments(const mi355_pair_moments* mm, int n, int n_images, const int32_t* fixed, const int32_t* label,
                                                 mi355_image_transform* out) {
    if (n < 0 || n_images <= 0 || (n > 0 && !mm) || !out) return MI355_ERR_ARG;
    // the two lists keep their storage between calls (per calling thread): 116 000 accepted pairs at C5 are 20 MB that a fresh vector pays
    // for in page faults on every step (a third of the call, measured on the box's host: 32 ms per call around 21 ms of solver)
    static thread_local std::vector<PairGroup> groups;
    static thread_local std::vector<Moments> mom;
    groups.clear(); mom.clear();
    if (groups.capacity() < (size_t)n) groups.reserve((size_t)n);
    if (mom.capacity() < (size_t)n) mom.reserve((size_t)n);
    size_t npoints = 0;
    for (int p = 0; p < n; p++) {
        if (mm[p].n_in <= 0) continue;
        if (mm[p].i < 0 || mm[p].i >= n_images || mm[p].j < 0 || mm[p].j >= n_images || mm[p].n_in > MI355_MAX_SELECTED) return MI355_ERR_ARG;
        if (label && !(label[mm[p].i] && label[mm[p].j])) continue;
        groups.push_back(PairGroup{mm[p].i, mm[p].j, (size_t)p, mm[p].n_in});
        Moments m;
        memcpy(m.aa, mm[p].aa, sizeof(m.aa)); memcpy(m.ab, mm[p].ab, sizeof(m.ab)); memcpy(m.bb, mm[p].bb, sizeof(m.bb));
        mom.push_back(m);
        npoints += (size_t)mm[p].n_in;
    }
    return align_from_moments(groups, mom, npoints, n_images, fixed, out);
}

// Select_Connected_Matched_Images, MosaicWithoutPos.cpp:2754-2796: images are nodes, every image pair that has at
// least one correspondence is an edge; label the largest connected group (union-find here instead of the
// reference's O(E^2) cluster merge, same partition).
namespace {
template <class Edge> int select_connected_core(int n, Edge&& edge, int n_images, int32_t* label) {
    std::vector<int> parent(n_images), size(n_images, 1), touched(n_images, 0);
    for (int i = 0; i < n_images; i++) parent[i] = i;
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (int p = 0; p < n; p++) {
        int a, b;
        if (!edge(p, a, b)) continue;
        if (a < 0 || a >= n_images || b < 0 || b >= n_images) return MI355_ERR_ARG;
        touched[a] = touched[b] = 1;
        int ra = find(a), rb = find(b);
        if (ra == rb) continue;
        if (ra > rb) { int t = ra; ra = rb; rb = t; }          // root = lowest index of the group
        parent[rb] = ra; size[ra] += size[rb];
    }
    int best = -1, best_size = 0;
    for (int i = 0; i < n_images; i++) if (touched[i] && find(i) == i && size[i] > best_size) { best = i; best_size = size[i]; }
    for (int i = 0; i < n_images; i++) label[i] = (best >= 0 && touched[i] && find(i) == best) ? 1 : 0;
    return MI355_OK;
}
}  // namespace

extern "C" int mi355_select_connected(const mi355_match_point_pairs* v, int n, int n_images, int32_t* label) {
    if (n < 0 || n_images <= 0 || (n > 0 && !v) || !label) return MI355_ERR_ARG;
    return select_connected_core(n, [v](int p, int& a, int& b) { a = v[p].ptA_i; b = v[p].ptB_i; return true; }, n_images, label);
}

// the same labelling straight from the pair records: an accepted pair with at least one inlier is an edge
extern "C" int mi355_select_connected_moments(const mi355_pair_moments* m, int n, int n_images, int32_t* label) {
    if (n < 0 || n_images <= 0 || (n > 0 && !m) || !label) return MI355_ERR_ARG;
    return select_connected_core(n, [m](int p, int& a, int& b) { a = m[p].i; b = m[p].j; return m[p].n_in > 0; }, n_images, label);
}
extern "C" int mi355_select_connected_results(const mi355_pair_result* r, int n_pairs, int n_images, int32_t* label) {
    if (n_pairs < 0 || n_images <= 0 || (n_pairs > 0 && !r) || !label) return MI355_ERR_ARG;
    return select_connected_core(n_pairs, [r](int p, int& a, int& b) { a = r[p].i; b = r[p].j; return r[p].accepted && r[p].n_in > 0; }, n_images, label);
}
