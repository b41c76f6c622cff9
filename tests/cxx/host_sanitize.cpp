// tests/cxx/host_sanitize.cpp -- the host-only entry points of the library (csrc/host_io.cpp, csrc/overlap.cpp: file formats,
// result flattening, connected component, global affine alignment, ResampleByOverlap) under AddressSanitizer +
// UndefinedBehaviorSanitizer.  Built by tests/test_sanitize.py from those two translation units directly (no HIP needed) and run
// on the reference's committed matchPairs.match / matchPairs.txt / tran0.txt.  Exit code 0 = no finding (the sanitizers abort).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "mi355_mosaic.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static int work(const std::string& gold, const std::string& tmp, int tag) {
    mi355_match_point_pairs* v = NULL; int n = 0;
    CHECK(mi355_load_match_pairs((gold + "/matchPairs.match").c_str(), &v, &n) == MI355_OK && n == 5918);
    const std::string t = tmp + "/san_" + std::to_string(tag);
    CHECK(mi355_write_match_pairs((t + ".match").c_str(), v, n) == MI355_OK);
    CHECK(mi355_write_match_pairs_txt((t + ".txt").c_str(), v, n) == MI355_OK);
    std::vector<int32_t> label(20);
    CHECK(mi355_select_connected(v, n, 20, &label[0]) == MI355_OK);
    std::vector<mi355_image_transform> T(20);
    CHECK(mi355_global_affine_align(v, n, 20, NULL, &T[0]) == MI355_OK);
    CHECK(std::fabs(T[1].m[5] + 100.6f) < 1.0f);                    // tran0.txt row 1: ty = -100.622
    CHECK(mi355_write_transforms((t + ".tran").c_str(), &T[0], 20) == MI355_OK);
    mi355_image_transform* back = NULL; int nb = 0;
    CHECK(mi355_load_tran0((t + ".tran").c_str(), &back, &nb) == MI355_OK && nb == 20);
    mi355_free(back);
    CHECK(mi355_load_transforms((gold + "/does_not_exist").c_str(), &back, &nb) == MI355_ERR_FAILED);
    // degenerate inputs: empty, single image, disconnected graph
    CHECK(mi355_select_connected(v, 0, 5, &label[0]) == MI355_OK);
    CHECK(mi355_global_affine_align(v, 0, 3, NULL, &T[0]) == MI355_OK);
    // results -> match pairs
    std::vector<mi355_pair_result> r(3);
    std::memset(&r[0], 0, sizeof(mi355_pair_result) * 3);
    r[0].i = 0; r[0].j = 1; r[0].n_in = 400; r[0].accepted = 1; r[2].i = 1; r[2].j = 2; r[2].n_in = 31; r[2].accepted = 1;
    mi355_match_point_pairs* mp = NULL; int nm = 0;
    CHECK(mi355_results_to_match_pairs(&r[0], 3, NULL, &mp, &nm) == MI355_OK && nm == 431);
    mi355_free(mp);
    // ResampleByOverlap: nearly identical axis-aligned images (the reference keeps them all: its segment intersections degenerate
    // for axis-aligned edges -- what matters here is memory safety; the decisions are pinned in tests/test_overlap.py), far apart
    // images, a skipped image, a degenerate matrix (zero area, NaN ratio)
    const int w[5] = {640, 640, 640, 640, 640}, h[5] = {480, 480, 480, 480, 480};
    float h9[45];
    for (int k = 0; k < 5; k++) { float* m = h9 + 9 * k; std::memset(m, 0, 36); m[0] = m[4] = m[8] = 1.0f; m[2] = 3.0f * k; }
    uint8_t keep[5];
    CHECK(mi355_resample_by_overlap(w, h, 5, h9, 0.7f, keep) == MI355_OK && keep[0] == 1 && keep[4] == 1);
    for (int k = 0; k < 5; k++) h9[9 * k + 2] = 1000.0f * k;
    h9[9 * 2 + 8] = 0.0f;
    CHECK(mi355_resample_by_overlap(w, h, 5, h9, 0.7f, keep) == MI355_OK && keep[1] == 1 && keep[2] == 1 && keep[3] == 1);
    std::memset(h9 + 9, 0, 36); h9[9 + 8] = 1.0f;                   // all corners collapse to one point: zero area, NaN ratio
    CHECK(mi355_resample_by_overlap(w, h, 5, h9, 0.7f, keep) == MI355_OK);
    int32_t pairs[64]; int np = 0;
    CHECK(mi355_pair_schedule(10, 182, 1, 3, pairs, 32, &np) == MI355_OK || np > 32);
    mi355_free(v);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string gold = argv[1], tmp = argv[2];
    if (work(gold, tmp, 0)) return 1;
    // the host-only entry points hold no shared state: four threads at once (ThreadSanitizer build runs the same program)
    int bad[4] = {0, 0, 0, 0};
    std::vector<std::thread> th;
    for (int t = 0; t < 4; t++) th.push_back(std::thread([&, t]() { bad[t] = work(gold, tmp, 1 + t); }));
    for (int t = 0; t < 4; t++) th[t].join();
    for (int t = 0; t < 4; t++) if (bad[t]) return 1;
    std::printf("SANITIZE_OK\n");
    return 0;
}
