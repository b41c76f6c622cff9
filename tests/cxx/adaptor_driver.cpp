// tests/cxx/adaptor_driver.cpp -- a C++ caller of libmi355mosaic.so THROUGH include/mi355_adaptor.h, the way a maintainer of the
// reference would call it (the reference's own signatures: Ransac2D, ImageProjectionTransform, MosaicImagesRefined,
// LaplacianPyramidBlending / MergeImagesRefined).  tests/test_gpu_cxx.py writes the inputs as raw binary files, runs this
// program on the GPU box and compares what it writes with the golden vectors / the Python-side results.
//
//   adaptor_driver <dir> ransac | warp | mosaic | blend | threads | surf | sift | select
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "mi355_adaptor.h"

using namespace mi355ref;

static std::vector<unsigned char> slurp(const std::string& p) {
    std::vector<unsigned char> v;
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", p.c_str()); std::exit(3); }
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (n && std::fread(&v[0], 1, (size_t)n, f) != (size_t)n) std::exit(3);
    std::fclose(f);
    return v;
}
static void spit(const std::string& p, const void* d, size_t n) {
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f || (n && std::fwrite(d, 1, n, f) != n)) std::exit(4);
    std::fclose(f);
}

// ransac.bin: int32 n_cases, then per case int32 n, uint32 seed, n x SfPoint p1, n x SfPoint p2
struct RCase { int n; unsigned seed; std::vector<SfPoint> p1, p2; };
static std::vector<RCase> read_ransac(const std::string& dir) {
    std::vector<unsigned char> raw = slurp(dir + "/ransac.bin");
    const unsigned char* p = &raw[0];
    int nc; std::memcpy(&nc, p, 4); p += 4;
    std::vector<RCase> cases(nc);
    for (int c = 0; c < nc; c++) {
        std::memcpy(&cases[c].n, p, 4); p += 4; std::memcpy(&cases[c].seed, p, 4); p += 4;
        cases[c].p1.resize(cases[c].n); cases[c].p2.resize(cases[c].n);
        std::memcpy(&cases[c].p1[0], p, 12 * (size_t)cases[c].n); p += 12 * (size_t)cases[c].n;
        std::memcpy(&cases[c].p2[0], p, 12 * (size_t)cases[c].n); p += 12 * (size_t)cases[c].n;
    }
    return cases;
}
// result record per case: int32 ok, int32 n_in, float H[9], n_in x int32 ids
static void run_case(const RCase& rc, std::vector<unsigned char>& out) {
    std::vector<SfPoint> in1, in2; float H[9];
    const bool ok = mi355::Ransac2D(rc.p1, rc.p2, in1, in2, H, 2.5f, 1000, rc.seed);
    const int head[2] = {ok ? 1 : 0, (int)in1.size()};
    const size_t o = out.size();
    out.resize(o + 8 + 36 + 4 * in1.size());
    std::memcpy(&out[o], head, 8); std::memcpy(&out[o + 8], H, 36);
    for (size_t i = 0; i < in1.size(); i++) std::memcpy(&out[o + 44 + 4 * i], &in1[i].id, 4);
}

// images.bin: int32 n, then per image int32 w, h, ws + ws*h bytes + float h9[9]
struct Img { int w, h, ws; std::vector<unsigned char> px; float h9[9]; };
static std::vector<Img> read_images(const std::string& path) {
    std::vector<unsigned char> raw = slurp(path);
    const unsigned char* p = &raw[0];
    int n; std::memcpy(&n, p, 4); p += 4;
    std::vector<Img> v(n);
    for (int k = 0; k < n; k++) {
        std::memcpy(&v[k].w, p, 4); std::memcpy(&v[k].h, p + 4, 4); std::memcpy(&v[k].ws, p + 8, 4); p += 12;
        v[k].px.assign(p, p + (size_t)v[k].ws * v[k].h); p += (size_t)v[k].ws * v[k].h;
        std::memcpy(v[k].h9, p, 36); p += 36;
    }
    return v;
}
static IplImage* to_ipl(const Img& im) {
    IplImage* q = cvCreateImage8U(im.w, im.h, 3);
    for (int y = 0; y < im.h; y++) std::memcpy(q->imageData + (size_t)y * q->widthStep, &im.px[(size_t)y * im.ws], (size_t)3 * im.w);
    return q;
}
static void write_ipl(const std::string& path, const IplImage* im) {
    std::vector<unsigned char> out(12 + (size_t)im->widthStep * im->height);
    const int hd[3] = {im->width, im->height, im->widthStep};
    std::memcpy(&out[0], hd, 12); std::memcpy(&out[12], im->imageData, (size_t)im->widthStep * im->height);
    spit(path, &out[0], out.size());
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: adaptor_driver <dir> ransac|warp|mosaic|blend|threads\n"); return 2; }
    const std::string dir = argv[1], mode = argv[2];
    if (!mi355::context()) { std::fprintf(stderr, "no context: %s\n", mi355_last_error(NULL)); return 5; }
    if (mode == "ransac") {
        std::vector<RCase> cases = read_ransac(dir);
        std::vector<unsigned char> out;
        for (size_t c = 0; c < cases.size(); c++) run_case(cases[c], out);
        spit(dir + "/ransac.out", out.empty() ? NULL : &out[0], out.size());
    } else if (mode == "threads") {
        // the reference calls the per-pair code from up to 8 worker threads at once (MosaicWithoutPos.cpp:5246-5292): 8 host
        // threads share the process-wide context, each runs every case (and a warp); all must reproduce the serial answers
        std::vector<RCase> cases = read_ransac(dir);
        std::vector<Img> imgs = read_images(dir + "/images.bin");
        std::vector<std::vector<unsigned char> > serial(cases.size());       // one record per case, computed before any thread starts
        for (size_t c = 0; c < cases.size(); c++) run_case(cases[c], serial[c]);
        BitmapImage src; src.imageData = &imgs[0].px[0]; src.width = imgs[0].w; src.height = imgs[0].h; src.widthStep = imgs[0].ws; src.nChannels = 3;
        BitmapImage* ref = NULL;
        if (mi355::ImageProjectionTransform(&src, ref, imgs[0].h9) != 0) return 6;
        const int T = 8;
        std::vector<int> bad(T, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.push_back(std::thread([&, t]() {
            for (int rep = 0; rep < 3; rep++) {
                for (size_t c = 0; c < cases.size(); c++) {
                    const size_t k = (c + 5 * t) % cases.size();              // every thread walks the cases in its own order
                    std::vector<unsigned char> mine;
                    run_case(cases[k], mine);
                    if (mine != serial[k]) bad[t]++;
                }
                BitmapImage* r = NULL;
                if (mi355::ImageProjectionTransform(&src, r, imgs[0].h9) != 0 || r->width != ref->width || r->height != ref->height ||
                    std::memcmp(r->imageData, ref->imageData, (size_t)r->widthStep * r->height) != 0) bad[t]++;
                ReleaseBitmap8U(r);                                          // ImageIO.cpp:78-93, as the reference's callers do
            }
        }));
        for (int t = 0; t < T; t++) th[t].join();
        int nbad = 0;
        for (int t = 0; t < T; t++) nbad += bad[t];
        std::vector<unsigned char> all;
        for (size_t c = 0; c < cases.size(); c++) all.insert(all.end(), serial[c].begin(), serial[c].end());
        spit(dir + "/ransac.out", all.empty() ? NULL : &all[0], all.size());
        std::printf("THREADS %s\n", nbad ? "MISMATCH" : "OK");
        return nbad ? 7 : 0;
    } else if (mode == "warp") {
        std::vector<Img> imgs = read_images(dir + "/images.bin");
        BitmapImage src; src.imageData = &imgs[0].px[0]; src.width = imgs[0].w; src.height = imgs[0].h; src.widthStep = imgs[0].ws; src.nChannels = 3;
        BitmapImage* res = NULL;
        if (mi355::ImageProjectionTransform(&src, res, imgs[0].h9) != 0) return 6;
        std::vector<unsigned char> out(12 + (size_t)res->widthStep * res->height);
        const int hd[3] = {res->width, res->height, res->widthStep};
        std::memcpy(&out[0], hd, 12); std::memcpy(&out[12], res->imageData, (size_t)res->widthStep * res->height);
        spit(dir + "/warp.out", &out[0], out.size());
        ReleaseBitmap8U(res);
    } else if (mode == "mosaic" || mode == "blend") {
        std::vector<Img> imgs = read_images(dir + "/images.bin");
        const int n = (int)imgs.size();
        std::vector<ImagePoseInfo> poses(n); std::vector<ImageTransform> tr(n);
        for (int k = 0; k < n; k++) { poses[k].pImg = to_ipl(imgs[k]); std::memcpy(tr[k].h.m, imgs[k].h9, 36); tr[k].fixed = (k == 0); }
        IplImage* result = NULL;
        if (mode == "mosaic") {
            const int rc = mi355::MosaicImagesRefined(&poses[0], n, &tr[0], result);
            if (rc != 0 || !result) return 6;
            for (int k = 0; k < n; k++) if (!poses[k].pImg) return 8;       // MosaicImagesRefined leaves the inputs alone
            for (int k = 0; k < n; k++) cvReleaseImage(&poses[k].pImg);
        } else {
            const int rc = mi355::MergeImagesRefined(&poses[0], n, &tr[0], 1.0f, result);   // m_scale = 1
            if (rc != 0 || !result) return 6;
            for (int k = 0; k < n; k++) if (poses[k].pImg) return 8;        // the inputs were consumed (MosaicImage.cpp:2464-2467, MWP.cpp:2182-2185)
        }
        write_ipl(dir + "/" + mode + ".out", result);
        cvReleaseImage(&result);
    } else if (mode == "surf") {
        // GetMatchedPairsOneToAllSurf through the adaptor: MatchPointPairs out (40-byte records) + nSuccess
        std::vector<Img> imgs = read_images(dir + "/images.bin");
        const int n = (int)imgs.size();
        std::vector<ImagePoseInfo> poses(n);
        for (int k = 0; k < n; k++) { poses[k].pImg = to_ipl(imgs[k]); poses[k].fixed = (k == 0); }
        std::vector<MatchPointPairs> v; int nSuccess = 0;
        if (mi355::GetMatchedPairsOneToAllSurf(&poses[0], n, v, nSuccess, 50, 0.5f, 200, 2.5f, 4u) != 0) return 6;
        std::vector<unsigned char> out(8 + v.size() * sizeof(MatchPointPairs));
        const int hd[2] = {nSuccess, (int)v.size()};
        std::memcpy(&out[0], hd, 8);
        if (!v.empty()) std::memcpy(&out[8], &v[0], v.size() * sizeof(MatchPointPairs));
        spit(dir + "/surf.out", &out[0], out.size());
        for (int k = 0; k < n; k++) cvReleaseImage(&poses[k].pImg);
    } else if (mode == "sift") {
        // GetMatchedPairsOneToAllSIFT_MultiThread through the adaptor: extraction + window matching in ONE call
        std::vector<Img> imgs = read_images(dir + "/images.bin");
        const int n = (int)imgs.size();
        std::vector<ImagePoseInfo> poses(n);
        for (int k = 0; k < n; k++) { poses[k].pImg = to_ipl(imgs[k]); poses[k].fixed = (k == 0); }
        std::vector<MatchPointPairs> v; int nSuccess = -1;
        if (mi355::GetMatchedPairsOneToAllSIFT_MultiThread(&poses[0], n, v, nSuccess, 2.5f, 9u) != 0) return 6;
        std::vector<unsigned char> out(8 + v.size() * sizeof(MatchPointPairs));
        const int hd[2] = {nSuccess, (int)v.size()};
        std::memcpy(&out[0], hd, 8);
        if (!v.empty()) std::memcpy(&out[8], &v[0], v.size() * sizeof(MatchPointPairs));
        spit(dir + "/sift.out", &out[0], out.size());
        for (int k = 0; k < n; k++) cvReleaseImage(&poses[k].pImg);
    } else if (mode == "select") {
        // SelectMatchPairs with element types shaped like cv::DMatch / cv::KeyPoint (OpenCV 2.4.0 features2d.hpp: KeyPoint carries
        // Point2f pt, DMatch {queryIdx, trainIdx, imgIdx, distance}) -- the spelling of the reference's call, MosaicWithoutPos.cpp:5146-5153
        // -- and once more with the C-ABI PODs; both must give the same lists.
        // select.bin: int32 n_cases; per case int32 {K, nMatch, width, height}, K x mi355_dmatch, K x float2 kp1, K x float2 kp2
        struct Point2f { float x, y; };
        struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
        struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
        std::vector<unsigned char> raw = slurp(dir + "/select.bin");
        const unsigned char* p = &raw[0];
        int nc; std::memcpy(&nc, p, 4); p += 4;
        std::vector<unsigned char> out;
        for (int c = 0; c < nc; c++) {
            int hd[4]; std::memcpy(hd, p, 16); p += 16;
            const int K = hd[0];
            std::vector<DMatch> m(K); std::vector<KeyPoint> k1(K), k2(K);
            std::vector<mi355_dmatch> pm(K); std::vector<mi355_keypoint> q1(K), q2(K);
            if (K) std::memcpy(&m[0], p, (size_t)16 * K);
            if (K) std::memcpy(&pm[0], p, (size_t)16 * K);
            p += (size_t)16 * K;
            for (int i = 0; i < K; i++) { std::memcpy(&k1[i].pt, p + (size_t)8 * i, 8); std::memset(&q1[i], 0, sizeof q1[i]); q1[i].x = k1[i].pt.x; q1[i].y = k1[i].pt.y; }
            p += (size_t)8 * K;
            for (int i = 0; i < K; i++) { std::memcpy(&k2[i].pt, p + (size_t)8 * i, 8); std::memset(&q2[i], 0, sizeof q2[i]); q2[i].x = k2[i].pt.x; q2[i].y = k2[i].pt.y; }
            p += (size_t)8 * K;
            std::vector<SfPoint> v1, v2, w1, w2;
            if (mi355::SelectMatchPairs(m, k1, k2, hd[1], hd[2], hd[3], 3, 3, v1, v2) != 0) return 6;
            if (mi355::SelectMatchPairs(pm, q1, q2, hd[1], hd[2], hd[3], 3, 3, w1, w2) != 0) return 6;
            if (v1.size() != w1.size() || v2.size() != w2.size() || v1.size() != v2.size()) return 7;
            if (!v1.empty() && (std::memcmp(&v1[0], &w1[0], v1.size() * sizeof(SfPoint)) || std::memcmp(&v2[0], &w2[0], v2.size() * sizeof(SfPoint)))) return 7;
            const int n = (int)v1.size();
            const size_t o = out.size();
            out.resize(o + 4 + (size_t)24 * n);
            std::memcpy(&out[o], &n, 4);
            if (n) { std::memcpy(&out[o + 4], &v1[0], (size_t)12 * n); std::memcpy(&out[o + 4 + (size_t)12 * n], &v2[0], (size_t)12 * n); }
        }
        spit(dir + "/select.out", out.empty() ? NULL : &out[0], out.size());
    } else return 2;
    std::printf("DONE %s\n", mode.c_str());
    return 0;
}
