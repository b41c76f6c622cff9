// micro-benchmark of the SIFT stage-A kernels at the 12 MP octave-0 size (8000x6000): includes the product TU
#include "../imagemosaicing_amd/csrc/sift.hip"
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int smode = getenv("MI355_BLUR_STREAM") ? atoi(getenv("MI355_BLUR_STREAM")) : 1;
    const int W = getenv("KW") ? atoi(getenv("KW")) : 8000, H = getenv("KH") ? atoi(getenv("KH")) : 6000;
    const size_t px = (size_t)W * H;
    float* lv[7];
    const int NB = getenv("KNB") ? atoi(getenv("KNB")) : 1;
    for (int i = 0; i < 7; i++) CK(hipMalloc(&lv[i], px * 4 * NB));
    std::vector<float> h(px);
    unsigned s = 12345;
    for (size_t i = 0; i < px; i++) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 24); }
    for (int f = 0; f < NB; f++) CK(hipMemcpy(lv[0] + (size_t)f * px, h.data(), px * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double sigma = 1.6, k = std::pow(2.0, 1.0 / 3);
    for (int i = 1; i < 6; i++) {
        const double sp = std::pow(k, (double)(i - 1)) * sigma, stt = sp * k;
        BlurArgs a; memset(&a, 0, sizeof(a));
        const int R = gauss_kernel_host(std::sqrt(stt * stt - sp * sp), a.k);
        a.src = lv[i - 1]; a.dst = lv[i]; a.w = W; a.h = H; a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
        a.nb = NB; a.fstride = px;
        launch_blur<false>(st, R, a, smode);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 10; it++) launch_blur<false>(st, R, a, smode);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        CK(hipMemcpy(h.data(), lv[i], px * 4, hipMemcpyDeviceToHost));
        unsigned long long ck = 1469598103934665603ull;
        for (size_t q = 0; q < px; q++) { unsigned u; memcpy(&u, &h[q], 4); ck = (ck ^ u) * 1099511628211ull; }
        printf("blur R=%2d: %.1f us/frame  %.0f GB/s algorithmic  checksum %016llx\n", R, ms * 1e3 / NB, NB * px * 8.0 / ms / 1e6, ck);
    }
    {   // base level: u8 BGR frame (W/2 x H/2) -> gray -> 2x up-sampling -> blur
        const int fw = W / 2, fh = H / 2, ws = fw * 3;
        uint8_t* bgr; CK(hipMalloc(&bgr, (size_t)ws * fh));
        std::vector<uint8_t> hb((size_t)ws * fh);
        for (size_t i = 0; i < hb.size(); i++) { s = s * 1664525u + 1013904223u; hb[i] = (uint8_t)(s >> 24); }
        CK(hipMemcpy(bgr, hb.data(), hb.size(), hipMemcpyHostToDevice));
        const int gp = (fw + 8 + 15) & ~15;
        uint8_t* gray; CK(hipMalloc(&gray, (size_t)gp * fh + 64));
        BlurArgs a; memset(&a, 0, sizeof(a));
        const int R = gauss_kernel_host(std::sqrt(1.6 * 1.6 - 1.0), a.k);
        a.bgr = bgr; a.bgr_ws = ws; a.dst = lv[6]; a.w = W; a.h = H; a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
        for (int mode = 0; mode < 2; mode++) {
            for (int it = 0; it < 11; it++) {
                if (it == 1) CK(hipEventRecord(e0, st));
                if (mode && base_streams(a, R, 1)) { launch_gray_pad(st, a.bgr, a.bgr_ws, a.w / 2, a.h / 2, gray, gp); launch_base_stream(st, a, gray, gp, 0); } else launch_blur<true>(st, R, a, 0);
            }
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            CK(hipMemcpy(h.data(), lv[6], px * 4, hipMemcpyDeviceToHost));
            unsigned long long ck = 1469598103934665603ull;
            for (size_t q = 0; q < px; q++) { unsigned u; memcpy(&u, &h[q], 4); ck = (ck ^ u) * 1099511628211ull; }
            printf("base R=%d mode %d: %.1f us  checksum %016llx\n", R, mode, ms * 1e3, ck);
        }
    }
    if (argc > 1) return 0;
    OctaveDev oc; for (int i = 0; i < 6; i++) oc.lv[i] = lv[i]; oc.w = W; oc.h = H;
    float* cube; CK(hipMalloc(&cube, 1 << 20));
    unsigned long long* cand; unsigned* cnt; CK(hipMalloc(&cand, 64 << 20)); CK(hipMalloc(&cnt, 8192)); CK(hipMemset(cnt, 0, 8192));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(cnt, 0, 8192));
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 5; it++) hipLaunchKernelGGL(extrema_kernel, dim3(((W + EW - 1) / EW) * ((H + EH - 1) / EH)), dim3(256), 0, st, oc, 0, cand, cnt, (8u << 20) / 64, cnt + 2047, BatchStride{0, 0, 0, 0, 0, 0}, cube, 0u);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        unsigned c; CK(hipMemcpy(&c, cnt, 4, hipMemcpyDeviceToHost));
        printf("extrema: %.1f us  %.0f GB/s algorithmic (6 levels)  cands/launch %u\n", ms * 1e3, px * 24.0 / ms / 1e6, c / 5);
    }
    return 0;
}
