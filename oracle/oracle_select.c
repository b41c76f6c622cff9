/* oracle/oracle_select.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * - orc_select_match_pairs : SelectMatchPairs (grid form), MosaicWithoutPos.cpp:4977-5028
 * - orc_bf_match           : exact brute-force 1-NN/2-NN in squared L2.  The reference calls
 *                            cv::FlannBasedMatcher().match (MosaicWithoutPos.cpp:5108-5110), an
 *                            APPROXIMATE 1-NN whose arithmetic is in OpenCV 2.4.0 (absent): the
 *                            exact answer is what this project defines (north_star) -- parity unpinned.
 * - orc_sort_matches       : stands for std::sort(matches) (MosaicWithoutPos.cpp:5111), whose order
 *                            among equal distances is unspecified; total order (dist2, queryIdx).
 * - orc_match_pair         : the j-loop body MosaicWithoutPos.cpp:5108-5221.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

int orc_select_match_pairs(const int32_t* matches, int n_matches, const float* kp1xy, const float* kp2xy,
                           int nMatch, int width, int height, int gridX, int gridY,
                           orc_sfpoint* out1, orc_sfpoint* out2, int* n_out)
{
    int nGrids = gridX * gridY;
    int perGrid = (int)((float)nMatch / (float)nGrids);      /* :4990 */
    int* label = (int*)calloc((size_t)nGrids, sizeof(int));
    int stepX = width / gridX, stepY = height / gridY;       /* :4994-4995 */
    int cnt = 0;
    for (int n = 0; n < n_matches; n++) {
        int q = matches[2 * n], t = matches[2 * n + 1];
        float x = kp1xy[2 * q], y = kp1xy[2 * q + 1];
        int nX = (int)(x / (float)stepX);                    /* :5008 */
        int nY = (int)(y / (float)stepY);
        int cell = gridX * nY + nX;                          /* :5011 -- aliases like the reference when nX==gridX */
        if (cell < 0) cell = 0;                              /* the reference would index out of bounds here: */
        if (cell >= nGrids) cell = nGrids - 1;               /* clamped (documented divergence, unreachable for SIFT keypoints) */
        if (label[cell] >= perGrid) continue;
        out1[cnt].x = x; out1[cnt].y = y; out1[cnt].id = q;
        out2[cnt].x = kp2xy[2 * t]; out2[cnt].y = kp2xy[2 * t + 1]; out2[cnt].id = t;
        cnt++;
        label[cell]++;
    }
    free(label);
    *n_out = cnt;
    return 0;
}

void orc_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2,
                  int32_t* nn1_idx, int32_t* nn1_d2, int32_t* nn2_d2)
{
    for (int i = 0; i < n1; i++) {
        int32_t best = 0x7fffffff, second = 0x7fffffff, bi = -1;
        const uint8_t* a = d1 + (size_t)i * 128;
        for (int j = 0; j < n2; j++) {
            const uint8_t* b = d2 + (size_t)j * 128;
            int32_t s = 0;
            for (int k = 0; k < 128; k++) { int32_t d = (int32_t)a[k] - (int32_t)b[k]; s += d * d; }
            if (s < best) { second = best; best = s; bi = j; }       /* ties keep the lowest train index */
            else if (s < second) second = s;
        }
        nn1_idx[i] = bi; nn1_d2[i] = best; if (nn2_d2) nn2_d2[i] = second;
    }
}

typedef struct { int32_t d2, q, t; } srt;
static int cmp_srt(const void* a, const void* b)
{
    const srt* x = (const srt*)a; const srt* y = (const srt*)b;
    if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
    return x->q < y->q ? -1 : (x->q > y->q ? 1 : 0);
}
void orc_sort_matches(const int32_t* nn1_idx, const int32_t* nn1_d2, int n1, int32_t* matches_out)
{
    srt* v = (srt*)malloc(sizeof(srt) * (size_t)(n1 > 0 ? n1 : 1));
    int m = 0;
    for (int i = 0; i < n1; i++) if (nn1_idx[i] >= 0) { v[m].d2 = nn1_d2[i]; v[m].q = i; v[m].t = nn1_idx[i]; m++; }
    qsort(v, (size_t)m, sizeof(srt), cmp_srt);
    for (int i = 0; i < m; i++) { matches_out[2 * i] = v[i].q; matches_out[2 * i + 1] = v[i].t; }
    free(v);
}

int orc_match_pair(const float* kp1xy, const uint8_t* d1, int n1, const float* kp2xy, const uint8_t* d2, int n2,
                   int width, int height, float ransac_dist, unsigned seed,
                   orc_sfpoint* in1, orc_sfpoint* in2, float H[9], int* n_selected)
{
    return orc_match_pair_ratio(kp1xy, d1, n1, kp2xy, d2, n2, width, height, ransac_dist, seed, 0.0f, in1, in2, H, n_selected);
}

/* ratio > 0: Lowe's ratio test on squared distances (d1 < ratio^2 * d2, both as float) removes matches BEFORE the grid
 * walk (north_star option; not in the reference, which has no ratio test -- SURVEY 0.1).  nMatch is still computed from the
 * number of 1-NN matches M like MosaicWithoutPos.cpp:5146-5147. */
int orc_match_pair_ratio(const float* kp1xy, const uint8_t* d1, int n1, const float* kp2xy, const uint8_t* d2, int n2,
                         int width, int height, float ransac_dist, unsigned seed, float ratio,
                         orc_sfpoint* in1, orc_sfpoint* in2, float H[9], int* n_selected)
{
    if (n_selected) *n_selected = 0;
    for (int i = 0; i < 9; i++) H[i] = 0.0f;
    if (n1 <= 0 || n2 <= 0) return 0;
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1);
    int32_t* dd  = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1);
    int32_t* m   = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1 * 2);
    int32_t* d2nd = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1);
    orc_bf_match(d1, n1, d2, n2, idx, dd, d2nd);
    orc_sort_matches(idx, dd, n1, m);
    int nm = n1;
    if (ratio > 0.0f) {                              /* drop the matches failing the ratio test, order kept */
        const float r2 = ratio * ratio;
        int k = 0;
        for (int i = 0; i < n1; i++) {
            int q = m[2 * i];
            if ((float)dd[q] < r2 * (float)d2nd[q]) { m[2 * k] = m[2 * i]; m[2 * k + 1] = m[2 * i + 1]; k++; }
        }
        nm = k;
    }
    free(d2nd);
    double lim = 0.3 * (double)n1;                                   /* :5146-5147  Min(400, 0.3*M) -> int */
    int nMatch = (int)(400.0 < lim ? 400.0 : lim);
    orc_sfpoint* s1 = (orc_sfpoint*)malloc(sizeof(orc_sfpoint) * (size_t)n1);
    orc_sfpoint* s2 = (orc_sfpoint*)malloc(sizeof(orc_sfpoint) * (size_t)n1);
    int ns = 0;
    orc_select_match_pairs(m, nm, kp1xy, kp2xy, nMatch, width, height, 3, 3, s1, s2, &ns);   /* :5149-5153 */
    if (n_selected) *n_selected = ns;
    int nin = 0;
    orc_ransac2d(s1, s2, ns, ransac_dist, 1000, seed, in1, in2, &nin, H);                     /* :5169 */
    free(idx); free(dd); free(m); free(s1); free(s2);
    return nin > 30 ? nin : 0;                                       /* :5049, :5201 */
}
