/* oracle/oracle_ransac.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of Ransac2D<SfPoint> (mosaicimage.h:1729-2035) with transformType = PROJECT_MODEL
 * (:1737), and of the glibc rand()/srand() stream it draws from (the reference calls
 * srand((unsigned)time(0)) at :1777; tests pin time() to a seed).
 */
#include "oracle.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

/* glibc stdlib/random_r.c, TYPE_3 (degree 31, separation 3) additive feedback generator:
 * seed word0 (0 -> 1); word[i] = 16807*word[i-1] mod 2^31-1 by Schrage's hi/lo split; 310 outputs
 * discarded; each output (f += b, both advance) >> 1. */
void orc_srand(orc_glibc_rand* g, unsigned seed)
{
    if (seed == 0) seed = 1;
    g->r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = g->r[i - 1] / 127773, lo = g->r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        g->r[i] = (int32_t)word;
    }
    g->f = 3; g->b = 0;
    for (int i = 0; i < 310; i++) (void)orc_rand(g);
}

int orc_rand(orc_glibc_rand* g)
{
    uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
    g->r[g->f] = (int32_t)v;
    int out = (int)(v >> 1);
    if (++g->f >= 31) g->f = 0;
    if (++g->b >= 31) g->b = 0;
    return out;
}

static void apply2(const float* M, float x, float y, float* X, float* Y)   /* matrix.h:1027-1036 */
{
    float inv = 1.0f / (M[6] * x + M[7] * y + 1.0f);
    *X = (M[0] * x + M[1] * y + M[2]) * inv;
    *Y = (M[3] * x + M[4] * y + M[5]) * inv;
}
static void apply3(const float* M, float x, float y, float* X, float* Y)   /* matrix.h:1003-1013 */
{
    *X = (M[0] * x + M[1] * y + M[2]) / (M[6] * x + M[7] * y + 1.0f);
    *Y = (M[3] * x + M[4] * y + M[5]) / (M[6] * x + M[7] * y + 1.0f);
}

int orc_ransac2d(const orc_sfpoint* p1, const orc_sfpoint* p2, int n, float dist, int sample_times,
                 unsigned seed, orc_sfpoint* in1, orc_sfpoint* in2, int* n_in, float H[9])
{
    *n_in = 0;
    if (n <= 0) return 0;                                   /* :1739-1744 */
    float d2 = dist * dist;                                 /* :1757 */
    if (n < 4) return 0;                                    /* :1760-1761 */
    float invn = 1.0f / (float)n;                           /* :1763 */
    const int maxTimes = 5000;                              /* :1765 */
    if (sample_times > maxTimes) sample_times = maxTimes;
    if (sample_times < 1) return 0;
    float (*hyp)[9] = (float (*)[9])calloc((size_t)sample_times, sizeof(float[9]));
    orc_glibc_rand g; orc_srand(&g, seed);                  /* :1777 */
    int maxSupport = 0, maxIdx = 0, real = 0;
    for (int t = 0; t < sample_times;) {                    /* :1785 */
        real++;
        if (real >= maxTimes) break;                        /* :1789-1792 */
        int s[4];
        do { for (int i = 0; i < 4; i++) s[i] = orc_rand(&g) % n; }      /* :1801-1813 */
        while (s[0] == s[1] || s[0] == s[2] || s[0] == s[3] || s[1] == s[2] || s[1] == s[3] || s[2] == s[3]);
        orc_sfpoint a[4], b[4];
        for (int i = 0; i < 4; i++) { a[i] = p1[s[i]]; b[i] = p2[s[i]]; }
        float h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        orc_solve_homography(a, b, 4, h);                   /* :1863 */
        if (h[8] > 5.0f) continue;                          /* :1864-1867: slot t not consumed */
        else if (h[8] < 5.0f && h[8] > 0.01f) {             /* :1868-1876 */
            float fine[9];
            orc_nlls_projection2(a, b, 4, fine, h, 1e-10f);
            memcpy(h, fine, sizeof(fine));
        }
        memcpy(hyp[t], h, sizeof(h));                       /* :1887 */
        int support = 0;
        for (int i = 0; i < n; i++) {                       /* :1890-1904 */
            float bx, by; apply2(hyp[t], p2[i].x, p2[i].y, &bx, &by);
            float dx = bx - p1[i].x, dy = by - p1[i].y;
            float dd = dx * dx + dy * dy;
            if (dd < d2) support++;
        }
        if (support > maxSupport) {                         /* :1905-1917 */
            maxSupport = support; maxIdx = t;
            if ((float)maxSupport * invn > 0.99f) break;
        }
        t++;
    }
    int cnt = 0;
    for (int i = 0; i < n; i++) {                           /* :1922-1944, true-division form */
        float bx, by; apply3(hyp[maxIdx], p2[i].x, p2[i].y, &bx, &by);
        float dx = bx - p1[i].x, dy = by - p1[i].y;
        float dd = dx * dx + dy * dy;
        if (dd < d2) { in1[cnt] = p1[i]; in2[cnt] = p2[i]; cnt++; }
    }
    *n_in = cnt;
    int ok = 1;
    if (cnt > 0) { if (!orc_solve_homography(in1, in2, cnt, H)) ok = 0; }   /* :1953-1961 */
    else ok = 0;
    if (ok) {                                               /* :1977-1986: NLLS from the winning hypothesis */
        float m[9];
        orc_nlls_projection2(in1, in2, cnt, m, hyp[maxIdx], 1e-10f);
        memcpy(H, m, sizeof(m));
    }
    free(hyp);
    if (cnt < 4) return 0;                                  /* :2024-2032 */
    return 1;
}
