// csrc/overlap.cpp -- ResampleByOverlap (MosaicImage.cpp:2069-2201): which images LaplacianPyramidBlending keeps.
//
// Host code: O(N^2) quadrilateral geometry on 8 floats per image, microseconds -- there is nothing for the GPU here; the
// result (vecAbandonInd) is the keep[] argument of mi355_chips_and_masks / mi355_mosaic_blended.  The float / double mix of
// the reference is kept operation by operation (float sqrt/acos/sin/cos/atan overloads of <cmath>, the double promotion in
// `0.5*L1*L2*sin(theta)` and in the comparisons against 1e-4 / 0.2 / 0.000001), so the decision is the reference's on the
// same libm; pinned against the reference's own code through oracle/_ref (tests/test_overlap.py) and tests/golden.
//   AreaOfQuadrangle                 MosaicImage.cpp:1884-1932      IsPointOnLineSegmentOfTwoPoints  :1935-1948
//   GetAllIntersecPoints             :1951-1996                     IsPointInQuadrangle              :1999-2027
//   GetPointsInOverlapRegion         :2030-2067                     ResampleByOverlap                :2069-2201
//   LineOf2Points1 ImageMath.cpp:88-103, ABCToPolar :144-176, AngleofPoint imageMath.h:26-88, AngleofPoint360 ImageMath.cpp:9-54,
//   IntersectionPointOf2PolarLines ImageMath.cpp:399-413, DistanceOfTwoPoints mvMath.h:186-192, pi = 3.1415926f Bitmap.h:54
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>
#include "../../include/mi355_mosaic.h"

namespace {

const float kPi = 3.1415926f;                      // Bitmap.h:54
struct Pt { float x, y; };

inline float dist2pts(float x1, float y1, float x2, float y2) {                 // mvMath.h:186-192 with T3 = float
    return std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
}

float quad_area(const Pt c[4]) {                                                 // :1884-1932
    const float dax = c[0].x - c[2].x, day = c[0].y - c[2].y;
    const float dbx = c[1].x - c[3].x, dby = c[1].y - c[3].y;
    const float diag_a = std::sqrt(dax * dax + day * day);
    const float diag_b = std::sqrt(dbx * dbx + dby * dby);
    const float e01 = dist2pts(c[0].x, c[0].y, c[1].x, c[1].y);
    const float e02 = dist2pts(c[0].x, c[0].y, c[2].x, c[2].y);
    const float e12 = dist2pts(c[1].x, c[1].y, c[2].x, c[2].y);
    const float e03 = dist2pts(c[0].x, c[0].y, c[3].x, c[3].y);
    const float e23 = dist2pts(c[2].x, c[2].y, c[3].x, c[3].y);
    const float half_t1 = (e01 + e02 + e12) * 0.5f;
    const float tri1 = std::sqrt(half_t1 * (half_t1 - e01) * (half_t1 - e02) * (half_t1 - e12));
    const float half_t2 = (e02 + e03 + e23) * 0.5f;
    const float tri2 = std::sqrt(half_t2 * (half_t2 - e02) * (half_t2 - e03) * (half_t2 - e23));
    if (((double)std::fabs(tri1) < 1e-4) && ((double)std::fabs(tri2) < 1e-4)) return 0.0f;
    if (diag_a * diag_b > 0) {
        const float cos_between = (dax * dbx + day * dby) / (diag_a * diag_b);
        float between = std::acos(cos_between);
        if (between < 0) between = between + 3.1415926f;
        return (float)(0.5 * (double)diag_a * (double)diag_b * (double)std::sin(between));   // double product of float factors, :1926
    }
    return 0.0f;
}

bool on_segment(Pt pt, Pt a, Pt b) {                                             // :1935-1948
    const float len_ab = dist2pts(a.x, a.y, b.x, b.y);
    const float to_a = dist2pts(pt.x, pt.y, a.x, a.y);
    const float to_b = dist2pts(pt.x, pt.y, b.x, b.y);
    return (double)std::fabs(to_a + to_b - len_ab) < 1e-4;
}

void line_of_2_points(float& a, float& b, float& c, float x1, float y1, float x2, float y2) {   // ImageMath.cpp:88-103
    if (std::fabs((double)(x1 - x2)) < 0.000001) { a = 1.0f; b = 0; c = -x1; }
    else { a = (y1 - y2) / (x1 - x2); b = -1.0f; c = y1 - a * x1; }
}

float angle_of_point(float x, float y) {                                         // imageMath.h:26-88 (T = float)
    float angle = 0.0f;
    if (x >= 0) {
        if (y >= 0) { if (x != 0) angle = (float)std::atan(y / x); else angle = (y != 0) ? kPi / 2 : 0.0f; }
        else { if (x != 0) angle = (float)std::atan(y / x); else angle = -kPi / 2; }
    } else {
        if (y >= 0) angle = kPi + (float)std::atan(y / x);                       // x != 0 here
        else angle = (float)std::atan(y / x) - kPi;
    }
    return angle;
}

void abc_to_polar(float a, float b, float c, float& rho, float& theta) {         // ImageMath.cpp:144-176
    float a1 = 0.0f, xc = 0.0f, yc = 0.0f;
    rho = std::fabs(a * 0 + b * 0 + c) / std::sqrt(a * a + b * b);
    if (b == 0) { yc = 0; xc = -c / a; }
    if (a == 0) { xc = 0; yc = -c / b; }
    if ((a != 0) && (b != 0)) { a1 = -1.0f / a; xc = -c / (a - a1); yc = a1 * xc; }
    theta = angle_of_point(xc, yc);
    if (theta < 0) theta = theta + 2 * kPi;
}

Pt polar_intersection(float rho1, float th1, float rho2, float th2) {            // ImageMath.cpp:399-413
    const float det = std::sin(th2) * std::cos(th1) - std::sin(th1) * std::cos(th2);
    Pt p;
    p.x = (rho1 * std::sin(th2) - rho2 * std::sin(th1)) / det;
    p.y = (-rho1 * std::cos(th2) + rho2 * std::cos(th1)) / det;
    return p;
}

void all_intersections(const Pt c1[4], const Pt c2[4], std::vector<Pt>& out) {   // :1951-1996
    for (int n = 0; n < 4; n++) { out.push_back(c1[n]); out.push_back(c2[n]); }
    const int from[4] = {0, 1, 2, 3}, to[4] = {1, 2, 3, 0};
    for (int ea = 0; ea < 4; ea++) {
        float la, lb, lc, ra, ta;
        line_of_2_points(la, lb, lc, c1[from[ea]].x, c1[from[ea]].y, c1[to[ea]].x, c1[to[ea]].y);
        abc_to_polar(la, lb, lc, ra, ta);
        for (int eb = 0; eb < 4; eb++) {
            float ma, mb, mc, rb, tb;
            line_of_2_points(ma, mb, mc, c2[from[eb]].x, c2[from[eb]].y, c2[to[eb]].x, c2[to[eb]].y);
            abc_to_polar(ma, mb, mc, rb, tb);
            const Pt p = polar_intersection(ra, ta, rb, tb);
            if (on_segment(p, c1[from[ea]], c1[to[ea]]) && on_segment(p, c2[from[eb]], c2[to[eb]])) out.push_back(p);
        }
    }
}

bool in_quad(Pt p, const Pt c[4]) {                                              // :1999-2027
    const int i1[4] = {0, 1, 2, 3}, i2[4] = {1, 2, 3, 0};
    const float area4 = quad_area(c);
    float acc = 0;
    for (int n = 0; n < 4; n++) {
        const Pt t[4] = {c[i1[n]], c[i2[n]], p, p};
        acc += quad_area(t);
    }
    return (double)std::fabs(area4 - acc) < 0.2;
}

float angle_360(float x, float y, float prev) {                                  // ImageMath.cpp:9-54; x < 0 always has x != 0
    float a = prev;
    if (x >= 0) {
        if (y >= 0) a = (x != 0) ? (float)std::atan(y / x) : kPi / 2;
        else a = (x != 0) ? 2 * kPi + (float)std::atan(y / x) : 3 * kPi / 2;
    } else {
        a = kPi + (float)std::atan(y / x);
    }
    return a;
}

struct Ang { float dist; int seq; bool operator<(const Ang& r) const { return dist < r.dist; } };   // pool::Distance, matrix.h:14-23

void points_in_overlap(const Pt c1[4], const Pt c2[4], const std::vector<Pt>& cand, std::vector<Pt>& out) {   // :2030-2067
    std::vector<Pt> v;
    float cx = 0, cy = 0;
    for (size_t i = 0; i < cand.size(); i++)
        if (in_quad(cand[i], c1) && in_quad(cand[i], c2)) { v.push_back(cand[i]); cx += cand[i].x; cy += cand[i].y; }
    cx /= v.size(); cy /= v.size();                                              // float / size_t -> float division (0/0 = NaN when empty, unused)
    std::vector<Ang> ang;
    for (size_t i = 0; i < v.size(); i++) { Ang a; a.seq = (int)i; a.dist = angle_360(v[i].x - cx, v[i].y - cy, 0.0f); ang.push_back(a); }
    std::sort(ang.begin(), ang.end());
    for (size_t i = 0; i < v.size(); i++) out.push_back(v[ang[i].seq]);
}

void image_quad(const float* m, int w, int h, Pt q[4]) {                         // :2096-2114 (two true divisions)
    const float cx[4] = {0.0f, (float)(w - 1), (float)(w - 1), 0.0f};
    const float cy[4] = {0.0f, 0.0f, (float)(h - 1), (float)(h - 1)};
    for (int i = 0; i < 4; i++) {
        q[i].x = (cx[i] * m[0] + cy[i] * m[1] + m[2]) / (cx[i] * m[6] + cy[i] * m[7] + m[8]);
        q[i].y = (cx[i] * m[3] + cy[i] * m[4] + m[5]) / (cx[i] * m[6] + cy[i] * m[7] + m[8]);
    }
}

}  // namespace

// keep[k] = vecAbandonInd[k] of the reference: 1 kept, 0 dropped.  Image 0 and image n-1 are always kept (:2082, :2198); an image
// with m[8] == 0 keeps its initial 1 (it is skipped later through m[8] itself).
extern "C" int mi355_resample_by_overlap(const int* w, const int* h, int n, const float* h9s, float overlapT, uint8_t* keep) {
    if (!w || !h || !h9s || !keep || n <= 0) return MI355_ERR_ARG;
    for (int k = 0; k < n; k++) keep[k] = 1;
    Pt q1[4], q2[4];
    for (int n1 = 1; n1 < n; n1++) {
        const float* m1 = h9s + 9 * n1;
        if (m1[8] == 0) continue;
        image_quad(m1, w[n1], h[n1], q1);
        const float area1 = quad_area(q1);
        bool satisfied = true;
        for (int n2 = 0; n2 < n1; n2++) {
            if (keep[n2] == 0) continue;
            const float* m2 = h9s + 9 * n2;
            if (m2[8] == 0) continue;
            image_quad(m2, w[n2], h[n2], q2);
            std::vector<Pt> cand, ov;
            all_intersections(q1, q2, cand);
            points_in_overlap(q1, q2, cand, ov);
            if (ov.size() == 3) ov.push_back(ov[2]);                             // :2170-2174
            float area2 = 0;
            if (ov.size() == 4) area2 = quad_area(&ov[0]);
            const float ratio = area2 / area1;
            if (ratio > overlapT) { satisfied = false; break; }
        }
        if (!satisfied) keep[n1] = 0;
    }
    keep[n - 1] = 1;
    return MI355_OK;
}
