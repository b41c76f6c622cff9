// csrc/detmath.h -- the fixed transcendental approximations that are part of the feature definitions (oracle/oracle_sift.c,
// oracle/oracle_surf.c): polynomial exp2 / atan2 / sincos evaluated with fmaf in a fixed order, so that the HIP kernels and the
// CPU oracles agree bit for bit.  Shared by sift.hip and surf.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ float det_exp2f(float x) {
    if (x < -126.0f) return 0.0f;
    if (x > 127.0f) x = 127.0f;
    const float n = rintf(x);
    const float f = x - n;
    float p = 1.535336188319500e-4f;
    p = fmaf(p, f, 1.339887440266574e-3f);
    p = fmaf(p, f, 9.618437357674640e-3f);
    p = fmaf(p, f, 5.550332471162809e-2f);
    p = fmaf(p, f, 2.402264791363012e-1f);
    p = fmaf(p, f, 6.931472028550421e-1f);
    p = fmaf(p, f, 1.0f);
    const float s = __uint_as_float((uint32_t)((int)n + 127) << 23);
    return p * s;
}
__device__ __forceinline__ float det_expf(float x) { return det_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float det_atan2deg(float y, float x) {
    const float p1 = 57.283627f, p3 = -18.667446f, p5 = 8.9140005f, p7 = -2.5397246f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + 2.220446e-16f), c2 = c * c;
        a = fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    } else {
        const float c = ax / (ay + 2.220446e-16f), c2 = c * c;
        a = 90.0f - fmaf(fmaf(fmaf(p7, c2, p5), c2, p3), c2, p1) * c;
    }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

__device__ __forceinline__ void det_sincosdeg(float deg, float& sn, float& cs) {
    const float q = rintf(deg * (1.0f / 90.0f));
    const float r = fmaf(-90.0f, q, deg);
    const float t = r * 0.017453292519943295f;
    const float t2 = t * t;
    float sp = fmaf(t2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, t2, 8.3333333e-3f);
    sp = fmaf(sp, t2, -1.6666667e-1f);
    const float s = fmaf(sp * t2, t, t);
    float cp = fmaf(t2, 2.4801587e-5f, -1.3888889e-3f);
    cp = fmaf(cp, t2, 4.1666667e-2f);
    cp = fmaf(cp, t2, -0.5f);
    const float c = fmaf(cp, t2, 1.0f);
    const int k = ((int)q) & 3;
    if (k == 0) { sn = s; cs = c; }
    else if (k == 1) { sn = c; cs = -s; }
    else if (k == 2) { sn = -s; cs = -c; }
    else { sn = -c; cs = s; }
}

