// scratch/blur_kernels.h -- the kernel variants of scratch/blur_lab.hip (lab code, not product)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <type_traits>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int MAX_R = 16, SIFT_BATCH_MAX = 16, FIXPT_SCALE = 48;
typedef int16_t lvl_t;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * n - 2 - p; }
    return p;
}
__device__ __forceinline__ int sat16(float v) {
    int q = (int)rintf(v);
    q = q < -32768 ? -32768 : (q > 32767 ? 32767 : q);
    return q;
}
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
template <int J, int NPP, class F> __device__ __forceinline__ bool static_rows(F& f) {
    if constexpr (J < NPP) { if (!f(std::integral_constant<int, J>{})) return false; return static_rows<J + 1, NPP>(f); }
    else return true;
}

struct Blur16Args {
    const lvl_t* src;
    const uint8_t* bgr[SIFT_BATCH_MAX];
    int bgr_ws[SIFT_BATCH_MAX];
    lvl_t* dst;
    lvl_t* ds;
    int w, h;
    int tiles_x, tiles_y;
    float k[2 * MAX_R + 1];
    float kp[2 * (MAX_R + 1)];               // tap pairs (k[t], k[t-1]) for t = 0 .. R  (k[-1] = 0)
    size_t fstride;
    int nb;
};

// =====================================================================================================================================
// V0: the round-3 kernel, verbatim
// =====================================================================================================================================
template <int R, int D, bool BGR>
__device__ __forceinline__ void blur16_stream_body(const Blur16Args a, int L, int nstrip, int nseg) {
    constexpr int N = 2 * R + 1;
    constexpr int NP0 = ((N + D - 1) / D) * D;
    constexpr int NP = (NP0 & 1) ? NP0 + D : NP0;
    constexpr int RA = (R + 3) & ~3, S = RA - R, SW = 256, BW = SW + 2 * RA;
    constexpr int NG = (S + 2 * R + 4 + 3) / 4;
    __shared__ v4f s_buf[4][2][(BW + 64) / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= a.nb) return;
    unit -= fr * per;
    const lvl_t* src = BGR ? nullptr : a.src + (size_t)fr * a.fstride;
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    lvl_t* dst = a.dst + (size_t)fr * a.fstride;
    lvl_t* ds = a.ds ? a.ds + (size_t)fr * a.fstride : nullptr;
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    if (seg >= nseg) return;
    const int x0 = strip * SW, y0 = seg * L;
    const int lact = (a.h - y0 < L) ? a.h - y0 : L;
    const int nin = lact + 2 * R;
    const int xm = x0 + 4 * lane;
    const int xl = xm < a.w - 4 ? xm : a.w - 4;
    const int chalo = reflect101((lane < RA) ? x0 - RA + lane : (lane < 2 * RA ? x0 + SW + (lane - RA) : x0), a.w);
    const int hpos = (lane < RA) ? lane : (lane < 2 * RA ? SW + lane : BW + lane - 2 * RA);
    const int wv = a.w - x0;
    const bool patch = (wv < SW) && (lane < R);
    const int p_src = patch ? RA + wv - 2 - lane : 0, p_dst = patch ? RA + wv + lane : BW + 32 + (lane & 31);
    const int hm1 = a.h - 1;
    auto src_row = [&](int i) {
        int gy = y0 - R + i;
        gy = gy < 0 ? -gy : gy;
        gy = gy > hm1 ? 2 * hm1 - gy : gy;
        return gy < 0 ? 0 : gy;
    };
    struct Raw { unsigned m0, m1, m2; unsigned h; };
    auto load_raw = [&](int i, Raw& r) {
        const int gy = src_row(i);
        if constexpr (BGR) {
            const uint8_t* rp = bgr + (size_t)gy * bws;
            const unsigned* q = reinterpret_cast<const unsigned*>(rp + 3 * xl);
            r.m0 = q[0]; r.m1 = q[1]; r.m2 = q[2];
            const uint8_t* hp = rp + 3 * chalo;
            r.h = (unsigned)hp[0] | ((unsigned)hp[1] << 8) | ((unsigned)hp[2] << 16);
        } else {
            const lvl_t* rp = src + (size_t)gy * a.w;
            const uint2 q = *reinterpret_cast<const uint2*>(rp + xl);
            r.m0 = q.x; r.m1 = q.y; r.m2 = 0;
            r.h = (unsigned)(unsigned short)rp[chalo];
        }
    };
    auto row_values = [&](const Raw& r, v4f& m, float& hv) {
        if constexpr (BGR) {
            auto g = [](unsigned b, unsigned gg, unsigned rr) { return (float)((int)((1868u * b + 9617u * gg + 4899u * rr + 8192u) >> 14) * FIXPT_SCALE); };
            m.x = g(r.m0 & 255u, (r.m0 >> 8) & 255u, (r.m0 >> 16) & 255u);
            m.y = g(r.m0 >> 24, r.m1 & 255u, (r.m1 >> 8) & 255u);
            m.z = g((r.m1 >> 16) & 255u, r.m1 >> 24, r.m2 & 255u);
            m.w = g((r.m2 >> 8) & 255u, (r.m2 >> 16) & 255u, r.m2 >> 24);
            hv = g(r.h & 255u, (r.h >> 8) & 255u, (r.h >> 16) & 255u);
        } else {
            m.x = (float)(int)(short)(r.m0 & 0xffffu); m.y = (float)((int)r.m0 >> 16);
            m.z = (float)(int)(short)(r.m1 & 0xffffu); m.w = (float)((int)r.m1 >> 16);
            hv = (float)(int)(short)(r.h & 0xffffu);
        }
    };
    Raw pf[D];
#pragma unroll
    for (int d = 0; d < D; d++) load_raw(d < nin ? d : nin - 1, pf[d]);
    v2f ring01[NP], ring23[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) { ring01[q] = (v2f){0.0f, 0.0f}; ring23[q] = (v2f){0.0f, 0.0f}; }
    float* const bufs = reinterpret_cast<float*>(&s_buf[wave][0][0]);
    auto put_row = [&](const Raw& r, float* buf) {
        v4f m; float hv;
        row_values(r, m, hv);
        *reinterpret_cast<v4f*>(buf + RA + 4 * lane) = m;
        buf[hpos] = hv;
    };
    auto patch_row = [&](float* buf) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        { const float t = buf[p_src]; buf[p_dst] = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    put_row(pf[0], bufs);
    load_raw(D < nin ? D : nin - 1, pf[0]);
    patch_row(bufs);
    for (int base = 0; base < nin; base += NP) {
        auto step = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;
            const int i = base + j;
            if (i >= nin) return false;
            float* const buf = bufs + (j & 1) * (BW + 64);
            float* const nbuf = bufs + ((j + 1) & 1) * (BW + 64);
            const v4f* w4 = reinterpret_cast<const v4f*>(buf) + lane;
            float e[NG * 4];
#pragma unroll
            for (int g = 0; g < NG; g++) { const v4f tt = w4[g]; e[4 * g] = tt.x; e[4 * g + 1] = tt.y; e[4 * g + 2] = tt.z; e[4 * g + 3] = tt.w; }
            const bool more = i + 1 < nin;
            if (more) {
                put_row(pf[(j + 1) % D], nbuf);
                load_raw(i + 1 + D < nin ? i + 1 + D : nin - 1, pf[(j + 1) % D]);
            }
            v2f r01, r23;
            {
                const v2f kk = {a.k[0], a.k[0]};
                r01 = kk * (v2f){e[S], e[S + 1]}; r23 = kk * (v2f){e[S + 2], e[S + 3]};
            }
#pragma unroll
            for (int t = 1; t <= 2 * R; t++) {
                const v2f kk = {a.k[t], a.k[t]};
                const v2f p01 = kk * (v2f){e[S + t], e[S + t + 1]}, p23 = kk * (v2f){e[S + t + 2], e[S + t + 3]};
                r01 = r01 + p01; r23 = r23 + p23;
            }
            ring01[j % NP] = r01; ring23[j % NP] = r23;
            if (more) patch_row(nbuf);
            if (i >= 2 * R) {
                constexpr int c = ((j - R) % NP + NP) % NP;
                v2f s01, s23;
                { const v2f kk = {a.k[R], a.k[R]}; s01 = kk * ring01[c]; s23 = kk * ring23[c]; }
#pragma unroll
                for (int jj = 1; jj <= R; jj++) {
                    const int up = ((j - R + jj) % NP + NP) % NP, dn = ((j - R - jj) % NP + NP) % NP;
                    const v2f kk = {a.k[R + jj], a.k[R + jj]};
                    const v2f a01 = ring01[up] + ring01[dn], a23 = ring23[up] + ring23[dn];
                    const v2f p01 = kk * a01, p23 = kk * a23;
                    s01 = s01 + p01; s23 = s23 + p23;
                }
                const int o0 = sat16(s01.x), o1 = sat16(s01.y), o2 = sat16(s23.x), o3 = sat16(s23.y);
                const int gy = y0 + i - 2 * R;
                if (xm < a.w) {
                    *reinterpret_cast<uint2*>(dst + (size_t)gy * a.w + xm) =
                        make_uint2((unsigned)(o0 & 0xffff) | ((unsigned)o1 << 16), (unsigned)(o2 & 0xffff) | ((unsigned)o3 << 16));
                    if (ds && !(gy & 1) && (gy >> 1) < (a.h >> 1))
                        *reinterpret_cast<unsigned*>(ds + (size_t)(gy >> 1) * (a.w >> 1) + (xm >> 1)) = (unsigned)(o0 & 0xffff) | ((unsigned)o2 << 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            return true;
        };
        if (!static_rows<0, NP>(step)) return;
    }
}
template <int R, int D, bool BGR, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void blur_v0(Blur16Args a, int L, int nstrip, int nseg) {
    blur16_stream_body<R, D, BGR>(a, L, nstrip, nseg);
}

// =====================================================================================================================================
// V2: PX (2 or 4) pixels per lane, tap-staggered row pass (no register moves), no LDS patch round trip, magic-number rounding,
//     incremental row pointers
// =====================================================================================================================================
// Row pass of the output pair (p, p + 1): acc += (k[t], k[t - 1]) * (e[p + t], e[p + t]) for t = 0 .. 2R + 1 with k[-1] = k[2R + 1] = 0:
// output p takes its taps 0 .. 2R in ascending order, output p + 1 the same one step later; the products with the zero taps are
// +0 and x + 0 = x exactly, so every product and sum is the reference's.  Both lanes of the packed instruction read the SAME window
// register (op_sel broadcast), so no pair is ever misaligned.  Pairs (k[t], k[t-1]) for t > R are the swapped pairs of 2R + 1 - t.
template <int R, int PX, int D, bool BGR>
__device__ __forceinline__ void blur_v2_body(const Blur16Args a, int L, int nstrip, int nseg) {
    static_assert(PX == 2 || PX == 4, "pixels per lane");
    constexpr int N = 2 * R + 1;
    constexpr int NP0 = ((N + D - 1) / D) * D;
    constexpr int NP = (NP0 & 1) ? NP0 + D : NP0;            // even (two LDS rows alternate) and a multiple of D
    constexpr int RA = (R + PX - 1) / PX * PX, S = RA - R, SW = 64 * PX, BW = SW + 2 * RA;
    constexpr int WN = S + 2 * R + PX;                        // window floats a lane needs: e[0 .. WN)
    constexpr int NG = (WN + PX - 1) / PX;                    // window reads of PX floats
    constexpr int BUF = (BW + 2 * PX + 63) & ~63;             // + dump area, kept a multiple of 64 floats
    constexpr int NC = PX / 2;                                // packed accumulator chains per lane
    constexpr int WPB = 4;                                    // waves per workgroup
    static_assert(2 * RA + R <= 64, "halo lanes");
    __shared__ __attribute__((aligned(16))) float s_buf[WPB][2][BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPB + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= a.nb) return;
    unit -= fr * per;
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    if (seg >= nseg) return;
    const int x0 = strip * SW, y0 = seg * L;
    const int lact = (a.h - y0 < L) ? a.h - y0 : L;
    const int nin = lact + 2 * R;
    const int wv = a.w - x0;                                  // valid columns of this strip
    // columns: the lane's own PX (clamped into the row for a partial last strip); one more sample for the lanes < 2RA (the halos,
    // reflect-101) and, in a partial last strip, for the lanes 2RA .. 2RA + R - 1: the reflected columns right of the image, written
    // AFTER the main samples (LDS operations of one wave execute in order) over whatever the clamped main loads left there
    const int xm = x0 + PX * lane;
    const int xl = xm < a.w - PX ? xm : a.w - PX;
    int hcol, hpos;
    if (lane < RA) { hcol = x0 - RA + lane; hpos = lane; }
    else if (lane < 2 * RA) { hcol = x0 + SW + (lane - RA); hpos = SW + lane; }
    else if (wv < SW && lane < 2 * RA + R) { hcol = x0 + wv + (lane - 2 * RA); hpos = RA + wv + (lane - 2 * RA); }
    else { hcol = x0; hpos = BW + (lane & (2 * PX - 1)); }
    hcol = reflect101(hcol, a.w);
    const int hm1 = a.h - 1;
    // source rows: reflect-101 walk, incremental
    int gy = y0 - R; int dir = 1;
    if (gy < 0) { gy = -gy; dir = -1; }
    if (gy > hm1) gy = hm1;                                   // (cannot happen: y0 < h)
    auto advance = [&]() {
        if (gy == 0) dir = 1;
        if (gy >= hm1) dir = -1;
        gy += dir;
        if (hm1 == 0) gy = 0;
    };
    const size_t fro = (size_t)fr * a.fstride;
    const lvl_t* src = BGR ? nullptr : a.src + fro;
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    struct Raw { unsigned m0, m1, m2; unsigned h; };
    auto load_raw = [&](Raw& r) {                            // loads source row gy, then advances
        if constexpr (BGR) {
            const uint8_t* rp = bgr + (size_t)gy * bws;
            if constexpr (PX == 4) {
                const unsigned* q = reinterpret_cast<const unsigned*>(rp + 3 * xl);
                r.m0 = q[0]; r.m1 = q[1]; r.m2 = q[2];
            } else {
                const unsigned* q = reinterpret_cast<const unsigned*>(rp + ((3 * xl) & ~3));     // 6 bytes from byte 3 xl (a multiple of 2... of 6)
                r.m0 = q[0]; r.m1 = q[1]; r.m2 = 0;
            }
            const uint8_t* hp = rp + 3 * hcol;
            r.h = (unsigned)hp[0] | ((unsigned)hp[1] << 8) | ((unsigned)hp[2] << 16);
        } else {
            const lvl_t* rp = src + (size_t)gy * a.w;
            if constexpr (PX == 4) { const uint2 q = *reinterpret_cast<const uint2*>(rp + xl); r.m0 = q.x; r.m1 = q.y; }
            else { r.m0 = *reinterpret_cast<const unsigned*>(rp + xl); r.m1 = 0; }
            r.m2 = 0;
            r.h = (unsigned)(unsigned short)rp[hcol];
        }
        advance();
    };
    auto g48 = [](unsigned b, unsigned gg, unsigned rr) { return (float)((int)((1868u * b + 9617u * gg + 4899u * rr + 8192u) >> 14) * FIXPT_SCALE); };
    float* const bufs = &s_buf[wave][0][0];
    auto put_row = [&](const Raw& r, float* buf) {
        float hv;
        if constexpr (BGR) {
            hv = g48(r.h & 255u, (r.h >> 8) & 255u, (r.h >> 16) & 255u);
            if constexpr (PX == 4) {
                v4f m;
                m.x = g48(r.m0 & 255u, (r.m0 >> 8) & 255u, (r.m0 >> 16) & 255u);
                m.y = g48(r.m0 >> 24, r.m1 & 255u, (r.m1 >> 8) & 255u);
                m.z = g48((r.m1 >> 16) & 255u, r.m1 >> 24, r.m2 & 255u);
                m.w = g48((r.m2 >> 8) & 255u, (r.m2 >> 16) & 255u, r.m2 >> 24);
                *reinterpret_cast<v4f*>(buf + RA + 4 * lane) = m;
            } else {
                const unsigned long long q = (((unsigned long long)r.m1 << 32) | r.m0) >> (((3 * xl) & 3) * 8);
                const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
                v2f m;
                m.x = g48(lo & 255u, (lo >> 8) & 255u, (lo >> 16) & 255u);
                m.y = g48(lo >> 24, hi & 255u, (hi >> 8) & 255u);
                *reinterpret_cast<v2f*>(buf + RA + 2 * lane) = m;
            }
        } else {
            hv = (float)(int)(short)(r.h & 0xffffu);
            if constexpr (PX == 4) {
                v4f m;
                m.x = (float)(int)(short)(r.m0 & 0xffffu); m.y = (float)((int)r.m0 >> 16);
                m.z = (float)(int)(short)(r.m1 & 0xffffu); m.w = (float)((int)r.m1 >> 16);
                *reinterpret_cast<v4f*>(buf + RA + 4 * lane) = m;
            } else {
                v2f m;
                m.x = (float)(int)(short)(r.m0 & 0xffffu); m.y = (float)((int)r.m0 >> 16);
                *reinterpret_cast<v2f*>(buf + RA + 2 * lane) = m;
            }
        }
        buf[hpos] = hv;
    };
    // tap pairs in scalar registers
    v2f kp[R + 1];
#pragma unroll
    for (int t = 0; t <= R; t++) kp[t] = (v2f){a.kp[2 * t], a.kp[2 * t + 1]};
    Raw pf[D];
#pragma unroll
    for (int d = 0; d < D; d++) load_raw(pf[d]);
    v2f ring[NC][NP];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int q = 0; q < NP; q++) ring[c][q] = (v2f){0.0f, 0.0f};
    put_row(pf[0], bufs);
    load_raw(pf[0]);
    lvl_t* drow = a.dst + fro + (size_t)y0 * a.w + xm;                                       // output row pointer of this lane
    lvl_t* dsrow = a.ds ? a.ds + fro + (size_t)(y0 >> 1) * (a.w >> 1) + (xm >> 1) : nullptr;   // (segments start on even rows: launcher)
    const bool in_img = xm < a.w, has_ds = a.ds != nullptr;
    for (int base = 0; base < nin; base += NP) {
        auto step = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;
            const int i = base + j;
            if (i >= nin) return false;
            float* const buf = bufs + (j & 1) * BUF;
            float* const nbuf = bufs + ((j + 1) & 1) * BUF;
            float e[NG * PX];
            if constexpr (PX == 4) {
                const v4f* w4 = reinterpret_cast<const v4f*>(buf) + lane;
#pragma unroll
                for (int g = 0; g < NG; g++) { const v4f tt = w4[g]; e[4 * g] = tt.x; e[4 * g + 1] = tt.y; e[4 * g + 2] = tt.z; e[4 * g + 3] = tt.w; }
            } else {
                const v2f* w2 = reinterpret_cast<const v2f*>(buf) + lane;
#pragma unroll
                for (int g = 0; g < NG; g++) { const v2f tt = w2[g]; e[2 * g] = tt.x; e[2 * g + 1] = tt.y; }
            }
            {   // always (past the segment's last row the walk stays inside the image and the rows are never used): a fixed number of
                // loads per step lets the compiler count its vmcnt waits
                put_row(pf[(j + 1) % D], nbuf);
                load_raw(pf[(j + 1) % D]);
            }
            // row pass
#pragma unroll
            for (int c = 0; c < NC; c++) {
                v2f acc = kp[0] * (v2f){e[S + 2 * c], e[S + 2 * c]};
#pragma unroll
                for (int t = 1; t <= 2 * R + 1; t++) {
                    const v2f kk = t <= R ? kp[t] : (v2f){kp[2 * R + 1 - t].y, kp[2 * R + 1 - t].x};
                    const float ee = e[S + 2 * c + t];
                    const v2f p = kk * (v2f){ee, ee};
                    acc = acc + p;
                }
                ring[c][j % NP] = acc;
            }
            if (i >= 2 * R) {
                constexpr int cc = ((j - R) % NP + NP) % NP;
                v2f s[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) s[c] = (v2f){kp[R].x, kp[R].x} * ring[c][cc];
#pragma unroll
                for (int jj = 1; jj <= R; jj++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int up = ((j - R + jj) % NP + NP) % NP, dn = ((j - R - jj) % NP + NP) % NP;
                    const v2f kk = {kp[R - jj].x, kp[R - jj].x};
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const v2f aa = ring[c][up] + ring[c][dn];
                        const v2f p = kk * aa;
                        s[c] = s[c] + p;
                    }
                }
                // round half to even into 16 bits: s in [0, 32767] (taps positive and normalised, samples in [0, 255 x 48]), so
                // s + 1.5 x 2^23 has the integer in its low mantissa bits and saturate_cast never saturates
                unsigned o[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const v2f m = s[c] + (v2f){12582912.0f, 12582912.0f};
                    o[c] = __builtin_amdgcn_perm(__float_as_uint(m.y), __float_as_uint(m.x), 0x05040100u);
                }
                if (in_img) {
                    if constexpr (PX == 4) *reinterpret_cast<uint2*>(drow) = make_uint2(o[0], o[1]);
                    else *reinterpret_cast<unsigned*>(drow) = o[0];
                    if (has_ds && !((i - 2 * R) & 1) && ((y0 + i - 2 * R) >> 1) < (a.h >> 1)) {
                        if constexpr (PX == 4) *reinterpret_cast<unsigned*>(dsrow) = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u);
                        else *reinterpret_cast<unsigned short*>(dsrow) = (unsigned short)o[0];
                    }
                }
                drow += a.w;
                if (!((i - 2 * R) & 1)) dsrow += (a.w >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            return true;
        };
        if (!static_rows<0, NP>(step)) return;
    }
}
template <int R, int PX, int D, bool BGR, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void blur_v2(Blur16Args a, int L, int nstrip, int nseg) {
    blur_v2_body<R, PX, D, BGR>(a, L, nstrip, nseg);
}

// =====================================================================================================================================
// V3: V2 + a deterministic number of memory operations per step (so that the compiler's vmcnt waits are counted instead of 0: gfx950
//     counts loads AND stores in vmcnt, a wait for the row prefetched D steps ago must not wait for the store issued a moment ago),
//     warm-up rows (row pass only) in their own unrolled loop, scalar row pointers + 32-bit lane offsets, exact-width window reads
// =====================================================================================================================================
template <int R, int PX, int D, bool BGR, bool DS>
__device__ __forceinline__ void blur_v3_body(const Blur16Args a, int L, int nstrip, int nseg) {
    static_assert(PX == 2 || PX == 4, "pixels per lane");
    constexpr int N = 2 * R + 1;
    constexpr int NP0 = ((N + D - 1) / D) * D;
    constexpr int NP = (NP0 & 1) ? NP0 + D : NP0;            // even (two LDS rows alternate) and a multiple of D
    constexpr int RA = (R + PX - 1) / PX * PX, S = RA - R, SW = 64 * PX, BW = SW + 2 * RA;
    constexpr int WN = S + 2 * R + PX;                        // window floats a lane needs: e[S .. WN)
    constexpr int BUF = (BW + 2 * PX + 63) & ~63;
    constexpr int NC = PX / 2;
    constexpr int WPB = 4;
    static_assert(2 * RA + R <= 64, "halo lanes");
    static_assert((2 * R) % 2 == 0, "");
    __shared__ __attribute__((aligned(16))) float s_buf[WPB][2][BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPB + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= a.nb) return;
    unit -= fr * per;
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    if (seg >= nseg) return;
    const int x0 = strip * SW, y0 = seg * L;
    const int lact = (a.h - y0 < L) ? a.h - y0 : L;
    const int nin = lact + 2 * R;
    const int wv = a.w - x0;
    const int xm = x0 + PX * lane;
    const int xl = xm < a.w - PX ? xm : a.w - PX;
    int hcol, hpos;
    if (lane < RA) { hcol = x0 - RA + lane; hpos = lane; }
    else if (lane < 2 * RA) { hcol = x0 + SW + (lane - RA); hpos = SW + lane; }
    else if (wv < SW && lane < 2 * RA + R) { hcol = x0 + wv + (lane - 2 * RA); hpos = RA + wv + (lane - 2 * RA); }
    else { hcol = x0; hpos = BW + (lane & (2 * PX - 1)); }
    hcol = reflect101(hcol, a.w);
    const int hm1 = a.h - 1;
    const size_t fro = (size_t)fr * a.fstride;
    // source rows: reflect-101 walk with a scalar row pointer
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    const long rowb = BGR ? (long)bws : 2L * a.w;                                  // bytes per source row
    int gy = y0 - R; int dir = 1;
    if (gy < 0) { gy = -gy; dir = -1; }
    const char* rp = (BGR ? reinterpret_cast<const char*>(bgr) : reinterpret_cast<const char*>(a.src + fro)) + (long)gy * rowb;
    const unsigned moff = BGR ? (PX == 4 ? 3u * xl : ((3u * xl) & ~3u)) : 2u * xl;  // byte offsets of the lane's samples in a row
    const unsigned hoff = BGR ? 3u * hcol : 2u * hcol;
    const unsigned bsh = BGR && PX == 2 ? ((3u * xl) & 3u) * 8u : 0u;
    struct Raw { unsigned m0, m1, m2; unsigned h; };
    auto load_raw = [&](Raw& r) {                            // loads source row gy, then advances (always: past the segment's last row the
        if constexpr (BGR) {                                 // walk stays inside the image and the rows are never used)
            const unsigned* q = reinterpret_cast<const unsigned*>(rp + moff);
            r.m0 = q[0]; r.m1 = q[1]; if constexpr (PX == 4) r.m2 = q[2]; else r.m2 = 0;
            const uint8_t* hp = reinterpret_cast<const uint8_t*>(rp + hoff);
            r.h = (unsigned)hp[0] | ((unsigned)hp[1] << 8) | ((unsigned)hp[2] << 16);
        } else {
            if constexpr (PX == 4) { const uint2 q = *reinterpret_cast<const uint2*>(rp + moff); r.m0 = q.x; r.m1 = q.y; }
            else { r.m0 = *reinterpret_cast<const unsigned*>(rp + moff); r.m1 = 0; }
            r.m2 = 0;
            r.h = (unsigned)*reinterpret_cast<const unsigned short*>(rp + hoff);
        }
        if (gy == 0) dir = 1;
        if (gy >= hm1) dir = -1;
        if (hm1 == 0) dir = 0;
        gy += dir;
        rp += dir > 0 ? rowb : (dir < 0 ? -rowb : 0L);
    };
    auto g48 = [](unsigned b, unsigned gg, unsigned rr) { return (float)((int)((1868u * b + 9617u * gg + 4899u * rr + 8192u) >> 14) * FIXPT_SCALE); };
    float* const bufs = &s_buf[wave][0][0];
    auto put_row = [&](const Raw& r, float* buf) {
        float hv;
        if constexpr (BGR) {
            hv = g48(r.h & 255u, (r.h >> 8) & 255u, (r.h >> 16) & 255u);
            if constexpr (PX == 4) {
                v4f m;
                m.x = g48(r.m0 & 255u, (r.m0 >> 8) & 255u, (r.m0 >> 16) & 255u);
                m.y = g48(r.m0 >> 24, r.m1 & 255u, (r.m1 >> 8) & 255u);
                m.z = g48((r.m1 >> 16) & 255u, r.m1 >> 24, r.m2 & 255u);
                m.w = g48((r.m2 >> 8) & 255u, (r.m2 >> 16) & 255u, r.m2 >> 24);
                *reinterpret_cast<v4f*>(buf + RA + 4 * lane) = m;
            } else {
                const unsigned long long q = (((unsigned long long)r.m1 << 32) | r.m0) >> bsh;
                const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
                v2f m;
                m.x = g48(lo & 255u, (lo >> 8) & 255u, (lo >> 16) & 255u);
                m.y = g48(lo >> 24, hi & 255u, (hi >> 8) & 255u);
                *reinterpret_cast<v2f*>(buf + RA + 2 * lane) = m;
            }
        } else {
            hv = (float)(int)(short)(r.h & 0xffffu);
            if constexpr (PX == 4) {
                v4f m;
                m.x = (float)(int)(short)(r.m0 & 0xffffu); m.y = (float)((int)r.m0 >> 16);
                m.z = (float)(int)(short)(r.m1 & 0xffffu); m.w = (float)((int)r.m1 >> 16);
                *reinterpret_cast<v4f*>(buf + RA + 4 * lane) = m;
            } else {
                v2f m;
                m.x = (float)(int)(short)(r.m0 & 0xffffu); m.y = (float)((int)r.m0 >> 16);
                *reinterpret_cast<v2f*>(buf + RA + 2 * lane) = m;
            }
        }
        buf[hpos] = hv;
    };
    v2f kp[R + 1];
#pragma unroll
    for (int t = 0; t <= R; t++) kp[t] = (v2f){a.kp[2 * t], a.kp[2 * t + 1]};
    Raw pf[D];
#pragma unroll
    for (int d = 0; d < D; d++) load_raw(pf[d]);
    v2f ring[NC][NP];
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int q = 0; q < NP; q++) ring[c][q] = (v2f){0.0f, 0.0f};
    put_row(pf[0], bufs);
    load_raw(pf[0]);
    // output rows: scalar row pointer + lane offset; lanes right of the image (partial last strip) store into the lane's clamped
    // column of a scratch row?  no: they keep their exec-masked store (the skipped store only makes the compiler's count conservative)
    char* dp = reinterpret_cast<char*>(a.dst + fro) + 2L * (long)y0 * a.w;
    char* dsp = DS ? reinterpret_cast<char*>(a.ds + fro) + 2L * (long)(y0 >> 1) * (a.w >> 1) : nullptr;
    const unsigned doff = 2u * xm, dsoff = 2u * (xm >> 1);
    const bool in_img = xm < a.w;
    const int ds_rows = a.h >> 1;
    int yo = y0;                                               // next output row
    // one row step: SLOT = ring slot of the row that arrives, PAR = its LDS buffer, PFS = prefetch slot of the row after it
    auto step = [&](auto slot_c, auto col_c) {
        constexpr int slot = decltype(slot_c)::value;
        constexpr bool COL = decltype(col_c)::value;
        constexpr int PAR = slot & 1, PFS = (slot + 1) % D;
        float* const buf = bufs + PAR * BUF;
        float* const nbuf = bufs + (PAR ^ 1) * BUF;
        float e[WN + 4];
        // window e[S .. WN): exact-width reads (no register of a read is dead: dead halves made the compiler serialise the reads)
        {
            const float* wb = buf + PX * lane;
            int m = S;
            if constexpr (PX == 4) {
                if constexpr ((S & 3) == 1) { e[1] = wb[1]; const v2f t = *reinterpret_cast<const v2f*>(wb + 2); e[2] = t.x; e[3] = t.y; m = 4; }
                if constexpr ((S & 3) == 2) { const v2f t = *reinterpret_cast<const v2f*>(wb + 2); e[2] = t.x; e[3] = t.y; m = 4; }
                if constexpr ((S & 3) == 3) { e[3] = wb[3]; m = 4; }
#pragma unroll
                for (; m + 4 <= WN; m += 4) { const v4f t = *reinterpret_cast<const v4f*>(wb + m); e[m] = t.x; e[m + 1] = t.y; e[m + 2] = t.z; e[m + 3] = t.w; }
                if (WN - m >= 2) { const v2f t = *reinterpret_cast<const v2f*>(wb + m); e[m] = t.x; e[m + 1] = t.y; m += 2; }
                if (WN - m >= 1) { e[m] = wb[m]; m += 1; }
            } else {
                if constexpr ((S & 1) == 1) { e[S] = wb[S]; m = S + 1; }
#pragma unroll
                for (; m + 2 <= WN; m += 2) { const v2f t = *reinterpret_cast<const v2f*>(wb + m); e[m] = t.x; e[m + 1] = t.y; }
                if (WN - m >= 1) { e[m] = wb[m]; m += 1; }
            }
        }
        put_row(pf[PFS], nbuf);
        load_raw(pf[PFS]);
        // row pass
#pragma unroll
        for (int c = 0; c < NC; c++) {
            v2f acc = kp[0] * (v2f){e[S + 2 * c], e[S + 2 * c]};
#pragma unroll
            for (int t = 1; t <= 2 * R + 1; t++) {
                const v2f kk = t <= R ? kp[t] : (v2f){kp[2 * R + 1 - t].y, kp[2 * R + 1 - t].x};
                const float ee = e[S + 2 * c + t];
                const v2f p = kk * (v2f){ee, ee};
                acc = acc + p;
            }
            ring[c][slot] = acc;
        }
        if constexpr (COL) {
            constexpr int cc = ((slot - R) % NP + NP) % NP;
            v2f s[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) s[c] = (v2f){kp[R].x, kp[R].x} * ring[c][cc];
#pragma unroll
            for (int jj = 1; jj <= R; jj++) {
                const int up = ((slot - R + jj) % NP + NP) % NP, dn = ((slot - R - jj) % NP + NP) % NP;
                const v2f kk = {kp[R - jj].x, kp[R - jj].x};
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const v2f aa = ring[c][up] + ring[c][dn];
                    const v2f p = kk * aa;
                    s[c] = s[c] + p;
                }
            }
            unsigned o[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const v2f m = s[c] + (v2f){12582912.0f, 12582912.0f};
                o[c] = __builtin_amdgcn_perm(__float_as_uint(m.y), __float_as_uint(m.x), 0x05040100u);
            }
            if (in_img) {
                if constexpr (PX == 4) *reinterpret_cast<uint2*>(dp + doff) = make_uint2(o[0], o[1]);
                else *reinterpret_cast<unsigned*>(dp + doff) = o[0];
            }
            dp += 2L * a.w;
            if constexpr (DS && (((slot - 2 * R) % 2 + 2) % 2 == 0)) {      // segments start on even rows and the ring size is even: slot parity = row parity
                if (in_img && (yo >> 1) < ds_rows) {
                    if constexpr (PX == 4) *reinterpret_cast<unsigned*>(dsp + dsoff) = __builtin_amdgcn_perm(o[1], o[0], 0x05040100u);
                    else *reinterpret_cast<unsigned short*>(dsp + dsoff) = (unsigned short)o[0];
                }
                dsp += 2L * (a.w >> 1);
            }
            yo++;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // warm-up: the first 2R rows only feed the ring
    {
        auto warm = [&](auto jc) -> bool { step(jc, std::false_type{}); return true; };
        static_rows<0, 2 * R>(warm);
    }
    for (int i0 = 2 * R; i0 < nin; i0 += NP) {
        const int left = nin - i0;
        auto body = [&](auto jc) -> bool {
            constexpr int j = decltype(jc)::value;
            if (j >= left) return false;
            step(std::integral_constant<int, (2 * R + j) % NP>{}, std::true_type{});
            return true;
        };
        if (!static_rows<0, NP>(body)) return;
    }
}
template <int R, int PX, int D, bool BGR, bool DS, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void blur_v3(Blur16Args a, int L, int nstrip, int nseg) {
    blur_v3_body<R, PX, D, BGR, DS>(a, L, nstrip, nseg);
}


// =====================================================================================================================================
// V4: the row loop in hand-scheduled assembly (imagemosaicing_amd/csrc/gen_blur16_asm.py); HIP computes the wave's geometry only
// =====================================================================================================================================
#include "../imagemosaicing_amd/csrc/blur16_asm.inc"
template <int R, bool BGR, bool DS>
__device__ __forceinline__ void blur_asm_body(const Blur16Args& a, int L, int nstrip, int nseg) {
    constexpr int RA = (R + 3) / 4 * 4, SW = 256, BW = SW + 2 * RA;
    constexpr int BUF = (BW + 8 + 63) & ~63;
    constexpr int WPB = 4;
    static_assert(2 * RA + R <= 64, "halo lanes");
    __shared__ __attribute__((aligned(16))) float s_buf[WPB][2][BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int unit = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * WPB + wave);
    const int per = nstrip * nseg, fr = unit / per;
    if (fr >= a.nb) return;
    unit -= fr * per;
    const int seg = unit / nstrip, strip = unit - seg * nstrip;
    if (seg >= nseg) return;
    const int x0 = strip * SW, y0 = seg * L;
    const int lact = (a.h - y0 < L) ? a.h - y0 : L;
    const int wv = a.w - x0;
    const int xm = x0 + 4 * lane;
    const int xl = xm < a.w - 4 ? xm : a.w - 4;
    int hcol, hpos;
    if (lane < RA) { hcol = x0 - RA + lane; hpos = lane; }
    else if (lane < 2 * RA) { hcol = x0 + SW + (lane - RA); hpos = SW + lane; }
    else if (wv < SW && lane < 2 * RA + R) { hcol = x0 + wv + (lane - 2 * RA); hpos = RA + wv + (lane - 2 * RA); }
    else { hcol = x0; hpos = BW + (lane & 7); }
    hcol = reflect101(hcol, a.w);
    const size_t fro = (size_t)fr * a.fstride;
    const uint8_t* bgr = a.bgr[0]; int bws = a.bgr_ws[0];
    if (BGR) {
#pragma unroll
        for (int q = 1; q < SIFT_BATCH_MAX; q++) if (fr == q) { bgr = a.bgr[q]; bws = a.bgr_ws[q]; }
    }
    const int rowb = BGR ? bws : 2 * a.w;
    int gy = y0 - R, dir = 1;
    if (gy < 0) { gy = -gy; dir = -1; }
    const unsigned long long rp = (unsigned long long)(BGR ? reinterpret_cast<uintptr_t>(bgr) : reinterpret_cast<uintptr_t>(a.src + fro)) + (unsigned long long)gy * (unsigned)rowb;
    const unsigned moff = BGR ? 3u * xl : 2u * xl;
    unsigned hoff, hsh = 0;
    if (BGR) {                                   // 8 bytes that hold the halo pixel's three and lie inside the row
        const int b0 = 3 * hcol; int st = b0 & ~3; if (st + 8 > bws) st = bws - 8;
        hoff = (unsigned)st; hsh = 8u * (unsigned)(b0 - st);
    } else hoff = 2u * hcol;
    float* const bufs = &s_buf[wave][0][0];
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(bufs);
    const unsigned lds_win = lds0 + 16u * lane, lds_main = lds0 + 4u * (RA + 4 * lane), lds_halo = lds0 + 4u * hpos;
    const unsigned long long dp = (unsigned long long)reinterpret_cast<uintptr_t>(a.dst + fro) + 2ull * (unsigned long long)y0 * a.w;
    const unsigned long long dsp = DS ? (unsigned long long)reinterpret_cast<uintptr_t>(a.ds + fro) + 2ull * (unsigned long long)(y0 >> 1) * (a.w >> 1) : 0ull;
    const unsigned doff = 2u * xm, dsoff = 2u * (xm >> 1);
    const unsigned long long smask = __ballot(xm < a.w);
    const unsigned long long kp = (unsigned long long)reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(Blur16Args, kp);
    const int n = lact + 2 * R + 1;
    const int hm1 = a.h - 1, dstr = 2 * a.w;
#define BLUR_ASM_CALL(NAME) NAME(lds_win, lds_main, lds_halo, moff, hoff, doff, dsoff, rp, dp, dsp, kp, smask, rowb, gy, dir, hm1, dstr, n)
    if constexpr (BGR) { static_assert(R == 6, "base level"); blur16_asm_r6_bgr(lds_win, lds_main, lds_halo, moff, hoff, doff, dsoff, hsh, rp, dp, dsp, kp, smask, rowb, gy, dir, hm1, dstr, n); }
    else if constexpr (R == 5) BLUR_ASM_CALL(blur16_asm_r5);
    else if constexpr (R == 6) BLUR_ASM_CALL(blur16_asm_r6);
    else if constexpr (R == 8 && DS) BLUR_ASM_CALL(blur16_asm_r8_ds);
    else if constexpr (R == 8) BLUR_ASM_CALL(blur16_asm_r8);
    else if constexpr (R == 10) BLUR_ASM_CALL(blur16_asm_r10);
    else if constexpr (R == 13) BLUR_ASM_CALL(blur16_asm_r13);
#undef BLUR_ASM_CALL
}
template <int R, bool BGR, bool DS, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void blur_v4(Blur16Args a, int L, int nstrip, int nseg) {
    blur_asm_body<R, BGR, DS>(a, L, nstrip, nseg);
}
