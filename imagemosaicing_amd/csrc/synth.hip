// csrc/synth.hip -- seeded synthetic "UAV terrain" frames generated directly in HBM (bench / test input only).
//
// A frame pixel (x,y) samples a procedural terrain f(u,v) at (u,v) = A (x,y,1): 6 octaves of lattice value noise
// (smoothstep interpolation) + one anisotropic Gaussian blob per 12 px cell (3x3 neighbourhood evaluated) +
// per-frame gain and additive pseudo-Gaussian noise.  f is a pure function of (u,v,seed): overlapping frames see
// the same ground, so adjacent frames match through the known homography A_i^-1 A_j.  Not part of the
// hot path and never timed.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned hash3(int x, int y, unsigned s) {
    unsigned h = (unsigned)x * 0x9E3779B1u ^ (unsigned)y * 0x85EBCA77u ^ s * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float u01(unsigned h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float value_noise(float u, float v, float cell, unsigned seed) {
    const float fu = u / cell, fv = v / cell;
    const float iu = floorf(fu), iv = floorf(fv);
    float a = fu - iu, b = fv - iv;
    a = a * a * (3.0f - 2.0f * a); b = b * b * (3.0f - 2.0f * b);
    const int x = (int)iu, y = (int)iv;
    const float n00 = u01(hash3(x, y, seed)), n10 = u01(hash3(x + 1, y, seed));
    const float n01 = u01(hash3(x, y + 1, seed)), n11 = u01(hash3(x + 1, y + 1, seed));
    const float t0 = n00 + (n10 - n00) * a, t1 = n01 + (n11 - n01) * a;
    return t0 + (t1 - t0) * b;
}

struct SynthArgs { uint8_t* dst; int w, h, ws; float A[6]; unsigned seed, frame_seed; float gain, noise; };

__global__ __launch_bounds__(256) void synth_kernel(SynthArgs a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.w || y >= a.h) return;
    const float u = a.A[0] * (float)x + a.A[1] * (float)y + a.A[2];
    const float v = a.A[3] * (float)x + a.A[4] * (float)y + a.A[5];
    float acc = 0.0f, amp = 1.0f, tot = 0.0f;
    float cell = 192.0f;
    for (int o = 0; o < 6; o++) { acc += amp * value_noise(u, v, cell, a.seed + 17u * o); tot += amp; amp *= 0.7f; cell *= 0.5f; }
    acc = acc / tot;
    float B = acc * 150.0f + 40.0f, G = acc * 170.0f + 30.0f, R = acc * 120.0f + 60.0f;
    const float bc = 12.0f;
    const int cx = (int)floorf(u / bc), cy = (int)floorf(v / bc);
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int gx = cx + dx, gy = cy + dy;
            const unsigned h0 = hash3(gx, gy, a.seed ^ 0xA511E9B3u);
            if ((h0 & 3u) == 0u) continue;                       // 1 cell in 4 is empty
            const float px = ((float)gx + u01(hash3(gx, gy, a.seed + 1u))) * bc, py = ((float)gy + u01(hash3(gx, gy, a.seed + 2u))) * bc;
            const float sx = 1.2f + 3.0f * u01(hash3(gx, gy, a.seed + 3u)), sy = 1.2f + 3.0f * u01(hash3(gx, gy, a.seed + 4u));
            const float ex = (u - px) / sx, ey = (v - py) / sy;
            const float g = __expf(-0.5f * (ex * ex + ey * ey));
            B += g * (180.0f * u01(hash3(gx, gy, a.seed + 5u)) - 90.0f);
            G += g * (180.0f * u01(hash3(gx, gy, a.seed + 6u)) - 90.0f);
            R += g * (180.0f * u01(hash3(gx, gy, a.seed + 7u)) - 90.0f);
        }
    // additive noise ~ N(0, noise): sum of 4 uniforms, per pixel and channel, seeded by the frame
    uint8_t* d = a.dst + (size_t)y * a.ws + 3 * x;
    float ch[3] = {B, G, R};
    for (int c = 0; c < 3; c++) {
        const unsigned h = hash3(x * 3 + c, y, a.frame_seed);
        const float n4 = u01(h) + u01(h * 0x9E3779B1u + 1u) + u01(h * 0x85EBCA77u + 2u) + u01(h * 0xC2B2AE3Du + 3u);
        float val = ch[c] * a.gain + (n4 - 2.0f) * 1.7320508f * a.noise;
        val = val < 0.0f ? 0.0f : (val > 255.0f ? 255.0f : val);
        d[c] = (uint8_t)(int)(val + 0.5f);
    }
}

}  // namespace

extern "C" int mi355_synth_frame_dev(mi355_ctx* ctx, uint8_t* d_dst, int w, int h, int ws, const float A6[6], uint32_t seed, uint32_t frame_seed,
                                     float gain, float noise_sigma) {
    if (!ctx || !d_dst || !A6 || w <= 0 || h <= 0 || ws < 3 * w) return MI355_ERR_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    SynthArgs a;
    a.dst = d_dst; a.w = w; a.h = h; a.ws = ws; memcpy(a.A, A6, sizeof(a.A)); a.seed = seed; a.frame_seed = frame_seed; a.gain = gain; a.noise = noise_sigma;
    hipLaunchKernelGGL(synth_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, ctx->stream, a);
    MI_HIP(hipGetLastError());
    return MI355_OK;
}
