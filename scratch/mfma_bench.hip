// scratch/mfma_bench.hip -- what the matrix pipe gives for the shapes bf_match_kernel could use (dependent chains, wave counts, i8)
// hipcc --offload-arch=gfx950 -O3 -o scratch/mfma_bench scratch/mfma_bench.hip && ./scratch/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(16 * sizeof(int)))) int i32x16;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int NACC, int READ> __global__ __launch_bounds__(512) void k_bf16(const bf16x8* in, float* out, int iters) {
    bf16x8 a[8], b[8];
    for (int k = 0; k < 8; k++) { a[k] = in[threadIdx.x * 16 + k]; b[k] = in[threadIdx.x * 16 + 8 + k]; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int e = 0; e < 16; e++) acc[n][e] = 0.0f;
    float m = 0.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc[k % NACC], 0, 0, 0);
        if (READ) { m = fmaxf(m, acc[0][0]); for (int e = 0; e < 16; e++) acc[0][e] = 0.0f; }
    }
    float s = m;
    for (int n = 0; n < NACC; n++) for (int e = 0; e < 16; e++) s += acc[n][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NACC, int READ> __global__ __launch_bounds__(512) void k_i8(const i32x4* in, int* out, int iters) {
    i32x4 a[4], b[4];
    for (int k = 0; k < 4; k++) { a[k] = in[threadIdx.x * 8 + k]; b[k] = in[threadIdx.x * 8 + 4 + k]; }
    i32x16 acc[NACC];
    for (int n = 0; n < NACC; n++) for (int e = 0; e < 16; e++) acc[n][e] = 0;
    int m = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[k], b[k], acc[k % NACC], 0, 0, 0);
        if (READ) { m = max(m, acc[0][0]); for (int e = 0; e < 16; e++) acc[0][e] = 0; }
    }
    int s = m;
    for (int n = 0; n < NACC; n++) for (int e = 0; e < 16; e++) s += acc[n][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    void *in, *out; hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20); hipMalloc(&out, 64 << 20);
    const int iters = 4000;
    for (int data = 0; data < 2; data++) {
    if (data) {                                   // descriptor-like operands: bf16 integers 0..255 / bytes 0..255 (the pipe clocks lower on toggling data)
        static unsigned short h[1 << 19];
        unsigned r = 12345;
        for (int i = 0; i < (1 << 19); i++) { r = r * 1664525u + 1013904223u; float v = (float)((r >> 24) & 255); unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); }
        hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
    }
    printf("---- operands: %s\n", data ? "random 0..255" : "zeros");
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++) {
        const int grid = 256 * wgs_per_cu;
        const double flop = (double)grid * 8 * iters * 8 * 2.0 * 32 * 32 * 16;        // 8 waves x iters x 8 MFMAs
        float t;
        t = timeit([&] { hipLaunchKernelGGL((k_bf16<1, 0>), dim3(grid), dim3(512), 0, 0, (const bf16x8*)in, (float*)out, iters); });
        printf("bf16 32x32x16, %d WG/CU (%d waves/SIMD), 1 acc chain            : %7.0f TFLOP/s\n", wgs_per_cu, 2 * wgs_per_cu, flop / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((k_bf16<2, 0>), dim3(grid), dim3(512), 0, 0, (const bf16x8*)in, (float*)out, iters); });
        printf("bf16 32x32x16, %d WG/CU, 2 accs alternating                      : %7.0f TFLOP/s\n", wgs_per_cu, flop / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((k_bf16<1, 1>), dim3(grid), dim3(512), 0, 0, (const bf16x8*)in, (float*)out, iters); });
        printf("bf16 32x32x16, %d WG/CU, chain of 8 then read + reset acc         : %7.0f TFLOP/s\n", wgs_per_cu, flop / t / 1e9);
        const double iop = (double)grid * 8 * iters * 4 * 2.0 * 32 * 32 * 32;
        t = timeit([&] { hipLaunchKernelGGL((k_i8<1, 0>), dim3(grid), dim3(512), 0, 0, (const i32x4*)in, (int*)out, iters); });
        printf("i8   32x32x32, %d WG/CU, 1 acc chain                             : %7.0f TOP/s\n", wgs_per_cu, iop / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((k_i8<2, 0>), dim3(grid), dim3(512), 0, 0, (const i32x4*)in, (int*)out, iters); });
        printf("i8   32x32x32, %d WG/CU, 2 accs alternating                      : %7.0f TOP/s\n", wgs_per_cu, iop / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL((k_i8<1, 1>), dim3(grid), dim3(512), 0, 0, (const i32x4*)in, (int*)out, iters); });
        printf("i8   32x32x32, %d WG/CU, chain of 4 then read + reset acc         : %7.0f TOP/s\n", wgs_per_cu, iop / t / 1e9);
    }
    }
    return 0;
}
