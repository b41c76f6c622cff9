// scratch/valu_bench.hip -- issue cost of the VALU instructions the match epilogue could use (cycles per wave64 instruction per SIMD)
// hipcc --offload-arch=gfx950 -O3 -o scratch/valu_bench scratch/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, ASM) \
__global__ __launch_bounds__(512) void NAME(int* out, int iters, int seed) { \
    int r[8]; for (int i = 0; i < 8; i++) r[i] = seed + threadIdx.x * (i + 3); \
    int a = seed * 7 + 1, b = seed ^ 0x55; \
    for (int it = 0; it < iters; it++) { \
        _Pragma("unroll") for (int u = 0; u < 8; u++) { \
            asm volatile(ASM : "+v"(r[0]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r[1]) : "v"(a), "v"(b)); \
            asm volatile(ASM : "+v"(r[2]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r[3]) : "v"(a), "v"(b)); \
            asm volatile(ASM : "+v"(r[4]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r[5]) : "v"(a), "v"(b)); \
            asm volatile(ASM : "+v"(r[6]) : "v"(a), "v"(b)); asm volatile(ASM : "+v"(r[7]) : "v"(a), "v"(b)); \
        } \
    } \
    int s = 0; for (int i = 0; i < 8; i++) s += r[i]; out[blockIdx.x * 512 + threadIdx.x] = s; }

DEFK(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFK(k_max_f32, "v_max_f32 %0, %0, %1")
DEFK(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
DEFK(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
DEFK(k_max_i32, "v_max_i32 %0, %0, %1")
DEFK(k_max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEFK(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %0, 5, %1")
DEFK(k_add_u32, "v_add_u32 %0, %0, %1")
DEFK(k_lshl_or, "v_lshl_or_b32 %0, %0, 5, %1")
DEFK(k_or, "v_or_b32 %0, %0, %1")
DEFK(k_cvt, "v_cvt_f32_i32 %0, %0")
DEFK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFK(k_cmp, "v_cmp_gt_i32 vcc, %0, %1")
DEFK(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFK(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFK(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")

template <class K> void run(const char* name, K k, int* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, out, iters, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, out, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 512 WGs x 8 waves over 1024 SIMDs = 4 waves per SIMD, each iters x 64 instructions
    const double inst_per_simd = 4.0 * iters * 64;
    printf("%-16s %6.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms * 1e-3 * 2.4e9 / inst_per_simd);
}
int main() {
    int* out; hipMalloc(&out, 512 * 512 * 4);
#define RUN(K) run(#K, K, out);
    RUN(k_fma) RUN(k_max_f32) RUN(k_max3_f32) RUN(k_med3_f32) RUN(k_max_i32) RUN(k_max3_i32) RUN(k_med3_i32) RUN(k_lshl_add) RUN(k_add_u32)
    RUN(k_lshl_or) RUN(k_or) RUN(k_cvt) RUN(k_cndmask) RUN(k_cmp) RUN(k_add3) RUN(k_mad_i24) RUN(k_pk_max_i16)
    return 0;
}
