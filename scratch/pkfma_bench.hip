// issue rate of v_pk_fma_f32 forms on gfx950: cycles per wave64 instruction, 1 and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, v2f kk, long long* cyc) {
    v2f a[16];
    v2f x = {out[threadIdx.x], out[threadIdx.x + 1]}, kv = {out[threadIdx.x + 2], out[threadIdx.x + 3]};
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (v2f){(float)i, 1.0f};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(kv), "v"(x));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(kk), "v"(x));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(kv), "v"(x));
            if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(kv.x), "v"(x.x));
            if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(kk), "v"(x));
            if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(a[i]) : "v"(kv), "v"(x));
            if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "s"(kk.x), "v"(x.x));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[threadIdx.x + blockIdx.x * blockDim.x + 8] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char* name, float* d, long long* dc) {
    for (int threads : {256, 512, 1024, 2048}) {
        const int grid = threads > 1024 ? 512 : 256, bt = threads > 1024 ? 1024 : threads;
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(bt), 0, 0, d, 10, (v2f){1.0f, 0.5f}, dc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(bt), 0, 0, d, iters, (v2f){1.0f, 0.5f}, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        // per SIMD: waves = threads / 256; instructions per wave = iters * 16
        printf("%-42s waves/SIMD %d: %.3f ms  -> %.2f ns per instr per SIMD-wave slot, cyclecounter %.2f per instr\n", name, threads / 256, ms,
               ms * 1e6 / ((double)iters * 16 * (threads / 256)), (double)c / ((double)iters * 16));
    }
}
int main() {
    float* d; long long* dc; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22); hipMalloc(&dc, 8);
    run<0>("pk_fma vgpr,vgpr", d, dc);
    run<1>("pk_fma sgpr(op_sel_hi bcast lo),vgpr", d, dc);
    run<2>("pk_fma vgpr(op_sel_hi bcast),vgpr", d, dc);
    run<4>("pk_fma sgpr,vgpr (no op_sel)", d, dc);
    run<3>("fma vgpr,vgpr", d, dc);
    run<6>("fma sgpr,vgpr", d, dc);
    run<5>("pk_mul vgpr,vgpr", d, dc);
    return 0;
}
