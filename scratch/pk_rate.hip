// scratch/pk_rate.hip -- issue rate and dependent latency of the packed f32 instructions blur16_stream is made of, in SHADER cycles
// (s_memtime) and the shader clock they imply (against s_memrealtime, 100 MHz), at 1 .. 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/pk_rate scratch/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: 8 independent pk_mul            1: 8 independent pk_add           2: one dependent pk_add chain
//      3: mul (independent) + add chain, ONE chain (the row pass of one output pair), no nop
//      4: two interleaved chains          5: as 3 with an s_nop 0 before every add (what hipcc emits)
//      6: 8 independent v_mul_f32 (unpacked)   7: dependent v_add_f32 chain   8: unpacked mul + add chain, two chains (4 px as 4 scalar chains -> here 2)
//      9: 8 independent pk_fma            10: column-pass pattern: pk_add (indep) -> pk_mul (dep) -> pk_add (dep chain), two chains interleaved
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, v2f kk, unsigned long long* stamps) {
    v2f a[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (v2f){out[threadIdx.x + i], 1.0f}; x[i] = (v2f){out[threadIdx.x + 8 + i], out[threadIdx.x + 16 + i]}; }
    v2f p, q;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(a[i]) : "s"(kk), "v"(x[i])); }
            if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(x[(i + 1) & 7]), "v"(x[i])); }
            if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(x[i])); }
            if (MODE == 3) { _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %1" : "+v"(a[0]), "=&v"(p) : "s"(kk), "v"(x[i])); }
            if (MODE == 4) { _Pragma("unroll") for (int i = 0; i < 2; i++) asm volatile("v_pk_mul_f32 %2, %4, %5 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %4, %6 op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3" : "+v"(a[0]), "+v"(a[1]), "=&v"(p), "=&v"(q) : "s"(kk), "v"(x[i]), "v"(x[i + 2])); }
            if (MODE == 5) { _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n s_nop 0\n v_pk_add_f32 %0, %0, %1" : "+v"(a[0]), "=&v"(p) : "s"(kk), "v"(x[i])); }
            if (MODE == 6) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i].x) : "s"(kk.x), "v"(x[i].x)); }
            if (MODE == 7) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(x[i].x)); }
            if (MODE == 8) { _Pragma("unroll") for (int i = 0; i < 2; i++) asm volatile("v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %6\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a[0].x), "+v"(a[1].x), "=&v"(p.x), "=&v"(q.x) : "s"(kk.x), "v"(x[i].x), "v"(x[i + 2].x)); }
            if (MODE == 9) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(kk), "v"(x[i])); }
            if (MODE == 10) { _Pragma("unroll") for (int i = 0; i < 2; i++) asm volatile("v_pk_add_f32 %2, %5, %6\n v_pk_add_f32 %3, %6, %7\n v_pk_mul_f32 %2, %4, %2 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %4, %3 op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3"
                : "+v"(a[0]), "+v"(a[1]), "=&v"(p), "=&v"(q) : "s"(kk), "v"(x[i]), "v"(x[i + 2]), "v"(x[i + 4])); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[threadIdx.x + blockIdx.x * blockDim.x + 64] = s + p.x + q.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}
static const int INSTR_PER_U[11] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 12};   // vector instructions per inner unit (nops not counted)
template <int MODE> void run(const char* name, float* d, unsigned long long* ds) {
    printf("%-58s", name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int threads = 256 * wps;          // per CU
        const int bt = threads > 512 ? 512 : threads, bpc = threads / bt;
        const int grid = 256 * bpc, iters = 4000;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(bt), 0, 0, d, 10, (v2f){1.0f, 0.5f}, ds);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(bt), 0, 0, d, iters, (v2f){1.0f, 0.5f}, ds);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[2]; CK(hipMemcpy(st, ds, 16, hipMemcpyDeviceToHost));
        const double n = (double)iters * 4 * INSTR_PER_U[MODE];
        const double ghz = (double)st[0] / ((double)st[1] * 10.0);       // shader cycles per 10 ns tick of the 100 MHz counter
        // cycles per instruction per SIMD = cycles * 1 / (n * waves on the SIMD)
        printf(" | w%d %5.2f/%5.2f %4.2f", wps, (double)st[0] / n, (double)ms * 1e-3 * ghz * 1e9 / (n * wps), ghz);
    }
    printf("\n");
}
int main() {
    float* d; unsigned long long* ds; CK(hipMalloc(&d, 1 << 24)); CK(hipMemset(d, 0, 1 << 24)); CK(hipMalloc(&ds, 16));
    { std::vector<float> h(1 << 22); for (size_t i = 0; i < h.size(); i++) h[i] = (float)(rand() % 12240) + 0.37f; CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    printf("per entry: shader cycles per instruction of ONE wave (its own s_memtime) / kernel wall time x clock / (instructions per wave x waves per SIMD) = cycles per instruction per SIMD / shader clock GHz\n");
    run<0>("0 pk_mul independent", d, ds);
    run<1>("1 pk_add independent", d, ds);
    run<2>("2 pk_add dependent chain", d, ds);
    run<3>("3 pk_mul + dependent pk_add, one chain", d, ds);
    run<5>("5 as 3 with s_nop 0 before the add", d, ds);
    run<4>("4 two such chains interleaved", d, ds);
    run<10>("10 column pattern add,mul,add two chains interleaved", d, ds);
    run<9>("9 pk_fma independent", d, ds);
    run<6>("6 v_mul_f32 independent", d, ds);
    run<7>("7 v_add_f32 dependent chain", d, ds);
    run<8>("8 v_mul + v_add two chains", d, ds);
    return 0;
}
