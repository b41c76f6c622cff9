#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = v;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 4096 * 64); (void)hipMemset(d, 0, 4096 * 64);
    for (int threads : {512, 768}) {
        hipLaunchKernelGGL(k, dim3(600), dim3(threads), 0, 0, d);
        unsigned h[600 * 16]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("block of %d threads: SIMD id per wave (wave index order), first 12 blocks\n", threads);
        for (int b = 0; b < 12; b++) { for (int w = 0; w < threads / 64; w++) printf("%u ", (h[b * 16 + w] >> 4) & 3); printf(" | cu %u se %u\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7); }
        int hist[16] = {0};
        for (int b = 0; b < 600; b++) { int ok = 1; for (int w = 0; w < threads / 64; w++) if (((h[b * 16 + w] >> 4) & 3) != (unsigned)(w & 3)) ok = 0; hist[ok]++; }
        printf("blocks with simd == wave %% 4 for every wave: %d of 600\n", hist[1]);
    }
    return 0;
}
