// which CUs a hipExtStreamCreateWithCUMask stream runs on: mask bit -> (XCD, SE, CU).  hipcc --offload-arch=gfx950 scratch/cumask_probe.hip -o /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void k(unsigned* out) {
    unsigned v, x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = v; out[2 * blockIdx.x + 1] = x; }
    for (int i = 0; i < 2000; i++) asm volatile("s_sleep 10");
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32;
    printf("CUs %d\n", ncu);
    unsigned* d; (void)hipMalloc(&d, 8 * 8192);
    auto run = [&](const char* name, std::vector<uint32_t> m) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
        (void)hipMemsetAsync(d, 0xff, 8 * 8192, s);
        hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, s, d);
        (void)hipStreamSynchronize(s);
        std::vector<unsigned> h(2 * 4096); (void)hipMemcpy(h.data(), d, 8 * 4096, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; int perx[8] = {0};
        std::set<unsigned> perxset[8];
        for (int b = 0; b < 4096; b++) { const unsigned v = h[2 * b], x = h[2 * b + 1] & 15; const unsigned cu = (v >> 8) & 15, se = (v >> 13) & 7; cus.insert((x << 8) | (se << 4) | cu); if (x < 8) perxset[x].insert((se << 4) | cu); }
        for (int x = 0; x < 8; x++) perx[x] = (int)perxset[x].size();
        printf("%-28s distinct (xcd,se,cu): %3d | per XCD:", name, (int)cus.size());
        for (int x = 0; x < 8; x++) printf(" %d", perx[x]);
        printf(" | xcd0 (se,cu):");
        for (unsigned q : perxset[0]) printf(" %u.%u", q >> 4, q & 15);
        printf("\n");
        (void)hipStreamDestroy(s);
    };
    std::vector<uint32_t> all(nw, 0xffffffffu);
    run("all", all);
    for (int nb : {8, 16, 32, 64, 128}) { std::vector<uint32_t> m(nw, 0); for (int b = 0; b < nb; b++) m[b >> 5] |= 1u << (b & 31); char nm[64]; snprintf(nm, 64, "low %d bits", nb); run(nm, m); }
    { std::vector<uint32_t> m(nw, 0); for (int b = 32; b < ncu; b++) m[b >> 5] |= 1u << (b & 31); run("all but low 32", m); }
    { std::vector<uint32_t> m(nw, 0); for (int b = 0; b < ncu; b += 8) m[b >> 5] |= 1u << (b & 31); run("every 8th bit", m); }
    return 0;
}
