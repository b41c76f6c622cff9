#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read6(const v4f* a, const v4f* b, const v4f* c, const v4f* d, const v4f* e, const v4f* f, size_t n4, float* out) {
    v4f acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += a[i] + b[i] + c[i] + d[i] + e[i] + f[i];
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = 1;
}
__global__ __launch_bounds__(256) void copy1(const v4f* a, v4f* b, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
    const size_t px = 8000ull * 6000, n4 = px / 4;
    float* p[7]; for (int i = 0; i < 7; i++) { hipMalloc(&p[i], px * 4); hipMemset(p[i], 0, px * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 46875}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); for (int it = 0; it < 5; it++) hipLaunchKernelGGL(read6, dim3(grid), dim3(256), 0, 0, (v4f*)p[0], (v4f*)p[1], (v4f*)p[2], (v4f*)p[3], (v4f*)p[4], (v4f*)p[5], n4, p[6]);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            if (rep) printf("read6 grid %d: %.1f us %.0f GB/s\n", grid, ms * 1e3, px * 24.0 / ms / 1e6);
            hipEventRecord(e0); for (int it = 0; it < 5; it++) hipLaunchKernelGGL(copy1, dim3(grid), dim3(256), 0, 0, (v4f*)p[0], (v4f*)p[1], n4);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            if (rep) printf("copy grid %d: %.1f us %.0f GB/s\n", grid, ms * 1e3, px * 8.0 / ms / 1e6);
        }
    }
    return 0;
}
