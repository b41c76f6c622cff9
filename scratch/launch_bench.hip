#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { float k[300]; };
__global__ void empty_k(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void big_k(Big b, int* p) { if (p && threadIdx.x == 12345) *p = (int)b.k[3]; }
int main() {
    hipStream_t s[3]; for (int i = 0; i < 3; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int* d; hipMalloc(&d, 4);
    Big b; for (int i = 0; i < 300; i++) b.k[i] = i;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int rep = 0; rep < 2; rep++) {
        hipDeviceSynchronize();
        auto t0 = now();
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s[0], d);
        auto t1 = now(); hipDeviceSynchronize(); auto t2 = now();
        printf("empty x2000 one stream: enqueue %.2f us/launch, total %.2f us/launch\n", us(t0, t1) / 2000, us(t0, t2) / 2000);
        t0 = now();
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(big_k, dim3(1), dim3(64), 0, s[0], b, d);
        t1 = now(); hipDeviceSynchronize(); t2 = now();
        printf("1.2KB-arg x2000: enqueue %.2f us/launch, total %.2f\n", us(t0, t1) / 2000, us(t0, t2) / 2000);
        t0 = now();
        for (int i = 0; i < 2000; i++) { hipEventRecord(e0, s[0]); hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s[0], d); hipEventRecord(e1, s[0]); }
        t1 = now(); hipDeviceSynchronize(); t2 = now();
        printf("event+launch+event x2000: enqueue %.2f us/iter, total %.2f\n", us(t0, t1) / 2000, us(t0, t2) / 2000);
        t0 = now();
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s[i % 3], d);
        t1 = now(); hipDeviceSynchronize(); t2 = now();
        printf("empty x2000 round-robin 3 streams: enqueue %.2f us/launch, total %.2f\n", us(t0, t1) / 2000, us(t0, t2) / 2000);
        t0 = now();
        for (int i = 0; i < 200; i++) hipMemsetAsync(d, 0, 4, s[0]);
        t1 = now(); hipDeviceSynchronize(); t2 = now();
        printf("memsetAsync x200: enqueue %.2f us, total %.2f\n", us(t0, t1) / 200, us(t0, t2) / 200);
    }
    return 0;
}
