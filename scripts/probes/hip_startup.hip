// What the first HIP calls of a process cost (the `slimm` command's start-up): hipcc --offload-arch=gfx950 -O2 -o hip_startup hip_startup.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int* p) { p[threadIdx.x] = threadIdx.x; }
int main(int argc, char** argv) {
    const bool parallel = argc > 1;
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    double t1 = now();
    printf("hipSetDevice + hipFree(0): %.1f ms\n", t1 - t0);
    hipStream_t s[4];
    if (parallel) {
        std::vector<std::thread> th;
        for (int i = 0; i < 4; ++i) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); });
        for (auto& t : th) t.join();
        printf("4 streams created by 4 threads: %.1f ms\n", now() - t1);
    } else {
        for (int i = 0; i < 4; ++i) {
            double a = now();
            hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            printf("stream %d: %.1f ms\n", i, now() - a);
        }
    }
    double t2 = now();
    int* d;
    hipMalloc(&d, 1 << 20);
    double t3 = now();
    printf("first hipMalloc: %.1f ms\n", t3 - t2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0], d);
    hipStreamSynchronize(s[0]);
    double t4 = now();
    printf("first launch + sync on stream 0: %.1f ms\n", t4 - t3);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[1], d);
    hipStreamSynchronize(s[1]);
    printf("first launch + sync on stream 1: %.1f ms\n", now() - t4);
    double t5 = now();
    void* h;
    hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    printf("hipHostMalloc 64 MB: %.1f ms\n", now() - t5);
    double t6 = now();
    for (int i = 0; i < 4; ++i) hipStreamDestroy(s[i]);
    for (int i = 0; i < 4; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    printf("destroy + create 4 streams again: %.1f ms\n", now() - t6);
    printf("total %.1f ms\n", now() - t0);
    return 0;
}
