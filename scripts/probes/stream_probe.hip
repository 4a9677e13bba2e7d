// What the memory system gives a kernel with k_front's traffic and nothing else to do: 16 bytes in per record (8-byte
// key, two 4-byte words), 8 bytes out for 5 records in 8 (densely) -- one pass, wide loads, no arithmetic to speak of --
// beside the read alone and a plain copy.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe scripts/probes/stream_probe.hip && /tmp/stream_probe [n_records]
// (a measuring aid, not part of the product or its tests)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ key2, const uint2* __restrict__ ref2,
                                                const uint2* __restrict__ pos2, uint2* __restrict__ o1, uint2* __restrict__ o2,
                                                uint32_t n2) {
    // one thread: two records = one 16-byte key load + one 8-byte load of each word array; every load instruction of a
    // wave covers consecutive bytes
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n2; i += gridDim.x * 256u) {
        const uint4 k = key2[i];
        const uint2 r = ref2[i], p = pos2[i];
        if (i < n2 / 8u * 5u) {  // 62.5 % of the records leave a target; the targets lie densely, as the front end's do
            o1[i] = make_uint2(k.x ^ r.x, k.z ^ r.y);
            o2[i] = make_uint2(k.y + p.x, k.w + p.y);
        } else {
            acc += k.x ^ k.y ^ k.z ^ k.w ^ r.x ^ r.y ^ p.x ^ p.y;  // (every record is read)
        }
    }
    if (acc == 0x12345678u) o1[0] = make_uint2(acc, acc);
}

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ key2, const uint2* __restrict__ ref2,
                                              const uint2* __restrict__ pos2, uint32_t* __restrict__ out, uint32_t n2) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n2; i += gridDim.x * 256u) {
        const uint4 k = key2[i];
        const uint2 r = ref2[i], p = pos2[i];
        acc += k.x ^ k.y ^ k.z ^ k.w ^ r.x ^ r.y ^ p.x ^ p.y;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) out[i] = in[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const uint32_t n2 = static_cast<uint32_t>(n / 2);
    void *key, *ref, *pos, *o1, *o2;
    CK(hipMalloc(&key, n * 8)); CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&pos, n * 4)); CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4));
    CK(hipMemset(key, 1, n * 8)); CK(hipMemset(ref, 2, n * 4)); CK(hipMemset(pos, 3, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t grid : {2048u, 4096u, 16384u}) {
        float best = 1e30f;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const uint4*)key, (const uint2*)ref, (const uint2*)pos,
                               (uint2*)o1, (uint2*)o2, n2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        const double bytes = 16.0 * n + 8.0 * n * 5 / 8;
        printf("k_front's traffic, grid %5u: %.3f ms  %.0f GB/s  (%.1f %% of 8 TB/s)\n", grid, best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
    }
    for (uint32_t grid : {4096u, 16384u}) {
        float best = 1e30f;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const uint4*)key, (const uint2*)ref, (const uint2*)pos,
                               (uint32_t*)o1, n2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("the 16 B/record read alone, grid %5u: %.3f ms  %.0f GB/s\n", grid, best, 16.0 * n / best / 1e6);
    }
    {
        float best = 1e30f;
        const uint32_t nn = static_cast<uint32_t>(n / 4);  // 4 n bytes in, 4 n bytes out (o1 holds 4 n)
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4*)key, (uint4*)o1, nn);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("plain copy (read + write)  : %.3f ms  %.0f GB/s\n", best, 8.0 * n / best / 1e6);
    }
    return 0;
}
