// Build and run (on the MI355X box): hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/issue scripts/ubench/issue.hip && scripts/ubench/issue
// Issue-rate microbenchmark (gfx950): how many wave-instructions per cycle does a CU issue for integer VALU, SALU and
// mixes of the two?  Each wave runs `iters` trips of an unrolled block of 64 instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t v0 = threadIdx.x, v1 = blockIdx.x, v2 = 3, v3 = 5;
    uint32_t s0 = blockIdx.x, s1 = 7, s2 = 11, s3 = 13;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // VALU only, 4 independent chains
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "scc");
        } else if (MODE == 1) {  // SALU only
            asm volatile(REP16("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 2) {  // 1:1 mix, independent
            asm volatile(REP16("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, %6\n")
                         REP16("v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %0\n s_add_u32 %7, %7, %4\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 3) {  // dependent VALU -> SGPR -> SALU -> VALU chain (v_cmp / s_and / v_cndmask)
            asm volatile(REP16("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %1, %2, vcc\n v_add_u32 %1, %1, %0\n")
                         : "+v"(v0), "+v"(v1), "+v"(v2) : : "vcc", "scc");
        } else if (MODE == 4) {  // v_readlane -> s_add -> v_add (scalar round trip)
            asm volatile(REP16("v_readlane_b32 %2, %0, 3\n s_add_u32 %2, %2, %3\n v_add_u32 %0, %2, %0\n v_add_u32 %1, %1, %0\n")
                         : "+v"(v0), "+v"(v1), "+s"(s0), "+s"(s1) : : "scc");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3;
}
template <int MODE>
void run(const char* name, uint32_t* d, int blocks_per_cu, int instr_per_trip) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winst = double(grid) * 4 * iters * instr_per_trip;  // wave-instructions
    printf("%-28s waves/SIMD %d  %.3f ms  %.2f wave-instr/ns  = %.2f per CU per cycle @2.4GHz\n", name, blocks_per_cu, ms,
           winst / (ms * 1e6), winst / (ms * 1e6) / 256 / 2.4);
    fflush(stdout);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("VALU only", d, w, 64);
        run<1>("SALU only", d, w, 64);
        run<2>("VALU+SALU 1:1", d, w, 128);
        run<3>("v_cmp->s_and->cndmask chain", d, w, 64);
        run<4>("readlane->s_add->v_add chain", d, w, 64);
    }
    return 0;
}
