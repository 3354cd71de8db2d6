// Micro-benchmark: issue cost (shader cycles per wave64 instruction per SIMD) of the instruction kinds the fp64
// collision sweep is made of.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(long long* out, int iters, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5, c = 1.0000001;
    int i0 = (int)seed, i1 = i0 + 1;
    unsigned long long s0 = 0, s1 = ~0ull;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b));) }
        if (KIND == 1) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 2) { REP8(asm volatile("v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4\n v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 3) { REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4\n v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (KIND == 4) { REP8(asm volatile("v_cmp_lt_f64 %0, %2, %3\n s_and_b64 %1, %1, %0\n v_cmp_lt_f64 %0, %3, %2\n s_and_b64 %1, %1, %0\n v_cmp_lt_f64 %0, %2, %3\n s_and_b64 %1, %1, %0\n v_cmp_lt_f64 %0, %3, %2\n s_and_b64 %1, %1, %0" : "+s"(s0), "+s"(s1) : "v"(a0), "v"(b) : "scc");) }
        if (KIND == 5) { REP8(asm volatile("v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, 0, %0, vcc\n v_cmp_lt_f64 vcc, %3, %2\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %0, 0, %0, vcc\n v_cmp_lt_f64 vcc, %3, %2\n v_cndmask_b32 %1, 0, %1, vcc" : "+v"(i0), "+v"(i1) : "v"(a0), "v"(b) : "vcc");) }
        if (KIND == 6) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(i0), "+v"(i1));) }
        if (KIND == 7) { REP8(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");) }
        if (KIND == 8) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
        if (KIND == 9) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 10) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(i0), "+v"(i1));) }
        if (KIND == 11) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0" : "+v"(i0), "+v"(i1));) }
    }
    long long t1 = clock64();
    if (a0 + a1 + a2 + a3 + (double)(i0 + i1) + (double)(s0 ^ s1) == 12345.678) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd)
{
    const int iters = 2000, ninst = iters * 64;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;     // one block per CU, 4 SIMDs per CU
    long long* d;
    hipMalloc(&d, sizeof(long long) * (2 + blocks * threads / 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(d, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, threads>>>(d, iters, 1.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 + blocks * threads / 64);
    hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (size_t i = 1; i < 1 + (size_t)blocks * threads / 64; ++i) avg += (double)h[i];
    avg /= (double)(blocks * threads / 64);
    // wall: ns per instruction per SIMD (all waves of a SIMD together)
    const double ns_per_inst_simd = (double)ms * 1e6 / ((double)ninst * waves_per_simd);
    printf("%-22s waves/SIMD=%d  wall %.3f ms  ns/inst/SIMD %.3f  clock64 ticks/inst/wave %.2f\n", name, waves_per_simd, ms, ns_per_inst_simd, avg / ninst);
    fflush(stdout);
    hipFree(d);
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : -1;
    for (int w : {1, 2, 4}) {
        switch (kind) {
            case 0: run<0>("v_fma_f64", w); break;
            case 1: run<1>("v_add_f64", w); break;
            case 2: run<2>("v_max_f64", w); break;
            case 3: run<3>("v_cmp_lt_f64->vcc", w); break;
            case 4: run<4>("v_cmp_f64+s_and dep", w); break;
            case 5: run<5>("v_cmp_f64+v_cndmask", w); break;
            case 6: run<6>("v_add_u32 dep", w); break;
            case 7: run<7>("s_and/s_or dep", w); break;
            case 8: run<8>("v_mul_f64", w); break;
            case 9: run<9>("v_rcp_f64", w); break;
            case 10: run<10>("v_cndmask_b32", w); break;
            case 11: run<11>("v_fma_f32 dep", w); break;
            default: printf("usage: valu_rates KIND(0..11)\n"); return 1;
        }
    }
    return 0;
}
