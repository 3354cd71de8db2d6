// v_cndmask_b32 on gfx950: profiles/r03_ubench_valu_classes.txt lists 9.5 ns per instruction and SIMD for it (every other VALU class: 1.1 or 1.8 ns).
// Which part of that is the instruction and which the way it was measured?  Forms: destination fed back / independent, vcc / SGPR-pair mask,
// constant operands, the mask freshly written by a v_cmp, and the replacements a kernel could use instead (v_bfi_b32 on a lane mask held in a
// VGPR, exec-masked v_mov, multiply by 0 / 1).
//   hipcc --offload-arch=gfx950 -O3 -o cndmask_rates cndmask_rates.hip && ./cndmask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CLOB "vcc", "s20", "s21", "s22", "s23", "s24", "s25"
template <int KIND>
__global__ void k_valu(long long* out, int iters, int seed)
{
    unsigned h0 = seed, h1 = seed + 1, h2 = seed + 2, h3 = seed + 3;
    unsigned a = seed * 7 + threadIdx.x, b = seed * 11 + threadIdx.x, m = (threadIdx.x & 1) ? 0xffffffffu : 0u;
    asm volatile("s_mov_b64 s[22:23], 0x55\n s_mov_b64 vcc, 0x55\n s_mov_b64 s[20:21], exec" ::: "s20", "s21", "s22", "s23", "vcc");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %5, vcc" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 1) { REP16(asm volatile("v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %5, %4, vcc\n v_cndmask_b32_e32 %2, %4, %5, vcc\n v_cndmask_b32_e32 %3, %5, %4, vcc" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 2) { REP16(asm volatile("v_cndmask_b32_e64 %0, %4, %5, s[22:23]\n v_cndmask_b32_e64 %1, %5, %4, s[22:23]\n v_cndmask_b32_e64 %2, %4, %5, s[22:23]\n v_cndmask_b32_e64 %3, %5, %4, s[22:23]" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 3) { REP16(asm volatile("v_cndmask_b32_e64 %0, 0, 1, s[22:23]\n v_cndmask_b32_e64 %1, 0, 1, s[22:23]\n v_cndmask_b32_e64 %2, 0, 1, s[22:23]\n v_cndmask_b32_e64 %3, 0, 1, s[22:23]" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 4) { REP16(asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cmp_lt_u32_e32 vcc, %1, %5\n v_cndmask_b32_e32 %1, %1, %5, vcc" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 5) { REP16(asm volatile("v_bfi_b32 %0, %6, %4, %0\n v_bfi_b32 %1, %6, %5, %1\n v_bfi_b32 %2, %6, %4, %2\n v_bfi_b32 %3, %6, %5, %3" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b), "v"(m) : CLOB);) }
        if (KIND == 6) { REP16(asm volatile("s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %0, %4\n s_mov_b64 exec, s[20:21]\n v_mov_b32_e32 %1, %5\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %2, %4\n s_mov_b64 exec, s[20:21]\n v_mov_b32_e32 %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 7) { REP16(asm volatile("v_max_u32_e32 %0, %0, %4\n v_max_u32_e32 %1, %1, %5\n v_max_u32_e32 %2, %2, %4\n v_max_u32_e32 %3, %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 8) { REP16(asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cmp_lt_u32_e32 vcc, %1, %5\n v_cmp_lt_u32_e32 vcc, %2, %4\n v_cmp_lt_u32_e32 vcc, %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 9) { REP16(asm volatile("v_and_b32_e32 %0, %6, %4\n v_and_b32_e32 %1, %6, %5\n v_and_b32_e32 %2, %6, %4\n v_and_b32_e32 %3, %6, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b), "v"(m) : CLOB);) }
        if (KIND == 10) { REP16(asm volatile("v_cndmask_b32_e64 %0, %4, %5, vcc\n v_cndmask_b32_e64 %1, %5, %4, vcc\n v_cndmask_b32_e64 %2, %4, %5, vcc\n v_cndmask_b32_e64 %3, %5, %4, vcc" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
        if (KIND == 11) { REP16(asm volatile("v_cndmask_b32_dpp %0, %4, %5, vcc row_shr:1 row_mask:0xf bank_mask:0xf\n v_or_b32_e32 %1, %1, %5\n v_or_b32_e32 %2, %2, %4\n v_or_b32_e32 %3, %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : CLOB);) }
    }
    if (h0 + h1 + h2 + h3 == 0x12345678u) out[0] = 1;
}
template <int KIND>
static void run(const char* name, int wps, double per_rep)
{
    const int blocks = 256, threads = 64 * 4 * wps, iters = 4000;
    long long* d;
    (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_valu<KIND><<<blocks, threads>>>(d, 10, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_valu<KIND><<<blocks, threads>>>(d, iters, 3);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD=%d  ns per VALU inst per SIMD %.3f\n", name, wps, (double)ms * 1e6 / (4000.0 * 16.0 * per_rep * wps));
    fflush(stdout);
    (void)hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_cndmask e32 vcc, destination fed back (r03's form)", w, 4);
        run<1>("v_cndmask e32 vcc, independent destination", w, 4);
        run<10>("v_cndmask e64 vcc, independent destination", w, 4);
        run<2>("v_cndmask e64 sgpr-pair mask, independent", w, 4);
        run<3>("v_cndmask e64 0, 1, sgpr-pair mask", w, 4);
        run<4>("v_cmp_lt_u32 -> vcc + v_cndmask on it (2 insts)", w, 4);
        run<8>("v_cmp_lt_u32 -> vcc alone", w, 4);
        run<5>("v_bfi_b32 with the lane mask in a VGPR", w, 4);
        run<6>("s_mov exec + v_mov_b32 (exec-masked move; 1 VALU each)", w, 4);
        run<7>("v_max_u32 e32", w, 4);
        run<9>("v_and_b32 e32 vgpr mask", w, 4);
    }
    return 0;
}
