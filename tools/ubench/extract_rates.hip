// Micro-benchmark for the pair kernel's inner loop (kernels_rdisc_mfma.hip): what does reading 1024 accumulator signs cost,
// per instruction kind, and how well do 4 x v_mfma_f32_32x32x8_f16 + 64 sign extractions overlap at 1 / 2 / 4 wavefronts per SIMD?
// Build: hipcc --offload-arch=gfx950 -O2 -o extract_rates extract_rates.hip     Run: ./extract_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// KIND 0..7: 64 VALU instructions of one kind per iteration, four independent chains
template <int KIND>
__global__ void k_valu(long long* out, int iters, int seed)
{
    unsigned h0 = seed, h1 = seed + 1, h2 = seed + 2, h3 = seed + 3;
    unsigned a = seed * 7 + threadIdx.x, b = seed * 11 + threadIdx.x;
    float fa = (float)seed, fb = (float)(seed + 1);
    unsigned long long s0 = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %5, 31" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 1) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 2) { REP16(asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %5, %4\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %5, %4" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 3) { REP16(asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %5, %4\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %5, %4" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(fa), "v"(fb));) }
        if (KIND == 4) { REP16(asm volatile("v_cmp_gt_f32 vcc, 0, %0\n v_cmp_gt_f32 vcc, 0, %1\n v_cmp_gt_f32 vcc, 0, %2\n v_cmp_gt_f32 vcc, 0, %3" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : : "vcc");) }
        if (KIND == 5) { REP16(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %4\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %5, %4" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 6) { REP16(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 7) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %4\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %5, %4" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 8) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b));) }
        if (KIND == 9) { REP16(asm volatile("v_readlane_b32 s4, %0, 3\n v_readlane_b32 s5, %1, 4\n v_readlane_b32 s6, %2, 5\n v_readlane_b32 s7, %3, 6" : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : : "s4", "s5", "s6", "s7");) }
    }
    long long t1 = clock64();
    if (h0 + h1 + h2 + h3 + (unsigned)s0 == 0x12345678u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)] = t1 - t0;
}

// one "chunk" of the pair kernel without memory: 4 MFMAs (2 A fragments x 2 B fragments), the 64 alignbits that read their
// accumulators, H assembly.  PIPE = 0: as the kernel does it (extraction follows its own MFMAs);  PIPE = 1: software pipelined --
// the MFMAs of chunk k+1 are issued before chunk k's accumulators are read (two accumulator sets, 128 VGPRs of accumulators)
template <int PIPE>
__global__ void k_chunk(long long* out, int iters, float seed)
{
    half4 a0, a1, b0, b1;
    for (int i = 0; i < 4; ++i) { a0[i] = (_Float16)(seed + i); a1[i] = (_Float16)(seed - i); b0[i] = (_Float16)(0.5f * i); b1[i] = (_Float16)(0.25f * i + threadIdx.x); }
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = seed * i; c1[i] = -seed * i; }
    unsigned long long acc = 0;
    auto extract = [&](const f32x16& x0, const f32x16& x1, const f32x16& x2, const f32x16& x3) {
        unsigned h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h0 = __builtin_amdgcn_alignbit(h0, __float_as_uint(x0[r]), 31);
            h1 = __builtin_amdgcn_alignbit(h1, __float_as_uint(x1[r]), 31);
            h2 = __builtin_amdgcn_alignbit(h2, __float_as_uint(x2[r]), 31);
            h3 = __builtin_amdgcn_alignbit(h3, __float_as_uint(x3[r]), 31);
        }
        acc += (unsigned long long)(h0 | (h1 << 16)) | ((unsigned long long)(h2 | (h3 << 16)) << 32);
    };
    long long t0 = clock64();
    if (PIPE == 0) {
        for (int it = 0; it < iters; ++it) {
            b0[0] = (_Float16)(float)it;
            f32x16 x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c0, 0, 0, 0);
            f32x16 x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b0, c1, 0, 0, 0);
            f32x16 x2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b1, c0, 0, 0, 0);
            f32x16 x3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, c1, 0, 0, 0);
            extract(x0, x1, x2, x3);
        }
    } else {
        f32x16 x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c0, 0, 0, 0);
        f32x16 x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b0, c1, 0, 0, 0);
        f32x16 x2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b1, c0, 0, 0, 0);
        f32x16 x3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, c1, 0, 0, 0);
        for (int it = 0; it < iters; it += 2) {
            b0[0] = (_Float16)(float)it;
            f32x16 y0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c0, 0, 0, 0);
            f32x16 y1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b0, c1, 0, 0, 0);
            f32x16 y2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b1, c0, 0, 0, 0);
            f32x16 y3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, c1, 0, 0, 0);
            extract(x0, x1, x2, x3);
            b1[0] = (_Float16)(float)it;
            x0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b0, c1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b1, c0, 0, 0, 0);
            x3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, c1, 0, 0, 0);
            extract(y0, y1, y2, y3);
        }
        extract(x0, x1, x2, x3);
    }
    long long t1 = clock64();
    if (acc == 0x1234567812345678ull) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename F>
static void time_it(const char* name, int wps, int iters, double units_per_iter, const char* unit, F launch)
{
    const int blocks = 256, threads = 64 * 4 * wps;
    long long* d;
    hipMalloc(&d, sizeof(long long) * (2 + blocks * threads / 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, threads, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, threads, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = (double)ms * 1e6 / ((double)iters * units_per_iter * wps);
    printf("%-34s waves/SIMD=%d  wall %.3f ms  ns per %s per SIMD %.3f\n", name, wps, ms, unit, ns);
    fflush(stdout);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
#define V(K, NAME) time_it(NAME, w, 4000, 64.0, "inst", [](int b, int t, long long* d, int it) { k_valu<K><<<b, t>>>(d, it, 3); });
        V(0, "v_alignbit_b32") V(1, "v_lshl_or_b32") V(2, "v_or3_b32") V(3, "v_min3_f32") V(4, "v_cmp_gt_f32 -> vcc") V(5, "v_and_or_b32")
        V(6, "v_or_b32 (VOP2)") V(7, "v_perm_b32") V(8, "v_add_u32 (VOP2)") V(9, "v_readlane_b32")
#undef V
    }
    for (int w : {1, 2, 3, 4}) {
        time_it("chunk: 4 mfma32x32x8 + 64 alignbit", w, 20000, 1.0, "chunk", [](int b, int t, long long* d, int it) { k_chunk<0><<<b, t>>>(d, it, 1.5f); });
        if (w <= 2) time_it("chunk, software pipelined", w, 20000, 1.0, "chunk", [](int b, int t, long long* d, int it) { k_chunk<1><<<b, t>>>(d, it, 1.5f); });
    }
    return 0;
}
