// How fast can hits be appended to per-column lists through returning global atomics?  n appends, columns drawn at random
// from ncol counters (variant 1) or in runs of 64 appends hitting 64 nearby columns (variant 2, the pattern of a refine
// batch whose survivors belong to one candidate chunk).  Prints ms and appends/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
struct hit { int j, pad; double d; };
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_append(int* cnt, hit* pool, int cap, int ncol, long long n, int variant)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t col;
    if (variant == 1) col = mix((uint32_t)i) % (uint32_t)ncol;
    else { const uint32_t base = mix((uint32_t)(i >> 6)) % (uint32_t)(ncol - 64); col = base + (mix((uint32_t)i) & 63u); }
    const int slot = atomicAdd(&cnt[col], 1);
    if (slot < cap) { hit h; h.j = (int)i; h.pad = slot; h.d = 1.0; *reinterpret_cast<uint4*>(&pool[(long long)col * cap + slot]) = *reinterpret_cast<uint4*>(&h); }
}
int main()
{
    const int ncol = 1000000, cap = 160; const long long n = 54000000;
    int* cnt; hit* pool;
    hipMalloc(&cnt, sizeof(int) * ncol); hipMalloc(&pool, sizeof(hit) * (size_t)ncol * cap);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 1; variant <= 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cnt, 0, sizeof(int) * ncol);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_append, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, cnt, pool, cap, ncol, n, variant);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("variant %d: %.3f ms for %lld appends = %.3g appends/s\n", variant, ms, n, n / (ms * 1e-3));
        }
    }
    return 0;
}
