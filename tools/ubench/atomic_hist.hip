// Returning device-scope atomics of a histogram pass (k_cellkey_count: 1e6 samples -> 15 625 cell counters, the returning atomic is the
// sample's arrival number): how does the rate depend on where the counters lie?  n atomics by n threads, counter = hash(thread) mod K,
// counter k at byte offset k * stride; returning and non-returning.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_hist atomic_hist.hip && ./atomic_hist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint32_t mixh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool RET>
__global__ void k_hist(int* cnt, int n, int K, int stride_ints, int* out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int k = (int)(mixh((uint32_t)p) % (uint32_t)K);
    if (RET) out[p] = atomicAdd(&cnt[(size_t)k * stride_ints], 1);
    else atomicAdd(&cnt[(size_t)k * stride_ints], 1);
}
int main()
{
    const int n = 1000000;
    int *cnt, *out;
    (void)hipMalloc(&cnt, (size_t)64 << 20);
    (void)hipMalloc(&out, sizeof(int) * n);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int K : {15625, 27000, 125000, 1000000}) for (int stride : {1, 4, 16, 64, 256}) {
        if ((size_t)K * stride * 4 > ((size_t)64 << 20)) continue;
        for (int ret = 1; ret >= 0; --ret) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipMemset(cnt, 0, (size_t)K * stride * 4);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0);
                if (ret) k_hist<true><<<(n + 255) / 256, 256>>>(cnt, n, K, stride, out);
                else k_hist<false><<<(n + 255) / 256, 256>>>(cnt, n, K, stride, out);
                (void)hipEventRecord(e1);
                (void)hipDeviceSynchronize();
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("K %7d stride %4d B %s  %7.1f us  (%.1f atomics / ns)\n", K, stride * 4, ret ? "returning" : "no return", best * 1e3, n / (best * 1e6));
        }
    }
    return 0;
}
