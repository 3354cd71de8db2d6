// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the two access patterns of this library (the guide calibrates it
// only for wide coalesced streaming reads, where it reports 1/2 of the bytes):
//   k_stream : every lane reads 16 B, consecutive lanes consecutive addresses (the MFMA operand stream)
//   k_gather : every lane reads one 48-byte row (3 x 16 B) at a pseudo-random row index of a 48 MB array (the row-state
//              gathers of the sweep / the refine) -- 1.5 64-byte sectors per row on average when rows straddle sectors.
// Build: hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_stream(const uint4* __restrict__ a, size_t n, uint4* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678u && acc.y == 1u) out[0] = acc;
}

__global__ void k_gather(const double* __restrict__ X, uint32_t nrows, size_t ngather, double* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; i < ngather; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const size_t y = h % nrows;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc += X[y * 6 + q];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

int main()
{
    const size_t stream_bytes = (size_t)4 << 30;                 // 4 GiB, read once
    const uint32_t nrows = 1000000;                              // 48 MB of 48-byte rows
    const size_t ngather = 100000000;                            // 1e8 row gathers
    uint4* a; double* X; void* out;
    hipMalloc(&a, stream_bytes); hipMalloc(&X, (size_t)nrows * 48); hipMalloc(&out, 64);
    hipMemset(a, 1, stream_bytes); hipMemset(X, 0, (size_t)nrows * 48);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    hipEventRecord(e0);
    k_stream<<<256 * 8, 256>>>(a, stream_bytes / 16, (uint4*)out);
    hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    printf("k_stream: %.3f GiB requested, %.3f ms -> %.2f TB/s\n", stream_bytes / 1073741824.0, ms, stream_bytes / ms / 1e9);
    hipEventRecord(e0);
    k_gather<<<256 * 12, 256>>>(X, nrows, ngather, (double*)out);
    hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    printf("k_gather: %.3e rows x 48 B = %.3f GiB of rows, 1.5 sectors/row = %.3f GiB of 64-byte sectors, %.3f ms -> %.3g rows/s\n",
           (double)ngather, ngather * 48.0 / 1073741824.0, ngather * 96.0 / 1073741824.0, ms, ngather / (ms * 1e-3));
    return 0;
}
