// hbm_bw.hip — what HBM bandwidth does one MI355X deliver to a READ-ONLY stream, a WRITE-ONLY stream and a copy?
// (The guide's 6.29 TB/s is a float4 copy: read + write summed.  The segment reduce is 88 % reads.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_bw tools/ubench/hbm_bw.hip && /tmp/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const float4 *__restrict__ a, float *out, size_t n4, int unroll_dummy)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * 4) {     // 4 independent 16-B loads in flight per lane
        float4 v0 = a[i], v1 = i + stride < n4 ? a[i + stride] : acc, v2 = i + 2 * stride < n4 ? a[i + 2 * stride] : acc,
               v3 = i + 3 * stride < n4 ? a[i + 3 * stride] : acc;
        acc.x += v0.x + v1.x + v2.x + v3.x; acc.y += v0.y + v1.y + v2.y + v3.y; acc.z += v0.z + v1.z + v2.z + v3.z; acc.w += v0.w + v1.w + v2.w + v3.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;      // never true: keeps the loads
}
__global__ __launch_bounds__(256) void write_kernel(float4 *__restrict__ a, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) a[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
// 7 : 1 read : write mix, like the segment reduce (reads 1.12 GB, writes 0.16 GB)
__global__ __launch_bounds__(256) void mix_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 7 * stride < n4; i += stride * 8) {
        float4 s = a[i];
#pragma unroll
        for (int k = 1; k < 7; ++k) { const float4 v = a[i + k * stride]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        b[i / 8] = s;
    }
}

int main()
{
    const size_t bytes = 4ull << 30, n4 = bytes / 16;      // 4 GiB per buffer: far beyond the 256 MB Infinity Cache
    float4 *a, *b; float *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * wg_per_cu;
        auto time = [&](auto &&launch, double moved, const char *name) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-10s %2d WG/CU: %7.3f ms  %6.2f TB/s\n", name, wg_per_cu, ms / 5, moved / (ms / 5 * 1e-3) / 1e12);
        };
        time([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, out, n4, 0); }, (double)bytes, "read");
        time([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, a, n4); }, (double)bytes, "write");
        time([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n4); }, 2.0 * bytes, "copy r+w");
        time([&] { hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(256), 0, 0, a, b, n4); }, (double)bytes * (7.0 / 8 + 1.0 / 8 / 1.0) , "mix 7r:1w");
    }
    return 0;
}
