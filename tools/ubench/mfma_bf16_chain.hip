// What does a DEPENDENT chain of v_mfma_f32_32x32x16_bf16 sustain per SIMD with 1 or 2 waves resident?
// (geometry of the split-bf16 MLP kernels: 256-thread workgroups, CHAINS independent accumulators per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(256, 2) void k(const bf16x8 *g, float *out, int iters)
{
    __shared__ float pad[12288];   // 48 KB: at most 3 workgroups per CU; grid decides 1 or 2 waves per SIMD
    pad[threadIdx.x] = 0;
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    // 20 distinct B operands + 6 distinct A operands, like the real kernel (hi/lo of 10 k16-steps, weight ring)
    bf16x8 bh[10], bl[10], aw[3][2];
    for (int u = 0; u < 10; ++u) { bh[u] = g[threadIdx.x + 64 * u]; bl[u] = g[threadIdx.x + 64 * u + 1024]; }
    for (int u = 0; u < 3; ++u) { aw[u][0] = g[threadIdx.x + 2048 + 64 * u]; aw[u][1] = g[threadIdx.x + 3072 + 64 * u]; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 10; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[u % 3][1], bh[u], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[u % 3][0], bl[u], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[u % 3][0], bh[u], acc[c], 0, 0, 0);
            }
    }
    float s = pad[threadIdx.x];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS> void run(const bf16x8 *g, float *o, int grid)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 400;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, g, o, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double n = (double)grid * 4 * iters * 30 * CHAINS;
    printf("grid %4d (%d wave/SIMD) chains %d: %.3f ms  %.0f TFLOP/s  %.1f cyc/MFMA/SIMD @2.4GHz\n", grid, grid / 256, CHAINS, ms,
           n * 32768 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n / 1024));
}

int main()
{
    bf16x8 *g;
    float *o;
    hipMalloc(&g, 1 << 20);
    hipMemset(g, 0x3c, 1 << 20);
    hipMalloc(&o, 1024 * 256 * 4);
    for (int grid : {256, 512}) {
        run<1>(g, o, grid);
        run<2>(g, o, grid);
        run<4>(g, o, grid);
    }
    return 0;
}
