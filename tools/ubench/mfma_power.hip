// What bf16 MFMA rate does the chip SUSTAIN under its power budget?  Pure v_mfma_f32_32x32x16_bf16 chains (4 independent
// accumulators per wave, 1 or 2 waves per SIMD, every CU), operands either constant (0x3c3c) or random bf16 bit patterns in
// [-2, 2), for `seconds` of back-to-back launches so that tools/power_probe.sh can sample rocm-smi (socket power, sclk) meanwhile.
//   ./mfma_power <random 0|1> <waves per SIMD 1|2> <seconds> [f16: 1 = v_mfma_f32_32x32x16_f16 on the same bit patterns]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16>
__global__ __launch_bounds__(256, 2) void k(const bf16x8 *g, float *out, int iters)
{
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a[6], b[10];
    for (int u = 0; u < 6; ++u) a[u] = g[threadIdx.x + 256 * u];
    for (int u = 0; u < 10; ++u) b[u] = g[threadIdx.x + 256 * (6 + u)];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 30; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (F16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + c) % 6]), __builtin_bit_cast(f16x8, b[u % 10]), acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) % 6], b[u % 10], acc[c], 0, 0, 0);
            }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
    const int random = argc > 1 ? atoi(argv[1]) : 1, waves = argc > 2 ? atoi(argv[2]) : 2;
    const double seconds = argc > 3 ? atof(argv[3]) : 4.0;
    const bool f16 = argc > 4 && atoi(argv[4]);
    std::vector<unsigned short> h(256 * 16 * 8);
    unsigned s = 12345u;
    for (auto &v : h) {
        s = s * 1664525u + 1013904223u;
        // sign, exponent 120..127 (|x| in [2^-7, 2)), random 7-bit mantissa: products stay finite over the run (zero-mean)
        if (f16) v = random ? (unsigned short)(((s >> 31) << 15) | ((8 + ((s >> 8) & 7)) << 10) | ((s >> 16) & 0x3ff)) : 0x3c3c;   // |x| in [2^-7, 2)
        else v = random ? (unsigned short)(((s >> 31) << 15) | ((120 + ((s >> 8) & 7)) << 7) | ((s >> 16) & 0x7f)) : 0x3c3c;
    }
    bf16x8 *g;
    float *o;
    hipMalloc(&g, h.size() * 2);
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&o, 512 * 256 * 4);
    const int grid = 256 * waves, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (f16) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, g, o, iters); else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, g, o, iters);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0;
    int n = 0;
    float last = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) if (f16) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, g, o, iters); else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, g, o, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        ms_sum += last;
        n += 20;
    }
    const double mfma = (double)grid * 4 * iters * 120;
    printf("%s random %d, %d wave/SIMD: %.3f ms per launch, %.0f TFLOP/s average, %.0f TFLOP/s last batch, %.2f GHz-equivalent MFMA rate per SIMD\n",
           f16 ? "f16" : "bf16", random, waves, ms_sum / n, mfma * 32768 * n / (ms_sum * 1e-3) / 1e12, mfma * 32768 * 20 / (last * 1e-3) / 1e12,
           mfma * 32 / 1024 * n / (ms_sum * 1e-3) / 1e9);
    return 0;
}
