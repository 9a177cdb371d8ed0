// mx_lone.hip — what one wave per SIMD (the weight-stationary edge encoder's shape: 256-thread workgroups, one per CU) pays per matrix instruction:
// s_memtime cycles per v_mfma_f32_32x32x16_f16 and per v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x bf8, and fp6 / fp4 for comparison), independent
// accumulator chains (4 in rotation), and for the edge stack's triple (f16, f16, scaled) on one accumulator with two accumulators alternating.
//   hipcc --offload-arch=gfx950 -O3 -o mx_lone tools/ubench/mx_lone.hip && ./mx_lone        (grid 1 = a lone workgroup; grid 256 = every CU busy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// VALU filler: N dependent fma's on a private register (what an epilogue chore looks like to the issue stage: a dependent chain)
template <int N>
__device__ __forceinline__ void filler(float &d)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d));
}
template <int N>
__device__ __forceinline__ void filler2(float &d, float &e)      // two independent chains, interleaved
{
#pragma unroll
    for (int i = 0; i < N; ++i) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e)); }
}
// 0: f16 only, 1: scaled fp8(A) x bf8(B), 2: scaled fp6 x fp6, 3: scaled fp4 x fp4, 4: the triple f16, f16, scaled on 2 accumulators
// 5: triple by triple with a 9-instruction chain after the first f16 MFMA and after the scaled one (the kernel's first r04 placement)
// 6: f16 MFMAs of both accumulators first (f1a f1b f2a f2b), then each scaled MFMA followed by two interleaved 9-instruction chains
// 7: triple by triple, back to back, each triple followed by two interleaved 9-instruction chains
// 8: as 7 with ONE 18-instruction chain                9: as 7 with the chains split: one after the second f16 MFMA, one after the scaled one
// 10: the fp16 part as FOUR v_mfma_f32_32x32x8_f16 (same work, twice the instructions), scaled MFMA, then [9|9]      11: 32x32x8 f16 alone (4 chains)
template <int MODE, int THREADS = 256>      // THREADS 512: two waves per SIMD (256 registers each) — does the partner wave's matrix work hide the VALU chains?
__global__ __launch_bounds__(THREADS, 1) void k(const v8i *g, float *out, unsigned long long *cyc, int iters)
{
    v16f acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    v8i a[4], b[4];
    for (int u = 0; u < 4; ++u) { a[u] = g[(threadIdx.x & 255) + 256 * u]; b[u] = g[(threadIdx.x & 255) + 256 * (4 + u)]; }
    const int sc = 0x7f7f7f7f;
    float f0 = 0.5f, f1 = 0.25f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 4], b[u % 4], 0, 1, 2, 3));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[c], 0, 0, 0);
                }
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c) % 4], b[u % 4], acc[c], 0, 1, 0, sc, 0, sc);
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c) % 4], b[u % 4], acc[c], 2, 2, 0, sc, 0, sc);
            } else if constexpr (MODE == 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c) % 4], b[u % 4], acc[c], 4, 4, 0, sc, 0, sc);
            } else if constexpr (MODE == 11) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 4], b[u % 4], 0, 1, 2, 3));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(x, x, 0, 1, 2, 3), __builtin_shufflevector(y, y, 0, 1, 2, 3), acc[c], 0, 0, 0);
                }
            } else if constexpr (MODE == 10) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 4], b[u % 4], 0, 1, 2, 3));
                    const f16x8 x2 = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 4, 5, 6, 7));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(x, x, 0, 1, 2, 3), __builtin_shufflevector(y, y, 0, 1, 2, 3), acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(x, x, 4, 5, 6, 7), __builtin_shufflevector(y, y, 4, 5, 6, 7), acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(x2, x2, 0, 1, 2, 3), __builtin_shufflevector(y, y, 0, 1, 2, 3), acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(x2, x2, 4, 5, 6, 7), __builtin_shufflevector(y, y, 4, 5, 6, 7), acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c + 1) % 4], b[u % 4], acc[c], 0, 1, 0, sc, 0, sc);
                    __builtin_amdgcn_sched_barrier(0);
                    filler2<9>(f0, f1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (MODE == 6) {
                const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 4], b[u % 4], 0, 1, 2, 3));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[c], 0, 0, 0);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 x2 = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 4, 5, 6, 7));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2, y, acc[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c + 1) % 4], b[u % 4], acc[c], 0, 1, 0, sc, 0, sc);
                    __builtin_amdgcn_sched_barrier(0);
                    filler2<9>(f0, f1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 4], b[u % 4], 0, 1, 2, 3));
                    const f16x8 x2 = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 4, 5, 6, 7));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[c], 0, 0, 0);
                    if constexpr (MODE == 5) { __builtin_amdgcn_sched_barrier(0); filler<9>(f0); __builtin_amdgcn_sched_barrier(0); }
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2, y, acc[c], 0, 0, 0);
                    if constexpr (MODE == 9) { __builtin_amdgcn_sched_barrier(0); filler<9>(f0); __builtin_amdgcn_sched_barrier(0); }
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c + 1) % 4], b[u % 4], acc[c], 0, 1, 0, sc, 0, sc);
                    if constexpr (MODE >= 5) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (MODE == 5 || MODE == 9) filler<9>(f1);
                        else if constexpr (MODE == 7) filler2<9>(f0, f1);
                        else filler<18>(f0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = f0 + f1;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    unsigned s = 7u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<unsigned> h(256 * 8 * 8);
    for (auto &v : h) {      // bytes with moderate exponents in every 8-bit format and as fp16 pairs
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) w |= (((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7)) << (8 * b);
        v = w;
    }
    v8i *g; float *o; unsigned long long *c;
    hipMalloc(&g, h.size() * 4); hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&o, 512 * 256 * 4); hipMalloc(&c, 512 * 8);
    const int iters = 400;
    const char *names[12] = {"f16 32x32x16 (4 chains)", "scaled fp8 x bf8 32x32x64 (4 chains)", "scaled fp6 32x32x64 (4 chains)", "scaled fp4 32x32x64 (4 chains)",
                             "triple f16, f16, scaled fp8 (2 chains)", "triple: f16 [9] f16 MX [9]", "f1a f1b f2a f2b, MX [9|9] x2", "triple back to back, then [9|9]",
                             "triple back to back, then [18]", "triple: f16 f16 [9] MX [9]",
                             "4 x 32x32x8 f16 + MX, then [9|9]", "f16 32x32x8 (4 chains)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // short launches (0.2 ms: the power controller has not reacted) and long ones (tens of ms on every CU: the sustained, power-limited state);
    // wall time per instruction next to the s_memtime ticks tells whether the counter follows the throttled clock
    for (int grid : {1, 256})
        for (int iters : {400, 40000}) {
            if (grid == 1 && iters > 400) continue;
            for (int mode = 0; mode < 12; ++mode) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 7: hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 8: hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 9: hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    case 10: hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    default: hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, g, o, c, iters); break;
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                }
                std::vector<unsigned long long> hc(grid);
                hipMemcpy(hc.data(), c, grid * 8, hipMemcpyDeviceToHost);
                double avg = 0; for (auto v : hc) avg += v; avg /= grid;
                const double n = (mode < 4 || mode == 11) ? iters * 8.0 * 4 : iters * 8.0 * 2;      // instructions (or triples) per wave
                printf("grid %3d iters %5d  %-42s %7.1f ticks, %7.2f ns per %s  (launch %.2f ms, %.2f ticks/ns)\n", grid, iters, names[mode], avg / n, ms * 1e6 / n,
                       (mode < 4 || mode == 11) ? "instruction" : "triple", ms, avg / (ms * 1e6));
            }
        }
    // two waves per SIMD: ticks per triple of ONE wave (the pipe is shared: 256 = fully hidden VALU, 2 x the lone-wave figure = nothing hidden)
    for (int grid : {1, 256})
        for (int iters : {400, 40000}) {
            if (grid == 1 && iters > 400) continue;
            for (int mode : {4, 7, 8}) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 4) hipLaunchKernelGGL((k<4, 512>), dim3(grid), dim3(512), 0, 0, g, o, c, iters);
                    else if (mode == 7) hipLaunchKernelGGL((k<7, 512>), dim3(grid), dim3(512), 0, 0, g, o, c, iters);
                    else hipLaunchKernelGGL((k<8, 512>), dim3(grid), dim3(512), 0, 0, g, o, c, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                }
                std::vector<unsigned long long> hc(grid);
                hipMemcpy(hc.data(), c, grid * 8, hipMemcpyDeviceToHost);
                double avg = 0; for (auto v : hc) avg += v; avg /= grid;
                const double n = iters * 8.0 * 2;
                printf("2 waves/SIMD grid %3d iters %5d  %-42s %7.1f ticks per triple of one wave, %7.2f ns per triple of the SIMD  (launch %.2f ms, %.2f ticks/ns)\n", grid, iters,
                       names[mode], avg / n, ms * 1e6 / n / 2, ms, avg / (ms * 1e6));
            }
        }
    return 0;
}
