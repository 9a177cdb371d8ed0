// One wave per SIMD (512-register kernels): what does an MFMA cost (s_memtime cycles) depending on
//   CD  : accumulators in architectural (v) or accumulation (a) registers
//   A   : A operand from architectural (v) or accumulation (a) registers
//   ORD : 0 = (lo, hi) of one accumulator back to back, then the other accumulator; 1 = lo0 lo1 hi0 hi1
//   FILL: independent VALU instructions issued after every MFMA pair (0, 3, 6, 9)
// Two accumulator chains, 40 MFMAs per "phase", fp16 32x32x16, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool CD_A, bool A_A>
__device__ __forceinline__ void mfma(f32x16 &acc, const f16x8 &w, const f16x8 &x)
{
    if constexpr (CD_A && A_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(x));
    else if constexpr (CD_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
    else if constexpr (A_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}

template <bool CD_A, bool A_A, int ORD, int FILL>
__global__ __launch_bounds__(256, 1) void k(const f16x8 *g, float *out, long long *cyc, int iters)
{
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 w[2][4], x[4];
    for (int c = 0; c < 2; ++c)
        for (int u = 0; u < 4; ++u) {
            w[c][u] = g[threadIdx.x + 256 * (c * 4 + u)];
            if (A_A) asm volatile("" : "+a"(w[c][u])); else asm volatile("" : "+v"(w[c][u]));
        }
    for (int u = 0; u < 4; ++u) x[u] = g[threadIdx.x + 256 * (8 + u)];
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = (float)threadIdx.x + i;
    if (CD_A) { asm volatile("" : "+a"(acc[0])); asm volatile("" : "+a"(acc[1])); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            if (ORD == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    mfma<CD_A, A_A>(acc[c], w[c][u & 3], x[u & 3]);
                    mfma<CD_A, A_A>(acc[c], w[c][(u + 1) & 3], x[u & 3]);
#pragma unroll
                    for (int i = 0; i < FILL; ++i) asm volatile("v_max_i32 %0, 0, %0" : "+v"(f[(i + 6 * c) % 12]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        mfma<CD_A, A_A>(acc[c], w[c][(u + hl) & 3], x[u & 3]);
#pragma unroll
                        for (int i = 0; i < (FILL + 1) / 2; ++i) asm volatile("v_max_i32 %0, 0, %0" : "+v"(f[(i + 6 * c) % 12]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (CD_A) { asm volatile("" : "+a"(acc[0])); asm volatile("" : "+a"(acc[1])); }
    float s = 0;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 12; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool CD_A, bool A_A, int ORD, int FILL>
void run(const f16x8 *g, float *o, long long *c)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CD_A, A_A, ORD, FILL>), dim3(256), dim3(256), 0, 0, g, o, c, iters);
    long long h[4];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("acc in %s, A from %s, order %s, %d fillers per pair: %.1f cycles per MFMA\n", CD_A ? "a" : "v", A_A ? "a" : "v",
           ORD ? "lo0 lo1 hi0 hi1" : "lo0 hi0 lo1 hi1", FILL, (double)h[0] / (iters * 40.0));
}

int main()
{
    std::vector<unsigned short> hbuf(256 * 12 * 8);
    unsigned s = 12345u;
    for (auto &v : hbuf) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 31) << 15) | ((8 + ((s >> 8) & 7)) << 10) | ((s >> 16) & 0x3ff)); }
    f16x8 *g; float *o; long long *c;
    (void)hipMalloc(&g, hbuf.size() * 2); (void)hipMemcpy(g, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 8);
    run<false, false, 0, 0>(g, o, c); run<false, false, 0, 6>(g, o, c); run<false, false, 1, 6>(g, o, c);
    run<false, true, 0, 0>(g, o, c);  run<false, true, 0, 6>(g, o, c);  run<false, true, 1, 6>(g, o, c);
    run<true, true, 0, 0>(g, o, c);   run<true, true, 0, 6>(g, o, c);   run<true, true, 1, 6>(g, o, c);
    run<true, false, 0, 6>(g, o, c);  run<false, true, 0, 3>(g, o, c);  run<false, true, 0, 9>(g, o, c);
    run<true, true, 0, 9>(g, o, c);   run<true, true, 1, 0>(g, o, c);
    return 0;
}
