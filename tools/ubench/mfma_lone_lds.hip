// One wave per SIMD, the loop shape of the weight-stationary edge encoder: per k16-step ONE ds_read_b128 of the B operand
// (issued two steps ahead, counted lgkmcnt wait), then (lo, hi) MFMA pairs on two accumulators.  What does the LDS-fed loop
// cost per MFMA compared with register-resident B operands (tools/ubench/mfma_lone.hip: 32.2)?
//   MODE 0: B from registers (reference)   1: ds_read + s_waitcnt lgkmcnt(2) per step   2: mode 1 + sched_barrier(0) after each pair
//   3: mode 2 with the wait placed one pair later (the read lands under the first pair of the step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma(f32x16 &acc, const f16x8 &w, const f16x8 &x)
{
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(x));
}
template <int OFF> __device__ __forceinline__ void lds_read(f16x8 &d, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF)); }
template <int N> __device__ __forceinline__ void wait(f16x8 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const f16x8 *g, float *out, long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) f16x8 img[10 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 640; i += 256) img[i] = g[i];
    __syncthreads();
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 w[2][10][2];
    for (int c = 0; c < 2; ++c)
        for (int u = 0; u < 10; ++u)
            for (int h = 0; h < 2; ++h) { w[c][u][h] = g[threadIdx.x + 256 * ((c * 10 + u) * 2 + h) % 4096]; asm volatile("" : "+a"(w[c][u][h])); }
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)img + lane * 16;
    f16x8 q[3];
    if (MODE == 0) for (int u = 0; u < 3; ++u) q[u] = img[u * 64 + lane];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE) { lds_read<0>(q[0], la); lds_read<1024>(q[1], la); }
#define STEP(U)                                                                                         \
        {                                                                                               \
            if (MODE && (U) + 2 < 10) lds_read<((U) + 2) * 1024>(q[((U) + 2) % 3], la);                 \
            constexpr int later = (9 - (U)) < 2 ? (9 - (U)) : 2;                                        \
            if (MODE == 1 || MODE == 2) wait<later>(q[(U) % 3]);                                        \
            if (MODE == 3 && (U) == 0) wait<later>(q[(U) % 3]);                                         \
            mfma(acc[0], w[0][U][1], q[(U) % 3]); mfma(acc[0], w[0][U][0], q[(U) % 3]);                  \
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);                                           \
            if (MODE == 3 && (U) < 9) wait<((8 - (U)) < 2 ? (8 - (U)) : 2) + 0>(q[((U) + 1) % 3]);      \
            mfma(acc[1], w[1][U][1], q[(U) % 3]); mfma(acc[1], w[1][U][0], q[(U) % 3]);                  \
            if (MODE >= 2) __builtin_amdgcn_sched_barrier(0);                                           \
        }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9)
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const f16x8 *g, float *o, long long *c)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, g, o, c, iters);
    long long h[4];
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: %.1f cycles per MFMA\n", MODE, (double)h[0] / (iters * 40.0));
}

int main()
{
    std::vector<unsigned short> hbuf(4096 * 8 + 256 * 8);
    unsigned s = 12345u;
    for (auto &v : hbuf) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 31) << 15) | ((8 + ((s >> 8) & 7)) << 10) | ((s >> 16) & 0x3ff)); }
    f16x8 *g; float *o; long long *c;
    (void)hipMalloc(&g, hbuf.size() * 2); (void)hipMemcpy(g, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 8);
    run<0>(g, o, c); run<1>(g, o, c); run<2>(g, o, c); run<3>(g, o, c);
    return 0;
}
