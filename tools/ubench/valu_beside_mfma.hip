// Issue cost of the edge encoder's epilogue instructions, alone and beside a wave that keeps the SIMD's matrix pipe busy.
//   hipcc --offload-arch=gfx950 -O3 -o valu_beside_mfma tools/ubench/valu_beside_mfma.hip && ./valu_beside_mfma
// One workgroup per CU.  256 threads: one wave per SIMD runs 8 independent instances of the instruction per loop trip (no dependences between
// them) and times itself with s_memtime.  512 threads: waves 4..7 do the same while waves 0..3 (their SIMD partners) issue back-to-back
// v_mfma_f32_32x32x16_f16 on four accumulators for longer than the measurement lasts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int OP>
__device__ __forceinline__ void op8(float (&o)[8], float x, float y, unsigned u, f32x2 (&p)[4], __attribute__((address_space(3))) u32x4 *lds, u32x4 q)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 3) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(o[i]) : "v"(u), "v"(x));
        else if constexpr (OP == 5) asm volatile("v_cvt_pk_bf8_f32 %0, %1, %2" : "+v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(o[i]) : "v"(x), "v"(y), "v"(u));
        else if constexpr (OP == 7) asm volatile("v_cvt_pknorm_i16_f32 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 8) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
        else if constexpr (OP == 9) asm volatile("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(o[i]) : "v"(x), "v"(y), "v"(u));
        else if constexpr (OP == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(o[i]) : "v"(x));
        else if constexpr (OP == 11) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"((unsigned)(uintptr_t)lds), "v"(q), "n"(0) : "memory");
        else if constexpr (OP == 12) asm volatile("v_and_b32 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 13) asm volatile("v_max_i32 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 14) asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(o[i]) : "v"(x), "v"(y));
        else if constexpr (OP == 15) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(o[i]) : "v"(x), "v"(y), "v"(u));
    }
}

template <int OP, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(const f16x8 *g, float *out, unsigned long long *cyc, int iters, int mfma_iters)
{
    __shared__ u32x4 sbuf[512];
    const int wave = threadIdx.x >> 6;
    float s = 0;
    if (THREADS == 512 && wave < 4) {      // the partner: matrix pipe busy
        v16f acc[4];
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        const f16x8 a = g[threadIdx.x & 255], b = g[256 + (threadIdx.x & 255)];
        for (int i = 0; i < mfma_iters; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        float o[8] = {1, 2, 3, 4, 5, 6, 7, 8};
        f32x2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {0.5f, 0.25f}, {1.5f, 2.5f}};
        const float x = 0.37f * (threadIdx.x + 1), y = 1.0f / (threadIdx.x + 2);
        const unsigned u = 0x07050301u;
        const u32x4 q = {1u, 2u, 3u, threadIdx.x};
        auto *lds = (__attribute__((address_space(3))) u32x4 *)&sbuf[threadIdx.x & 511];
        // let the partner fill the pipe first
        for (int i = 0; i < 64; ++i) op8<0>(o, x, y, u, p, lds, q);
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            op8<OP>(o, x, y, u, p, lds, q);
            op8<OP>(o, x, y, u, p, lds, q);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) s += o[i];
        for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    }
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, const f16x8 *g, float *o, unsigned long long *c)
{
    const int iters = 2000, grid = 256;
    double res[2];
    for (int two = 0; two < 2; ++two) {
        hipMemset(c, 0, grid * 8 * 8);
        // 16 instructions per trip; the partner must outlast iters * 16 * ~16 cycles: 32-cycle MFMAs x 4 per trip
        if (two) hipLaunchKernelGGL((k<OP, 512>), dim3(grid), dim3(512), 0, 0, g, o, c, iters, iters * 16 * 24 / 128 + 2000);
        else hipLaunchKernelGGL((k<OP, 256>), dim3(grid), dim3(256), 0, 0, g, o, c, iters, 0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> hc(grid * 8);
        hipMemcpy(hc.data(), c, grid * 8 * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (auto v : hc) if (v) { sum += v; ++n; }
        res[two] = sum / n / (iters * 16.0);
    }
    printf("%-28s %6.2f cycles per instruction alone   %6.2f beside a wave issuing fp16 MFMAs back to back\n", name, res[0], res[1]);
}

int main()
{
    std::vector<_Float16> h(512 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.01f * (float)((i * 37) % 101) - 0.5f);
    f16x8 *g; float *o; unsigned long long *c;
    hipMalloc(&g, h.size() * 2); hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 256 * 8 * 8);
    run<0>("v_fma_f32", g, o, c);
    run<1>("v_max_f32", g, o, c);
    run<2>("v_cvt_pk_f16_f32", g, o, c);
    run<3>("v_pk_max_u16", g, o, c);
    run<14>("v_pk_max_f16", g, o, c);
    run<4>("v_fma_mix_f32", g, o, c);
    run<5>("v_cvt_pk_bf8_f32", g, o, c);
    run<6>("v_perm_b32", g, o, c);
    run<7>("v_cvt_pknorm_i16_f32", g, o, c);
    run<8>("v_pk_mul_f32", g, o, c);
    run<9>("v_max3_f32 |.|", g, o, c);
    run<15>("v_med3_f32", g, o, c);
    run<10>("v_mov_b32", g, o, c);
    run<12>("v_and_b32", g, o, c);
    run<13>("v_max_i32", g, o, c);
    run<11>("ds_write_b128", g, o, c);
    return 0;
}
