// mx_mfma.hip — groundwork for DESIGN §11.3: the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950.
//   (1) operand layout and scale semantics, checked against a CPU sum.  D[row][col] = sum over two 32-element K blocks of
//         2^(sa - 127) 2^(sb - 127) sum_i A[row][k_i] B[k_i][col],  one E8M0 scale byte per lane (byte op_sel of the scale VGPR), and
//         fp6 / (fp4): lane l = (row or col l & 31, block l >> 5) holds ITS block's 32 elements (6 dwords) and its own scale applies to them;
//         fp8:         lane l holds 16 elements of EACH block — dwords 0-3: k = 16 (l >> 5) + 0..15 of block 0, dwords 4-7: the same of block 1 —
//                      and block b takes its scales from the lanes of half b (found by doubling one half's scales: mx probe in DESIGN §11.3).
//       Element i of an A lane pairs with element i of the B lane of the same half in both cases, so operands produced by the same conversion
//       (v_cvt_scalef32_2xpk16_fp6_f32: 32 floats -> 6 dwords in ONE instruction; weights packed on the device the same way) need no layout table.
//   (2) sustained rate with random operands on every CU (the chip is power-limited in this regime, profiles/r02_power_probe.txt), against
//       v_mfma_f32_32x32x16_f16 in the same loop shape: what a "correction product" in fp8 / fp6 / fp4 would cost next to an fp16 product.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mx_mfma tools/ubench/mx_mfma.hip && /tmp/mx_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- (1) layout -------------------------------------------------------------------------------------------------------------
__global__ void fp8_once(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *d)
{
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    d[threadIdx.x] = acc;
}
__global__ void fp6_once(const float *fa, const float *fb, const int *sa, const int *sb, v16f *d)
{
    v16f x0, x1, y0, y1;
    for (int i = 0; i < 16; ++i) {
        x0[i] = fa[threadIdx.x * 32 + i]; x1[i] = fa[threadIdx.x * 32 + 16 + i];
        y0[i] = fb[threadIdx.x * 32 + i]; y1[i] = fb[threadIdx.x * 32 + 16 + i];
    }
    const v6u pa = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x0, x1, 1.0f), pb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(y0, y1, 1.0f);
    v8i A = {(int)pa[0], (int)pa[1], (int)pa[2], (int)pa[3], (int)pa[4], (int)pa[5], 0, 0};
    v8i B = {(int)pb[0], (int)pb[1], (int)pb[2], (int)pb[3], (int)pb[4], (int)pb[5], 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 2, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    d[threadIdx.x] = acc;
}

static float e4m3(unsigned char v)      // OCP e4m3fn
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const float mag = e ? ldexpf(1.0f + m / 8.0f, e - 7) : ldexpf(m / 8.0f, -6);
    return s ? -mag : mag;
}

// ---- (3) per-lane block scales on inexact data: the conversion builtin vs inline asm with an early-clobber destination ---------------
// (through the builtin, hipcc of ROCm 7.2 may let the 6 destination registers overlap the SECOND source tuple; the instruction then converts
//  values it has already overwritten.  Whether it does depends on the register allocation of the surrounding code.)
template <bool ASM>
__device__ __forceinline__ v8i chunk6(const v16f &a, const v16f &b)
{
    float m = 0.0f;
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(fabsf(a[r]), fabsf(b[r])));
    int sb = (int)(__builtin_bit_cast(unsigned, m) >> 23) - 2;      // scale 2^(exponent(max) - 2): the block's largest value lands in [4, 8), saturating at 7.5
    sb = sb < 1 ? 1 : sb;
    const float scale = __builtin_bit_cast(float, (unsigned)sb << 23);
    v6u q;
    if constexpr (ASM) asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(q) : "v"(a), "v"(b), "v"(scale));
    else q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
    return v8i{(int)q[0], (int)q[1], (int)q[2], (int)q[3], (int)q[4], (int)q[5], sb, 0};
}
template <bool ASM>
__global__ void corr_once(const float *A, const float *B, v16f *D)      // A[32][64] (W_lo-like, +-1e-4), B[64][32] (ReLU'd activations)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v16f a0, a1, b0, b1;
    for (int r = 0; r < 16; ++r) {
        a0[r] = A[i * 64 + 32 * h + r]; a1[r] = A[i * 64 + 32 * h + 16 + r];
        b0[r] = B[(32 * h + r) * 32 + i]; b1[r] = B[(32 * h + 16 + r) * 32 + i];
    }
    const v8i qa = chunk6<ASM>(a0, a1), qb = chunk6<ASM>(b0, b1);
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc, 2, 2, 0, qa[6], 0, qb[6]);
    D[l] = acc;
}

// ---- (2) rate ---------------------------------------------------------------------------------------------------------------
template <int FMT>      // -1: f16 32x32x16;  0 fp8, 2 fp6, 4 fp4: MX 32x32x64
__global__ __launch_bounds__(256, 2) void rate(const v8i *g, float *out, int iters, int sc)
{
    v16f acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    v8i a[4], b[6];
    for (int u = 0; u < 4; ++u) a[u] = g[threadIdx.x + 256 * u];
    for (int u = 0; u < 6; ++u) b[u] = g[threadIdx.x + 256 * (4 + u)];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (FMT < 0) {
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[(u + c) % 4], a[(u + c) % 4], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[u % 6], b[u % 6], 0, 1, 2, 3));
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[c], 0, 0, 0);
                } else {
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + c) % 4], b[u % 6], acc[c], FMT, FMT, 0, sc, 0, sc);
                }
            }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    unsigned s = 2024u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 8; };
    // ---------------- (1) fp8 layout + scales
    {
        std::vector<unsigned char> A(64 * 32), B(64 * 32);
        std::vector<int> sa(64), sb(64);
        for (auto &v : A) v = (unsigned char)(((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7));       // |x| in [2^-2, 2^3)
        for (auto &v : B) v = (unsigned char)(((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7));
        for (auto &v : sa) v = 124 + rnd() % 7;
        for (auto &v : sb) v = 124 + rnd() % 7;
        v8i *da, *db; int *dsa, *dsb; v16f *dd;
        hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 64);
        hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(fp8_once, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        std::vector<float> D(64 * 16);
        hipMemcpy(D.data(), dd, 64 * 64, hipMemcpyDeviceToHost);
        double worst = 0, mag = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);      // C/D layout of the 32x32 shapes
                double ref = 0;
                for (int blk = 0; blk < 2; ++blk) {      // block blk = bytes 16 blk .. 16 blk + 15 of BOTH lane halves, scales from the lanes of half blk
                    double part = 0;
                    for (int h = 0; h < 2; ++h)
                        for (int i = 16 * blk; i < 16 * blk + 16; ++i) part += (double)e4m3(A[(h * 32 + row) * 32 + i]) * e4m3(B[(h * 32 + col) * 32 + i]);
                    ref += ldexp(part, (sa[blk * 32 + row] - 127) + (sb[blk * 32 + col] - 127));
                }
                worst = fmax(worst, fabs(ref - D[l * 16 + r])); mag = fmax(mag, fabs(ref));
            }
        printf("fp8 e4m3 x e4m3, per-lane E8M0 scales 124..130: max |D - cpu| = %.3e (max |D| %.1f) -> %s\n", worst, mag,
               worst <= 1e-4 * mag ? "layout + scale hypothesis CONFIRMED" : "hypothesis WRONG");
    }
    // ---------------- (1b) fp6 through the conversion instruction
    {
        const float vals[16] = {0.f, 0.125f, 0.25f, 0.5f, 0.75f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 6.f, 7.5f};     // exact in e2m3
        std::vector<float> A(64 * 32), B(64 * 32);
        std::vector<int> sa(64), sb(64);
        for (auto &v : A) v = vals[rnd() & 15] * ((rnd() & 1) ? -1.f : 1.f);
        for (auto &v : B) v = vals[rnd() & 15] * ((rnd() & 1) ? -1.f : 1.f);
        for (auto &v : sa) v = 124 + rnd() % 7;
        for (auto &v : sb) v = 124 + rnd() % 7;
        float *da, *db; int *dsa, *dsb; v16f *dd;
        hipMalloc(&da, 8192); hipMalloc(&db, 8192); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 64);
        hipMemcpy(da, A.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 8192, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(fp6_once, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        std::vector<float> D(64 * 16);
        hipMemcpy(D.data(), dd, 64 * 64, hipMemcpyDeviceToHost);
        double worst = 0, mag = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int h = 0; h < 2; ++h) {
                    double part = 0;
                    for (int i = 0; i < 32; ++i) part += (double)A[(h * 32 + row) * 32 + i] * B[(h * 32 + col) * 32 + i];
                    ref += ldexp(part, (sa[h * 32 + row] - 127) + (sb[h * 32 + col] - 127));
                }
                worst = fmax(worst, fabs(ref - D[l * 16 + r])); mag = fmax(mag, fabs(ref));
            }
        printf("fp6 e2m3 x e2m3 via v_cvt_scalef32_2xpk16_fp6_f32 (scale 1.0), per-lane scales: max |D - cpu| = %.3e (max |D| %.1f) -> %s\n", worst, mag,
               worst <= 1e-4 * mag ? "CONFIRMED" : "WRONG");
    }
    // ---------------- (3) a W_lo . x correction product with per-lane block scales
    {
        std::vector<float> A(32 * 64), B(64 * 32);
        auto uni = [&] { return rnd() / 16777216.0f; };
        for (auto &v : A) v = (uni() - 0.5f) * 2e-4f * (uni() < 0.3f ? 0.1f : 1.0f);
        for (auto &v : B) { const float g = (uni() + uni() + uni() - 1.5f) * 2.0f; v = g > 0 ? g : 0.0f; }
        float *dA, *dB; v16f *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 64 * 64);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int use_asm = 0; use_asm < 2; ++use_asm) {
            if (use_asm) hipLaunchKernelGGL(corr_once<true>, dim3(1), dim3(64), 0, 0, dA, dB, dD); else hipLaunchKernelGGL(corr_once<false>, dim3(1), dim3(64), 0, 0, dA, dB, dD);
            std::vector<float> D(64 * 16);
            hipMemcpy(D.data(), dD, 64 * 64, hipMemcpyDeviceToHost);
            double num = 0, den = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double ref = 0;
                    for (int kk = 0; kk < 64; ++kk) ref += (double)A[row * 64 + kk] * B[kk * 32 + col];
                    num += (D[l * 16 + r] - ref) * (D[l * 16 + r] - ref); den += ref * ref;
                }
            printf("fp6 correction product, per-lane scales from the block maxima, conversion via %s: rms relative error %.3f\n",
                   use_asm ? "inline asm (early-clobber destination)" : "the builtin", sqrt(num / den));
        }
    }
    // ---------------- (2) sustained rate, random operands
    {
        std::vector<unsigned> h(256 * 10 * 8);
        for (auto &v : h) {       // bytes: sign, exponent 5..9 of 15 (fp8: |x| in [2^-2, 2^3)), random mantissa; as fp16 pairs: finite normal numbers; fp6 / fp4: any bits are numbers
            unsigned w = 0;
            for (int b = 0; b < 4; ++b) w |= (((rnd() & 1) << 7) | ((5 + rnd() % 5) << 3) | (rnd() & 7)) << (8 * b);
            v = w;
        }
        v8i *g; float *o;
        hipMalloc(&g, h.size() * 4); hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMalloc(&o, 512 * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int grid = 512, iters = 1500;
        auto run = [&](auto kern, const char *name, double flop_per_mfma, double f16_equiv) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, g, o, iters, 120); hipDeviceSynchronize();
            float ms = 0; int n = 0; double sum = 0;
            for (int rep = 0; rep < 12; ++rep) {       // ~1-2 s: long enough for the power controller to settle
                hipEventRecord(e0);
                for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, g, o, iters, 120);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 4) { sum += ms; n += 10; }
            }
            const double mfma = (double)grid * 4 * iters * 96, per = sum / n * 1e-3;
            printf("%-22s %7.3f ms per launch  %6.0f TFLOP/s  = %.2f fp16-MFMA-times per instruction (K = %d)\n", name, per * 1e3,
                   mfma * flop_per_mfma / per / 1e12, f16_equiv > 0 ? (per / mfma) / f16_equiv : 1.0, flop_per_mfma > 40000 ? 64 : 16);
            return per / mfma;
        };
        const double t16 = run(rate<-1>, "f16 32x32x16", 32768.0, -1);
        run(rate<0>, "MX fp8 32x32x64", 131072.0, t16);
        run(rate<2>, "MX fp6 32x32x64", 131072.0, t16);
        run(rate<4>, "MX fp4 32x32x64", 131072.0, t16);
    }
    return 0;
}
