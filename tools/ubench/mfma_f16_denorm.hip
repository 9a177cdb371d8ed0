// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs on gfx950?  (A two-term fp16 weight split W = hi + lo puts most
// lo terms below 2^-14.)  A = 2^-20 everywhere (subnormal), B = 1.0: every output should be 16 * 2^-20 = 1.52587890625e-05.
// Also: B = 2^-20 with A = 1.0, and the product of two subnormals' neighbours 2^-12 * 2^-12 = 2^-24 per term (fp32 accumulate).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, float av, float bv)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}
int main()
{
    float *o, h[64];
    (void)hipMalloc(&o, 256);
    const float cases[4][2] = {{9.5367431640625e-07f, 1.0f}, {1.0f, 9.5367431640625e-07f}, {5.9604644775390625e-08f, 1.0f}, {0.000244140625f, 0.000244140625f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c[0], c[1]);
        (void)hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
        printf("A = %.10e, B = %.10e: out = %.10e, expected %.10e\n", c[0], c[1], h[0], 16.0 * (double)c[0] * (double)c[1]);
    }
    return 0;
}
