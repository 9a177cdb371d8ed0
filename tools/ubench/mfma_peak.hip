// Micro-benchmark: what MFMA rate does v_mfma_f32_32x32x2_f32 sustain in the geometry the MLP kernels use
// (256-thread workgroups, 2 per CU, one dependent accumulator chain per wave)?
//   mode 0: pure MFMA chain            mode 1: + one ds_read_b128 per 4 MFMAs (operand from LDS)
//   mode 2: mode 1 + s_barrier every 76 MFMAs      mode 3: mode 2 + 20 KB LDS refill per barrier (global -> LDS)
//   mode 4: mode 3 with LDS padded to 64 KB (exactly 2 workgroups per CU, like the 200+ VGPR kernels)
//   mode 5: mode 4 + row-per-lane 640-B row stores every 20 tiles     mode 6: mode 5 + dependent gather prologue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ g, float *out, int tiles)
{
    __shared__ __attribute__((aligned(16))) float lds[MODE >= 4 ? 16384 : 2 * 5120];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * 5120; i += 256) lds[i] = g[i];
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float b = g[tid];
    const float *wrow = lds + (tid & 31) * 160 + 4 * (((tid >> 5) & 1) ^ ((tid >> 1) & 7));
    const int lane = tid & 63, jj = lane & 31, hh = lane >> 5;
    for (int t = 0; t < tiles; ++t) {
        if (MODE >= 6 && t % 20 == 0) {   // dependent gather chain like the edge-feature prologue
            const int *ip = reinterpret_cast<const int *>(g);
            int e = (blockIdx.x * 128 + (tid >> 6) * 32 + jj + t) & 0xfffff;
            int r = ip[e] & 0xfffff;
            float v = g[r] + g[(r * 7) & 0xfffff];
            b += v * 1e-9f;
        }
        const float *buf = wrow + (t & 1) * 5120;
        v4f pf[5];
        if (MODE == 3) {
            const v4f *gp = reinterpret_cast<const v4f *>(g) + (t % 20) * 1280 + tid;
            for (int u = 0; u < 5; ++u) pf[u] = gp[256 * u];
        }
#pragma unroll
        for (int m = 0; m < 19; ++m) {
            v4f w;
            if (MODE >= 1) w = *reinterpret_cast<const v4f *>(buf + 8 * m);
            else w = v4f(b);
#pragma unroll
            for (int p = 0; p < 4; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[p], b, acc, 0, 0, 0);
        }
        if (MODE == 3) {
            v4f *d = reinterpret_cast<v4f *>(lds + ((t + 1) & 1) * 5120) + tid;
            for (int u = 0; u < 5; ++u) d[256 * u] = pf[u];
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 5 && t % 20 == 19) {   // Eterm-style epilogue: lane (j,h) stores 20 x 16 B of row j
            float *row = out + ((size_t)(blockIdx.x * 128 + (tid >> 6) * 32 + jj) * 160) + 4 * hh;
            for (int u = 0; u < 20; ++u) *reinterpret_cast<v4f *>(row + 8 * u) = v4f(acc[u & 15]);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + tid] = s;
}

int main()
{
    float *g, *o;
    hipMalloc(&g, 64 << 20);
    hipMemset(g, 0, 64 << 20);
    if (getenv("RANDOM_DATA")) {
        float *hbuf = (float *)malloc(64 << 20);
        for (size_t i = 0; i < (64u << 20) / 4; ++i) hbuf[i] = (float)rand() / RAND_MAX * 0.2f - 0.1f;
        hipMemcpy(g, hbuf, 64 << 20, hipMemcpyHostToDevice);
        free(hbuf);
    }
    hipMalloc(&o, (size_t)19300 * 128 * 160 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int tiles = 20;
    for (int grid : {512, 19239}) {
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                default: hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, g, o, tiles * (grid == 512 ? 38 : 1)); break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                const double nmfma = (double)grid * 4 * tiles * (grid == 512 ? 38 : 1) * 76;
                const double tf = nmfma * 4096 / (ms * 1e-3) / 1e12;
                if (rep == 2) printf("grid %5d mode %d: %.3f ms  %.1f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", grid, mode, ms, tf,
                                     ms * 1e-3 * 2.4e9 / (nmfma / 1024));
            }
        }
    }
    return 0;
}
