// ag_cost.hip — trajectory cost terms of the MPPI planner that touch every particle of every sampled rollout.
//
// chamfer (src/planning/losses.py:4-10):  x (B,N,3), y (By,M,3), By in {1,B}
//     dis[b,m,n] = ||x[b,n] - y[b,m]||_2 ;  out[b] = mean_m min_n dis + mean_n min_m dis
// The reference materialises two (B,M,N,3) repeats (12 GB at B=1024, N=M=1000); here one workgroup per sample keeps
// both clouds in LDS and does the N*M pair sweep twice (rows / columns) with broadcast reads.  sqrt is monotone, so
// min(sqrt(d2)) == sqrt(min d2) bit-for-bit; only the two means differ from torch by summation order.
#include "ag_common.h"

namespace {

// xmask / ymask: optional per-point validity (u8, (B,N) / (By,M)); masked-out points take no part in either direction —
// this is mean_chamfer's `state[i][mask[i]]` compaction (losses.py:12-24) without the per-sample host loop.  Invalid
// points are parked at +inf in LDS (their squared distance to anything finite is +inf, so no min ever picks them) and
// skipped as query points.  A sample with no valid point on either side yields NaN (the reference raises there).
// r06: the sweep is VALU work (9 instructions per pair with separately rounded products: what torch's ((x - y) ** 2).sum(-1) computes) and was bound by its LDS
// reads instead — one query point per thread and trip, three broadcast reads per pair.  Now a thread keeps FOUR query points in registers and walks the other
// cloud two points per trip (structure-of-arrays in LDS: three 8-byte broadcast reads per eight pairs; the two points of a trip as packed fp32 operations): the
// same arithmetic per pair, the same minimum (min is exact in any order), the same order of a thread's additions — the same bits, 0.90 -> 0.4x ms per
// 1 024 x 1 000 x 1 000 call.
typedef float ag_f2 __attribute__((ext_vector_type(2)));
constexpr int kChamQ = 4;      // query points per thread and sweep

// one direction: for every valid point of `q` (SoA planes qx / qy / qz, Q points) the squared distance to its nearest point of `o` (On points, On even, padded
// with +inf); adds sqrt(min) to s and 1 to c in ascending point order per thread
__device__ __forceinline__ void chamfer_sweep(const float *qx, const float *qy, const float *qz, int Q, const float *ox, const float *oy, const float *oz, int On,
                                              int tid, float &s, float &c)
{
    for (int q0 = tid; q0 < Q; q0 += 256 * kChamQ) {
        float a0[kChamQ], a1[kChamQ], a2[kChamQ], best[kChamQ];
        bool ok[kChamQ];
#pragma unroll
        for (int k = 0; k < kChamQ; ++k) {
            const int q = q0 + 256 * k;
            const float v = q < Q ? qx[q] : INFINITY;
            ok[k] = v != INFINITY;
            a0[k] = ok[k] ? v : 0.f; a1[k] = ok[k] ? qy[q] : 0.f; a2[k] = ok[k] ? qz[q] : 0.f;
            best[k] = INFINITY;
        }
        for (int n = 0; n < On; n += 2) {
            const ag_f2 X = *reinterpret_cast<const ag_f2 *>(ox + n), Y = *reinterpret_cast<const ag_f2 *>(oy + n), Z = *reinterpret_cast<const ag_f2 *>(oz + n);
#pragma unroll
            for (int k = 0; k < kChamQ; ++k) {
                const ag_f2 d0 = X - a0[k], d1 = Y - a1[k], d2 = Z - a2[k];
                const ag_f2 d = (d0 * d0 + d1 * d1) + d2 * d2;
                best[k] = fminf(best[k], fminf(d.x, d.y));
            }
        }
#pragma unroll
        for (int k = 0; k < kChamQ; ++k)
            if (ok[k]) { s += sqrtf(best[k]); c += 1.f; }
    }
}

__global__ __launch_bounds__(256) void chamfer_kernel(const float *x, const float *y, const unsigned char *xmask,
                                                      const unsigned char *ymask, int N, int M, int y_batched, float *out)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[4][4];
    const int Np = (N + 1) & ~1, Mp = (M + 1) & ~1;      // planes of even length (8-byte reads of point pairs): the pad point sits at +inf
    float *sx = sm, *sy = sm + 3 * Np;                    // x cloud: planes sx, sx + Np, sx + 2 Np; y cloud likewise
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int by = y_batched ? b : 0;
    const float *xb = x + (size_t)b * N * 3, *yb = y + (size_t)by * M * 3;
    const unsigned char *xm = xmask ? xmask + (size_t)b * N : nullptr, *ym = ymask ? ymask + (size_t)by * M : nullptr;
    for (int i = tid; i < 3 * Np; i += 256) {
        const int n = i / 3, c = i - 3 * n;
        sx[c * Np + n] = (n >= N || (xm && !xm[n])) ? INFINITY : xb[i];
    }
    for (int i = tid; i < 3 * Mp; i += 256) {
        const int m = i / 3, c = i - 3 * m;
        sy[c * Mp + m] = (m >= M || (ym && !ym[m])) ? INFINITY : yb[i];
    }
    __syncthreads();
    float s_y = 0.f, s_x = 0.f, c_y = 0.f, c_x = 0.f;
    chamfer_sweep(sy, sy + Mp, sy + 2 * Mp, M, sx, sx + Np, sx + 2 * Np, Np, tid, s_y, c_y);      // for every target point: nearest particle
    chamfer_sweep(sx, sx + Np, sx + 2 * Np, N, sy, sy + Mp, sy + 2 * Mp, Mp, tid, s_x, c_x);      // for every particle: nearest target point
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s_y += __shfl_xor(s_y, o); s_x += __shfl_xor(s_x, o);
        c_y += __shfl_xor(c_y, o); c_x += __shfl_xor(c_x, o);
    }
    if (lane == 0) { red[0][wave] = s_y; red[1][wave] = s_x; red[2][wave] = c_y; red[3][wave] = c_x; }
    __syncthreads();
    if (tid == 0) {
        const float ny = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]), nx = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
        const float v = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / ny + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / nx;
        out[b] = (nx > 0.f && ny > 0.f) ? v : NAN;
    }
}

}  // namespace

int ag_launch_chamfer(const float *x, const float *y, const unsigned char *xmask, const unsigned char *ymask, int B, int N, int M,
                      int y_batched, float *out, hipStream_t s)
{
    const size_t smem = (size_t)3 * (((N + 1) & ~1) + ((M + 1) & ~1)) * sizeof(float);      // (planes padded to an even number of points)
    if ((size_t)3 * (N + M) * sizeof(float) > 150 * 1024) return -1;
    if (smem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(chamfer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(chamfer_kernel, dim3(B), dim3(256), smem, s, x, y, xmask, ymask, N, M, y_batched, out);
    return 0;
}
