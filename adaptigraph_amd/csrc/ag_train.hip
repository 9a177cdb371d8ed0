// ag_train.hip — graph kernels of the TRAINING path: the gather / segment-reduce pieces of
// DynamicsPredictor.forward (src/dynamics/gnn/model.py:220-295) and their adjoints, on the CSR adjacency.
//
// The reference trains with one-hot Rr/Rs `bmm`s, whose autograd is again dense bmm.  Here the forward gathers rows by
// index and segment-reduces messages over receiver-sorted edges, and the backward is the transposed pair: gradients
// w.r.t. gathered rows are segment sums over a (pointer, permutation) view of the same edges, so every reduction has
// a fixed order — no atomics, bit-reproducible gradients.  Dense layers stay library GEMMs (torch / hipBLASLt) in this
// path; only the inference path fuses them (ag_mlp.hip).  Feature width D is arbitrary (plain row-major torch tensors):
// one thread per (row, feature), consecutive threads on consecutive features, so every access is coalesced along D.
#include "ag_common.h"

namespace {

// out[e, :] = x[idx[e], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *x, const int *idx, float *out, long long total, int D)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long e = t / D;
    const int d = (int)(t - e * D);
    out[t] = x[(long long)idx[e] * D + d];
}

// out[n, :] = sum_{k in [ptr[n], ptr[n+1])} vals[(perm ? perm[k] : k), :]      (ascending k: fixed summation order)
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *vals, const int *ptr, const int *perm, float *out,
                                                          long long total, int D)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long n = t / D;
    const int d = (int)(t - n * D);
    float acc = 0.0f;
    for (int k = ptr[n]; k < ptr[n + 1]; ++k) acc += vals[(long long)(perm ? perm[k] : k) * D + d];
    out[t] = acc;
}

// agg[n, :] = sum_{e in row n} relu((eterm[e] + hr[n]) + hs[send[e]])     (model.py:283-295 after the W_rp column split)
__global__ __launch_bounds__(256) void message_fwd_kernel(const float *eterm, const float *hr, const float *hs, const int *row_ptr,
                                                          const int *send, float *agg, long long total, int D)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long n = t / D;
    const int d = (int)(t - n * D);
    const float r = hr[t];
    float acc = 0.0f;
    for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e)
        acc += fmaxf((eterm[(long long)e * D + d] + r) + hs[(long long)send[e] * D + d], 0.0f);
    agg[t] = acc;
}

// adjoint: g_e[e] = g_agg[recv(e)] * [pre-activation > 0]  (= d/d eterm[e]),  g_hr[n] = sum_{e in row n} g_e[e];
// d/d hs is the sender-side segment sum of g_e (segment_sum_kernel over the sender-sorted view).
__global__ __launch_bounds__(256) void message_bwd_kernel(const float *eterm, const float *hr, const float *hs, const int *row_ptr,
                                                          const int *send, const float *g_agg, float *g_e, float *g_hr,
                                                          long long total, int D)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long n = t / D;
    const int d = (int)(t - n * D);
    const float r = hr[t], ga = g_agg[t];
    float acc = 0.0f;
    for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) {
        const float pre = (eterm[(long long)e * D + d] + r) + hs[(long long)send[e] * D + d];
        const float gm = pre > 0.0f ? ga : 0.0f;
        g_e[(long long)e * D + d] = gm;
        acc += gm;
    }
    g_hr[t] = acc;
}

inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

void ag_launch_gather_rows(const float *x, const int *idx, float *out, long long E, int D, hipStream_t s)
{
    if (E * D > 0) hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(E * D)), dim3(256), 0, s, x, idx, out, E * D, D);
}
void ag_launch_segment_sum(const float *vals, const int *ptr, const int *perm, float *out, long long N, int D, hipStream_t s)
{
    if (N * D > 0) hipLaunchKernelGGL(segment_sum_kernel, dim3(blocks_for(N * D)), dim3(256), 0, s, vals, ptr, perm, out, N * D, D);
}
void ag_launch_message_fwd(const float *eterm, const float *hr, const float *hs, const int *row_ptr, const int *send, float *agg,
                           long long N, int D, hipStream_t s)
{
    if (N * D > 0) hipLaunchKernelGGL(message_fwd_kernel, dim3(blocks_for(N * D)), dim3(256), 0, s, eterm, hr, hs, row_ptr, send, agg, N * D, D);
}
void ag_launch_message_bwd(const float *eterm, const float *hr, const float *hs, const int *row_ptr, const int *send,
                           const float *g_agg, float *g_e, float *g_hr, long long N, int D, hipStream_t s)
{
    if (N * D > 0)
        hipLaunchKernelGGL(message_bwd_kernel, dim3(blocks_for(N * D)), dim3(256), 0, s, eterm, hr, hs, row_ptr, send, g_agg, g_e, g_hr, N * D, D);
}
