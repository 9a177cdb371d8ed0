// ag_train.hip — graph kernels of the TRAINING path: the gather / segment-reduce pieces of
// DynamicsPredictor.forward (src/dynamics/gnn/model.py:220-295) and their adjoints, on the CSR adjacency, and the
// weight-gradient reduction of the fused dense chains (the chains themselves: ag_mlp.hip).
//
// The reference trains with one-hot Rr/Rs `bmm`s, whose autograd is again dense bmm.  Here the forward gathers rows by
// index and segment-reduces messages over receiver-sorted edges, and the backward is the transposed pair: gradients
// w.r.t. gathered rows are segment sums over a (pointer, permutation) view of the same edges, so every reduction has
// a fixed order — no atomics, bit-reproducible gradients.  Feature width D is arbitrary (plain row-major torch tensors):
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

// ---- weight gradients of the fused dense chains (ag_mlp.hip: chain_backward_kernel) ---------------------------------------
// dW_l[o][k] = sum_rows dz_l[row][o] * prev_l[row][k]  and  db_l[o] = sum_rows dz_l[row][o]   (prev_l = input of layer l).
// The shape defeats library GEMMs: a 150 x 150 output contracted over 10^4..10^5 rows has 25 output tiles, so hipBLASLt ran
// it on 25 workgroups (88 us per layer, 4.2 of the 10.9 ms training step, profiles/r02_train_trace.txt).  Here the ROWS are split into
// slabs, one wave per (slab, layer, 32-feature strip o in [32w, 32w+32)) against all five k-tiles, two rows per exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: A = dz^T, B = prev; both read as 128-byte row segments), and
// a second kernel adds the slab partials in ascending slab order: fixed summation order, bit-reproducible, no atomics.
// The bias gradient rides along as one more k column: prev is read as 1.0 at column n_in.
// rows per workgroup: chosen per call so that slabs x layers ~ 1.5 workgroups per CU (dw_slab_rows); a multiple of 16
struct DwArgs {
    const float *dz[4], *prev[4];
    int dz_ld[4], prev_ld[4], n_in[4];
    long long rows;
    int slab;                            // rows per workgroup
    float *partial;                      // [n_slabs][n_layers][160][160]
    float *out;                          // [n_layers][160][160]: out[l][o][k] = dW_l[o][k] (k < n_in), out[l][o][n_in] = db_l[o]
    float *w_dst[4], *b_dst[4];          // optional: ACCUMULATE layer l's gradients straight into these (row stride w_ld) instead of `out`
    int w_ld[4], n_out[4];
    int n_slabs, n_layers;
};

typedef float f32x16_t __attribute__((ext_vector_type(16)));

// KT = k-tiles that hold data (1 for the narrow first layers, 5 otherwise): compile-time so that the row loop is one
// straight-line block (a run-time tile count put a branch around every load and MFMA and serialised the loads).
template <int KT>
__device__ __forceinline__ void dw_rows(const DwArgs &a, int l, long long r0, long long r1, int w, int j, int h, f32x16_t (&acc)[AG_NT])
{
    const float *dz = a.dz[l], *pv = a.prev[l];
    const int ld = a.prev_ld[l], n_in = a.n_in[l], zld = a.dz_ld[l];
    constexpr int U = 4;                                  // row pairs per fetch (U = 8 needs 304 registers: one wave per SIMD); double-buffered: the loads of the next U pairs are
    const int zc = 32 * w + j < zld ? 32 * w + j : zld - 1;   // issued before the MFMAs of the current ones
    const float zokf = 32 * w + j < zld ? 1.0f : 0.0f;
    int kc[KT];
    float kinf[KT], kone[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int k = 32 * t + j;
        kinf[t] = k < n_in ? 1.0f : 0.0f; kc[t] = k < n_in ? k : n_in - 1; kone[t] = k == n_in ? 1.0f : 0.0f;   // column n_in reads as 1 (bias gradient)
    }
    auto fetch = [&](long long rb, float (&av)[U], float (&bv)[U][KT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = rb + 2 * u + h;           // lane half h takes row rb + 2u + h (the MFMA's k index)
            // Always a valid (clamped) address, loads unconditional, values masked ARITHMETICALLY: written as selects, the
            // compiler sinks each load under its own branch and the 24 loads of a fetch serialise.
            const float okf = r < r1 ? 1.0f : 0.0f;
            const long long rc = r < r1 ? r : r1 - 1;
            av[u] = dz[(size_t)rc * zld + zc] * (okf * zokf);
#pragma unroll
            for (int t = 0; t < KT; ++t) bv[u][t] = (pv[(size_t)rc * ld + kc[t]] * kinf[t] + kone[t]) * okf;
        }
    };
    auto fma_all = [&](const float (&av)[U], const float (&bv)[U][KT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][t], acc[t], 0, 0, 0);
    };
    float a0[U], b0[U][KT], a1[U], b1[U][KT];
    fetch(r0, a0, b0);
    for (long long rb = r0; rb < r1; rb += 4 * U) {
        fetch(rb + 2 * U, a1, b1);
        fma_all(a0, b0);
        fetch(rb + 4 * U, a0, b0);
        fma_all(a1, b1);
    }
}

// One WAVE per (slab, layer, 32-feature strip) task, four tasks per 256-thread workgroup: ~2 000 equal tasks fill the 1 024
// SIMDs two deep whatever the layer count (five-wave workgroups of one slab each left a SIMD with two waves and the others
// with one, and 380 workgroups on 256 CUs ran in two uneven rounds: 269 us for the edge chain vs 55 us of fp32 MFMA time).
__global__ __launch_bounds__(256, 2) void dw_partial_kernel(DwArgs a)
{
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= a.n_slabs * a.n_layers * AG_NT) return;
    const int w = task % AG_NT, sl = task / AG_NT, l = sl % a.n_layers, slab = sl / a.n_layers;
    f32x16_t acc[AG_NT];
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const long long r0 = (long long)slab * a.slab;
    const long long r1 = r0 + a.slab < a.rows ? r0 + a.slab : a.rows;
    if (r0 < r1) {
        if (a.n_in[l] + 1 <= 32) dw_rows<1>(a, l, r0, r1, w, j, h, acc);
        else dw_rows<AG_NT>(a, l, r0, r1, w, j, h, acc);
    }
    float *dst = a.partial + ((size_t)slab * a.n_layers + l) * AG_FP * AG_FP;
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) dst[(size_t)(32 * w + 8 * q + 4 * h + p) * AG_FP + 32 * t + j] = acc[t][4 * q + p];
}

// out = sum over slabs, ascending: 32 elements x 8 slab groups per workgroup; group g adds slabs g, g+8, ... in order and the
// eight group sums meet through LDS in group order — a fixed summation tree, so results are bit-reproducible.
__global__ __launch_bounds__(256) void dw_reduce_kernel(DwArgs a)
{
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5, l = blockIdx.y;
    const int t = blockIdx.x * 32 + e;
    float s = 0.0f;
    for (int slab = g; slab < a.n_slabs; slab += 8) s += a.partial[((size_t)slab * a.n_layers + l) * AG_FP * AG_FP + t];
    part[g][e] = s;
    __syncthreads();
    if (g == 0) {
        float r = part[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) r += part[k][e];
        if (a.w_dst[l]) {            // accumulate into the parameter's .grad storage (each element has exactly one writer)
            const int o = t / AG_FP, k = t - o * AG_FP;
            if (o < a.n_out[l]) {
                if (k < a.n_in[l]) a.w_dst[l][(size_t)o * a.w_ld[l] + k] += r;
                else if (k == a.n_in[l] && a.b_dst[l]) a.b_dst[l][o] += r;
            }
        } else {
            a.out[(size_t)l * AG_FP * AG_FP + t] = r;
        }
    }
}

// rel_inputs of DynamicsPredictor.forward (model.py:220-253) from ONE per-node table tab = [attrs (A) | group (G) | state_norm (S)]:
//   out[e] = [ tab[r][:A] | tab[s][:A] | sum_G |tab[r][A:A+G] - tab[s][A:A+G]| | tab[r][A+G:] - tab[s][A+G:] ]      (r, s = receiver, sender of e)
// one thread per edge (rows are 15 / 17 floats: a few cache lines per thread, the node table is L2-resident), and its adjoint in
// two launches: per-edge gradients w.r.t. the receiver's and the sender's row, then one fixed-order reduction per node over its
// received (row_ptr) and sent (col_ptr + perm) edges.  Replaces ~25 cat / slice / abs / sum launches per model forward+backward.
__global__ __launch_bounds__(256) void edge_inputs_fwd_kernel(const float *tab, int D, int A, int G, const int *recv, const int *send, float *out,
                                                              long long E)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float *tr = tab + (size_t)recv[e] * D, *ts = tab + (size_t)send[e] * D;
    float *o = out + (size_t)e * (2 * A + 1 + (D - A - G));
    for (int k = 0; k < A; ++k) { o[k] = tr[k]; o[A + k] = ts[k]; }
    float gd = 0.0f;
    for (int k = A; k < A + G; ++k) gd += fabsf(tr[k] - ts[k]);
    o[2 * A] = gd;
    for (int k = A + G; k < D; ++k) o[2 * A + 1 + (k - A - G)] = tr[k] - ts[k];
}
// g_r[e] = d loss / d tab[recv[e]] through edge e, g_s[e] likewise for the sender (both [E][D])
__global__ __launch_bounds__(256) void edge_inputs_bwd_kernel(const float *tab, int D, int A, int G, const int *recv, const int *send,
                                                              const float *gout, float *g_r, float *g_s, long long E)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float *tr = tab + (size_t)recv[e] * D, *ts = tab + (size_t)send[e] * D;
    const float *go = gout + (size_t)e * (2 * A + 1 + (D - A - G));
    float *gr = g_r + (size_t)e * D, *gs = g_s + (size_t)e * D;
    for (int k = 0; k < A; ++k) { gr[k] = go[k]; gs[k] = go[A + k]; }
    for (int k = A; k < A + G; ++k) {
        const float d = tr[k] - ts[k], sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // d|x|/dx = sign(x), 0 at 0 (torch.abs)
        gr[k] = go[2 * A] * sg; gs[k] = -go[2 * A] * sg;
    }
    for (int k = A + G; k < D; ++k) { const float v = go[2 * A + 1 + (k - A - G)]; gr[k] = v; gs[k] = -v; }
}
// out[n] = sum_{e in row n} g_r[e] + sum_{k in col n} g_s[perm[k]]   (ascending e, then ascending k)
__global__ __launch_bounds__(256) void edge_inputs_reduce_kernel(const float *g_r, const float *g_s, const int *row_ptr, const int *col_ptr,
                                                                 const int *perm, float *out, long long total, int D)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long n = t / D;
    const int d = (int)(t - n * D);
    float acc = 0.0f;
    for (int e = row_ptr[n]; e < row_ptr[n + 1]; ++e) acc += g_r[(size_t)e * D + d];
    for (int k = col_ptr[n]; k < col_ptr[n + 1]; ++k) acc += g_s[(size_t)perm[k] * D + d];
    out[t] = acc;
}

// y = relu(a + b + c) and its adjoint g * [y > 0] — the node update's residual (model.py:36-40) as one launch each way
__global__ __launch_bounds__(256) void add3_relu_kernel(const float4 *a, const float4 *b, const float4 *c, float4 *y, long long n4)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const float4 u = a[t], v = b[t], w = c[t];
    y[t] = make_float4(fmaxf((u.x + v.x) + w.x, 0.f), fmaxf((u.y + v.y) + w.y, 0.f), fmaxf((u.z + v.z) + w.z, 0.f), fmaxf((u.w + v.w) + w.w, 0.f));
}
__global__ __launch_bounds__(256) void relu_mask_kernel(const float4 *g, const float4 *y, float4 *out, long long n4)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const float4 u = g[t], v = y[t];
    out[t] = make_float4(v.x > 0.f ? u.x : 0.f, v.y > 0.f ? u.y : 0.f, v.z > 0.f ? u.z : 0.f, v.w > 0.f ? u.w : 0.f);
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

static int dw_slab_rows(long long rows, int n_layers)
{
    long long slab = (rows * n_layers * AG_NT + 2047) / 2048;          // ~2 048 wave tasks = two per SIMD
    slab = (slab + 15) / 16 * 16;
    return (int)(slab < 64 ? 64 : (slab > 2048 ? 2048 : slab));       // >= 64 rows per task bounds the partial-sum traffic; a 256-row floor (4x fewer tasks) measured 17 % slower end to end
}
size_t ag_weight_grads_ws_floats(long long rows, int n_layers)
{
    const int slab = dw_slab_rows(rows, n_layers);
    const long long slabs = (rows + slab - 1) / slab;
    return (size_t)(slabs > 0 ? slabs : 1) * n_layers * AG_FP * AG_FP;
}
void ag_launch_weight_grads(int n_layers, const float *const *dz, const int *dz_ld, const float *const *prev, const int *prev_ld, const int *n_in,
                            long long rows, float *partial, float *out, float *const *w_dst, const int *w_ld, float *const *b_dst, const int *n_out,
                            hipStream_t s)
{
    DwArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        a.dz[l] = dz[l]; a.dz_ld[l] = dz_ld[l]; a.prev[l] = prev[l]; a.prev_ld[l] = prev_ld[l]; a.n_in[l] = n_in[l];
        a.w_dst[l] = w_dst ? w_dst[l] : nullptr; a.b_dst[l] = (w_dst && b_dst) ? b_dst[l] : nullptr;
        a.w_ld[l] = w_dst ? w_ld[l] : 0; a.n_out[l] = w_dst ? n_out[l] : 0;
    }
    a.rows = rows; a.partial = partial; a.out = out; a.n_layers = n_layers;
    a.slab = dw_slab_rows(rows, n_layers);
    a.n_slabs = (int)((rows + a.slab - 1) / a.slab);
    if (a.n_slabs < 1) a.n_slabs = 1;
    hipLaunchKernelGGL(dw_partial_kernel, dim3((a.n_slabs * n_layers * AG_NT + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(AG_FP * AG_FP / 32, n_layers), dim3(256), 0, s, a);
}

void ag_launch_add3_relu(const float *a, const float *b, const float *c, float *y, long long n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(add3_relu_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, s, reinterpret_cast<const float4 *>(a),
                                  reinterpret_cast<const float4 *>(b), reinterpret_cast<const float4 *>(c), reinterpret_cast<float4 *>(y), n / 4);
}
void ag_launch_relu_mask(const float *g, const float *y, float *out, long long n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, s, reinterpret_cast<const float4 *>(g),
                                  reinterpret_cast<const float4 *>(y), reinterpret_cast<float4 *>(out), n / 4);
}

void ag_launch_edge_inputs_fwd(const float *tab, int D, int A, int G, const int *recv, const int *send, float *out, long long E, hipStream_t s)
{
    if (E > 0) hipLaunchKernelGGL(edge_inputs_fwd_kernel, dim3(blocks_for(E)), dim3(256), 0, s, tab, D, A, G, recv, send, out, E);
}
void ag_launch_edge_inputs_bwd(const float *tab, int D, int A, int G, const int *recv, const int *send, const int *row_ptr, const int *col_ptr,
                               const int *perm, const float *gout, float *g_r, float *g_s, float *gtab, long long E, long long M, hipStream_t s)
{
    if (E > 0) hipLaunchKernelGGL(edge_inputs_bwd_kernel, dim3(blocks_for(E)), dim3(256), 0, s, tab, D, A, G, recv, send, gout, g_r, g_s, E);
    if (M > 0) hipLaunchKernelGGL(edge_inputs_reduce_kernel, dim3(blocks_for(M * D)), dim3(256), 0, s, g_r, g_s, row_ptr, col_ptr, perm, gtab, M * D, D);
}
