// ag_mlp.hip — fused dense-MLP kernels on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the reference's Encoder / Propagator / ParticlePredictor stacks
// (src/dynamics/gnn/model.py:4-60) and the one-hot gathers feeding them (model.py:214-253).
//
// Design (CDNA4-first):
//  * One wave owns 32 rows (edges or nodes).  The product is computed TRANSPOSED, D^T = W . X^T:
//    the weight matrix is the MFMA A operand (32 out-features x 2 k), the activations are the B
//    operand (2 k x 32 rows).  The 32x32 accumulator layout then gives lane (j = lane&31, h = lane>>5)
//    the features {32t + 8q + 4h + p} of row j — which is exactly the B-operand image the NEXT layer
//    needs if its k-loop visits k in the order (t, q, p) with lanes h=0/1 supplying k and k+4.
//    So activations never leave registers between layers: bias + ReLU are applied in place and the
//    accumulators of layer L are the operands of layer L+1.  No LDS round trip, no transposes.
//  * Weights stream through LDS in 20 KB chunk images (32 out-features x 160 floats, bias stored as
//    column 150 and multiplied by a constant-1 activation so it rides the MFMA chain), double-buffered:
//    the next chunk is fetched to registers while the current one feeds 76 MFMAs, then written to the
//    other buffer, one barrier per chunk.  The image is XOR-swizzled at 16-byte granularity
//    (col16 ^= (row >> 1) & 7) on the host, which makes the per-lane ds_read_b128 fragment reads
//    bank-conflict-free at the 640-byte row stride (MI355X_MICROARCH.md §LDS lane groups).
//  * 256-thread workgroups (one wave per SIMD), 2 workgroups per CU (40 KB LDS each, <=256 VGPR),
//    so one workgroup's barrier/epilogue hides under the other's MFMAs.
//  * fp32-input MFMA is an exact k-ordered fma chain (cdna_hip_programming.md §3), so results match the
//    reference's fp32 forward to summation-order noise (~1e-7), far inside the 1e-4 gate.
#include "ag_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // native vector (HIP's float4 is a union-y struct that defeats SROA)

namespace {

__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }

struct ChunkPipe {
    const float4 *g;   // weight stream (global), chunk c at g + c*AG_CHUNK_F4
    int c;             // chunk currently resident in LDS buffer (c & 1)
    int total;
    float *lds;        // 2 * AG_CHUNK_FLOATS
    v4f pf0, pf1, pf2, pf3, pf4;      // chunk c+1 in flight (global -> registers), written to LDS mid-tile
};

__device__ __forceinline__ void pipe_fetch(ChunkPipe &P, int chunk)
{
    const int cn = chunk < P.total ? chunk : P.total - 1;   // past the end: re-fetch the last chunk (harmless)
    const v4f *g = reinterpret_cast<const v4f *>(P.g + (size_t)cn * AG_CHUNK_F4 + threadIdx.x);
    P.pf0 = g[0]; P.pf1 = g[256]; P.pf2 = g[512]; P.pf3 = g[768]; P.pf4 = g[1024];
}

__device__ __forceinline__ void pipe_commit(ChunkPipe &P, int buf)
{
    v4f *d = reinterpret_cast<v4f *>(P.lds + buf * AG_CHUNK_FLOATS) + threadIdx.x;
    d[0] = P.pf0; d[256] = P.pf1; d[512] = P.pf2; d[768] = P.pf3; d[1024] = P.pf4;
}

__device__ __forceinline__ void pipe_start(ChunkPipe &P)
{
    pipe_fetch(P, 0);
    pipe_commit(P, 0);
    pipe_fetch(P, 1);
    __syncthreads();
}

struct ZeroInit {
    __device__ __forceinline__ f32x16 operator()(int /*ti*/) const
    {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        return acc;
    }
};

struct ResidInit {  // accumulator := Pn + h (packed tables), i.e. W_pp[:, :F].enc + b_pp + residual (model.py:36-40,299-301)
    const float *pn, *hh;   // already offset to this wave's 32-row block and this lane's (h, j)
    __device__ __forceinline__ f32x16 operator()(int ti) const
    {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = ((ti * 4 + q) * 2) * 128;
            const float4 a = *reinterpret_cast<const float4 *>(pn + off);
            const float4 b = *reinterpret_cast<const float4 *>(hh + off);
            acc[4 * q + 0] = a.x + b.x; acc[4 * q + 1] = a.y + b.y; acc[4 * q + 2] = a.z + b.z; acc[4 * q + 3] = a.w + b.w;
        }
        return acc;
    }
};

// out[ti] = act(W_chunk(ti) . in + init(ti)) for NT out-tiles.
// K = number of input columns visited (k >= K is zero padding).  With BIAS the layer's bias is column K of the
// packed weights and the matching activation "feature K" is forced to 1.0 here, so the bias rides the MFMA chain
// (columns >= AG_F of every activation table are padding, nothing else reads them).
// Weight pipeline per tile: [first half of the MFMAs] -> commit chunk c+1 (fetched half a tile ago) to the idle
// LDS buffer and fetch chunk c+2 into the same registers -> [second half] -> one barrier.  The barrier is the
// only thing between two tiles; no global or LDS-store latency sits on that path.
template <int K, int NT, bool RELU, bool BIAS, class Init>
__device__ __forceinline__ void dense_layer(ChunkPipe &P, const f32x16 (&in)[(K + 32) / 32], f32x16 (&out)[NT],
                                            const Init &init)
{
    constexpr int KE = K + (BIAS ? 1 : 0);
    constexpr int PT = (KE + 7) / 8;      // quads (= 4 k-steps = one ds_read_b128 per lane) per tile
    constexpr int MID = PT / 2;
    const int tid = threadIdx.x;
    const int lane = tid & 63, i = lane & 31, h = lane >> 5;
    // per-lane fragment addresses: row i, 16-byte column (8t + 2q + h) ^ ((i >> 1) & 7)  (host pre-swizzles the
    // chunk image the same way; keeps the 640-byte-stride rows conflict-free for ds_read_b128)
    const int sw = (i >> 1) & 7;
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = i * AG_WSTRIDE + 4 * ((2 * q + h) ^ sw);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        const float *buf = P.lds + (P.c & 1) * AG_CHUNK_FLOATS;
        f32x16 acc = init(ti);
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int t = m / 4, q = m % 4;
            if (m == MID) {
                __builtin_amdgcn_sched_barrier(0);
                pipe_commit(P, (P.c + 1) & 1);
                pipe_fetch(P, P.c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 w = *reinterpret_cast<const float4 *>(buf + qoff[q] + 32 * t);
            const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k0 = 32 * t + 8 * q + p;        // column seen by the h = 0 half (h = 1: k0 + 4)
                if (k0 < KE) {
                    float x = in[t][4 * q + p];
                    if (BIAS && (k0 == K || k0 + 4 == K)) x = (h == (k0 == K ? 0 : 1)) ? 1.0f : x;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[p], x, acc, 0, 0, 0);
                }
            }
        }
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
        }
        out[ti] = acc;
        __syncthreads();
        ++P.c;
    }
}

// ---- variant 1: weights straight from L2 (no LDS, no barriers) -------------------------------------------
// The stream is repacked "fragment-major": chunk (out-tile) c, quad m = 4t + q holds, for lane l = (i, h),
// the 4 floats W[32c' + i][32t + 8q + 4h .. +3] at float4 index c*1280 + m*64 + l, so one wave-wide
// global_load_dwordx4 moves a fully coalesced 1 KiB fragment block = the A operands of 4 MFMAs.
// Weights (1.7 MB) are L2-resident; a register ring of D quads keeps D loads in flight ahead of the MFMAs.
// Waves never synchronise, so a stalled wave never holds up its workgroup.
template <int K, int NT, bool RELU, bool BIAS, int D, class Init>
__device__ __forceinline__ void dense_layer_l2(const float4 *__restrict__ gs, int &c, const f32x16 (&in)[(K + 32) / 32],
                                               f32x16 (&out)[NT], const Init &init)
{
    constexpr int KE = K + (BIAS ? 1 : 0);
    constexpr int PT = (KE + 7) / 8;          // quads per out-tile that touch columns < KE
    constexpr int NQ = PT * NT;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const float4 *base = gs + (size_t)c * AG_CHUNK_F4 + lane;
    float4 ring[D];
#pragma unroll
    for (int n = 0; n < D && n < NQ; ++n) ring[n] = base[((n / PT) * 20 + (n % PT)) * 64];
    f32x16 acc;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const int ti = n / PT, m = n % PT, t = m / 4, q = m % 4;
        if (m == 0) acc = init(ti);
        const float4 w = ring[n % D];
        if (n + D < NQ) ring[n % D] = base[(((n + D) / PT) * 20 + ((n + D) % PT)) * 64];
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k0 = 32 * t + 8 * q + p;
            if (k0 < KE) {
                float x = in[t][4 * q + p];
                if (BIAS && (k0 == K || k0 + 4 == K)) x = (h == (k0 == K ? 0 : 1)) ? 1.0f : x;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[p], x, acc, 0, 0, 0);
            }
        }
        if (m == PT - 1) {
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
            }
            out[ti] = acc;
        }
    }
    c += NT;
}

// Policy object so the kernels are written once for both weight paths.
template <int V> struct Net;
template <> struct Net<0> {
    ChunkPipe P;
    __device__ __forceinline__ Net(const float4 *lds_stream, const float4 * /*l2_stream*/, int total, float *lds)
        : P{lds_stream, 0, total, lds, v4f(0.f), v4f(0.f), v4f(0.f), v4f(0.f), v4f(0.f)} { pipe_start(P); }
    template <int K, int NT, bool RELU, bool BIAS, class Init>
    __device__ __forceinline__ void layer(const f32x16 (&in)[(K + 32) / 32], f32x16 (&out)[NT], const Init &init)
    {
        dense_layer<K, NT, RELU, BIAS>(P, in, out, init);
    }
};
template <> struct Net<1> {
    const float4 *g;
    int c;
    __device__ __forceinline__ Net(const float4 * /*lds_stream*/, const float4 *l2_stream, int /*total*/, float * /*lds*/)
        : g(l2_stream), c(0) {}
    template <int K, int NT, bool RELU, bool BIAS, class Init>
    __device__ __forceinline__ void layer(const f32x16 (&in)[(K + 32) / 32], f32x16 (&out)[NT], const Init &init)
    {
        dense_layer_l2<K, NT, RELU, BIAS, 6>(g, c, in, out, init);
    }
};

#define AG_LDS_DECL(V)                                                                     \
    __shared__ __attribute__((aligned(16))) float lds_[(V) == 0 ? 2 * AG_CHUNK_FLOATS : 4]; \
    float *lds = lds_;

// ---- register image <-> HBM movers ------------------------------------------------------------
__device__ __forceinline__ void store_rowmajor(float *row, const f32x16 (&v)[AG_NT], int h, bool valid)
{
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(row + 32 * t + 8 * q + 4 * h) =
                make_float4(v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
}

__device__ __forceinline__ void load_rowmajor(const float *row, f32x16 (&v)[AG_NT], int h)
{
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (32 * t + 8 * q < 152) x = *reinterpret_cast<const float4 *>(row + 32 * t + 8 * q + 4 * h);
            v[t][4 * q] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
        }
}

__device__ __forceinline__ void store_packed(float *blk_lane, const f32x16 (&v)[AG_NT])
{   // blk_lane = table + block*5120 + h*128 + j*4
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(blk_lane + ((t * 4 + q) * 2) * 128) =
                make_float4(v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
}

template <int N>
__device__ __forceinline__ void copy_tiles(f32x16 (&dst)[N], const f32x16 (&src)[N])
{
#pragma unroll
    for (int t = 0; t < N; ++t) dst[t] = src[t];
}

// ---------------------------------------------------------------------------------------------
// Node encoder + pstep-invariant node terms.
//   enc = Encoder([attrs | phys | action])                      model.py:168-195, 268
//   h0  = enc                                                     model.py:269
//   Pn  = W_pp[:, :F] . enc + b_pp     (first column block of particle_propagator, model.py:300)
//   Hr  = W_rp[:, F:2F] . h0,  Hs = W_rp[:, 2F:3F] . h0   (receiver / sender column blocks of
//          relation_propagator applied at NODE level instead of per edge, model.py:283-289)
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void node_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL(V)
    if (a.prio && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int g = blockIdx.x * AG_ROWS_PER_BLOCK + wave * 32 + j;
    const bool valid = g < Mn;
    const int gc = valid ? g : 0;
    const int b = gc / a.N, i = gc - b * a.N;

    // p_inputs = [attrs(2) | physics_param (0 for tool slots) | action(3)], k = 4h + p
    f32x16 in0[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) in0[0][r] = 0.0f;
    {
        const int A = AG_ATTR, Pd = a.phys_dim;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = 4 * h + p;
            float v = 0.0f;
            if (k < A) v = a.attrs[(size_t)gc * A + k];
            else if (k < A + Pd) v = i < a.n_p ? a.phys[(size_t)b * Pd + (k - A)] : 0.0f;
            else if (k < A + Pd + 3) v = a.action[(size_t)gc * 3 + (k - A - Pd)];
            else if (k == A + Pd + 3) v = 1.0f;   // bias column of particle_encoder.model.0
            in0[0][p] = v;
        }
    }
    Net<V> net(w.node_encode, w.node_encode_l2, 30, lds);
    f32x16 x[AG_NT], y[AG_NT];
    net.template layer<AG_NODE_IN_MAX - 1, AG_NT, true, false>(in0, x, ZeroInit{});
    net.template layer<AG_F, AG_NT, true, true>(x, y, ZeroInit{});
    net.template layer<AG_F, AG_NT, true, true>(y, x, ZeroInit{});   // x = particle_encode = h0
    const size_t blk = (size_t)(blockIdx.x * 4 + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
    store_packed(a.h + blk, x);
    net.template layer<AG_F, AG_NT, false, true>(x, y, ZeroInit{});  // Pn
    store_packed(a.pn + blk, y);
    net.template layer<AG_F, AG_NT, false, false>(x, y, ZeroInit{});  // Hr
    store_rowmajor(a.hr + (size_t)gc * AG_FP, y, h, valid);
    net.template layer<AG_F, AG_NT, false, false>(x, y, ZeroInit{});  // Hs
    store_rowmajor(a.hs + (size_t)gc * AG_FP, y, h, valid);
}

// ---------------------------------------------------------------------------------------------
// Edge encoder + pstep-invariant edge term.
//   rel_inputs = [attrs_r | attrs_s | sum|g_r - g_s| | state_norm_r - state_norm_s]   model.py:220-253
//   enc_e      = Encoder(rel_inputs)                                                   model.py:274
//   Eterm      = W_rp[:, :F] . enc_e + b_rp      (first column block of relation_propagator, model.py:289)
// The one-hot gathers Rr.bmm / Rs.bmm become indexed reads of the (L2-resident) raw node inputs.
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void edge_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL(V)
    if (a.prio && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int E = a.row_ptr[Mn];
    if (a.edge_counter && blockIdx.x == 0 && tid == 0) atomicAdd(a.edge_counter, (unsigned long long)E);
    if ((int)blockIdx.x * AG_ROWS_PER_BLOCK >= E) return;
    const int e = blockIdx.x * AG_ROWS_PER_BLOCK + wave * 32 + j;
    const bool valid = e < E;
    const int r = valid ? a.edge_recv[e] : 0, s = valid ? a.edge_send[e] : 0;
    const int b = r / a.N, ri = r - b * a.N, si = s - b * a.N;

    float feat[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) feat[k] = 0.0f;
    feat[0] = a.attrs[(size_t)r * 2]; feat[1] = a.attrs[(size_t)r * 2 + 1];
    feat[2] = a.attrs[(size_t)s * 2]; feat[3] = a.attrs[(size_t)s * 2 + 1];
    {
        float gd = 0.0f;   // g = cat([p_instance, 0]) (model.py:235), group_diff = sum |g_r - g_s| (:238)
        for (int ii = 0; ii < a.n_inst; ++ii) {
            const float gr = ri < a.n_p ? a.p_instance[((size_t)b * a.n_p + ri) * a.n_inst + ii] : 0.0f;
            const float gs = si < a.n_p ? a.p_instance[((size_t)b * a.n_p + si) * a.n_inst + ii] : 0.0f;
            gd += fabsf(gr - gs);
        }
        feat[4] = gd;
    }
    feat[AG_EDGE_IN] = 1.0f;   // bias column of relation_encoder.model.0
    {
        const float *st = a.state + (size_t)b * AG_NHIS * a.N * 3;
        float pr[AG_NHIS][3], ps[AG_NHIS][3];
#pragma unroll
        for (int hh = 0; hh < AG_NHIS; ++hh)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pr[hh][c] = st[((size_t)hh * a.N + ri) * 3 + c];
                ps[hh][c] = st[((size_t)hh * a.N + si) * 3 + c];
            }
#pragma unroll
        for (int hh = 0; hh + 1 < AG_NHIS; ++hh)   // state_res = state[:,1:] - state[:,:-1]  (model.py:155)
#pragma unroll
            for (int c = 0; c < 3; ++c) feat[5 + hh * 3 + c] = (pr[hh + 1][c] - pr[hh][c]) - (ps[hh + 1][c] - ps[hh][c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) feat[5 + (AG_NHIS - 1) * 3 + c] = pr[AG_NHIS - 1][c] - ps[AG_NHIS - 1][c];
    }
    f32x16 in0[1];
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) in0[0][r16] = 0.0f;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) in0[0][4 * q + p] = h ? feat[8 * q + 4 + p] : feat[8 * q + p];

    Net<V> net(w.edge_encode, w.edge_encode_l2, 20, lds);
    f32x16 x[AG_NT], y[AG_NT];
    net.template layer<AG_EDGE_IN + 1, AG_NT, true, false>(in0, x, ZeroInit{});
    net.template layer<AG_F, AG_NT, true, true>(x, y, ZeroInit{});
    net.template layer<AG_F, AG_NT, true, true>(y, x, ZeroInit{});    // relation_encode
    net.template layer<AG_F, AG_NT, false, true>(x, y, ZeroInit{});   // Eterm
    store_rowmajor(a.eterm + (size_t)(valid ? e : 0) * AG_FP, y, h, valid);
}

// ---------------------------------------------------------------------------------------------
// Node update for one propagation step (model.py:299-301), then either the next step's node-level
// relation terms (Hr, Hs) or — after the last step — the decoder + clamp + integrate (model.py:306-309).
// ---------------------------------------------------------------------------------------------
template <bool LAST, int V>
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void node_update_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL(V)
    if (a.prio && ((blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int g = blockIdx.x * AG_ROWS_PER_BLOCK + wave * 32 + j;
    const bool valid = g < Mn;
    const int gc = valid ? g : 0;

    f32x16 x[AG_NT], y[AG_NT];
    load_rowmajor(a.agg + (size_t)gc * AG_FP, x, h);
    Net<V> net(LAST ? w.node_last : w.node_mid, LAST ? w.node_last_l2 : w.node_mid_l2, LAST ? 16 : 15, lds);
    const size_t blk = (size_t)(blockIdx.x * 4 + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
    net.template layer<AG_F, AG_NT, true, false>(x, y, ResidInit{a.pn + blk, a.h + blk});   // particle_effect'
    if (!LAST) {
        store_packed(a.h + blk, y);
        net.template layer<AG_F, AG_NT, false, false>(y, x, ZeroInit{});
        store_rowmajor(a.hr + (size_t)gc * AG_FP, x, h, valid);
        net.template layer<AG_F, AG_NT, false, false>(y, x, ZeroInit{});
        store_rowmajor(a.hs + (size_t)gc * AG_FP, x, h, valid);
    } else {
        net.template layer<AG_F, AG_NT, true, true>(y, x, ZeroInit{});    // linear_0 + ReLU
        net.template layer<AG_F, AG_NT, true, true>(x, y, ZeroInit{});    // linear_1 + ReLU
        f32x16 m[1];
        net.template layer<AG_F, 1, false, true>(y, m, ZeroInit{});       // linear_2 -> rows 0..2 of tile 0
        const int b = gc / a.N, i = gc - b * a.N;
        if (valid && h == 0 && i < a.n_p) {
            const float *cur = a.state + (((size_t)b * AG_NHIS + (AG_NHIS - 1)) * a.N + i) * 3;
            float *pm = a.pred_motion + ((size_t)b * a.n_p + i) * 3;
            float *pp = a.pred_pos + ((size_t)b * a.n_p + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float mv = m[0][c];
                pm[c] = mv;
                pp[c] = cur[c] + fminf(fmaxf(mv, -a.clamp), a.clamp);   // model.py:309
            }
        }
    }
}

}  // namespace

static inline int blocks_for(int rows) { return (rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK; }

#define AG_LAUNCH(KERNEL, rows)                                                                          \
    do {                                                                                                 \
        if (a.mlp_variant == 1)                                                                          \
            hipLaunchKernelGGL((KERNEL<1>), dim3(blocks_for(rows)), dim3(AG_MLP_THREADS), 0, s, w, a);   \
        else                                                                                             \
            hipLaunchKernelGGL((KERNEL<0>), dim3(blocks_for(rows)), dim3(AG_MLP_THREADS), 0, s, w, a);   \
    } while (0)

void ag_launch_node_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s) { AG_LAUNCH(node_encode_kernel, a.B * a.N); }

void ag_launch_edge_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    if (a.e_cap <= 0) return;
    AG_LAUNCH(edge_encode_kernel, a.e_cap);
}

void ag_launch_node_update(const AgWeights &w, const AgFwdArgs &a, int last, hipStream_t s)
{
    const dim3 grid(blocks_for(a.B * a.N)), block(AG_MLP_THREADS);
    if (last) {
        if (a.mlp_variant == 1) hipLaunchKernelGGL((node_update_kernel<true, 1>), grid, block, 0, s, w, a);
        else hipLaunchKernelGGL((node_update_kernel<true, 0>), grid, block, 0, s, w, a);
    } else {
        if (a.mlp_variant == 1) hipLaunchKernelGGL((node_update_kernel<false, 1>), grid, block, 0, s, w, a);
        else hipLaunchKernelGGL((node_update_kernel<false, 0>), grid, block, 0, s, w, a);
    }
}
