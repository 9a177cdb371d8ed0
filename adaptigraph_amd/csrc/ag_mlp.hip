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
    const float4 *g;   // weight stream (global), chunk k at g + k*AG_CHUNK_F4; the stream is walked cyclically
    int total;         // chunks in the stream (= chunks per row tile)
    int fetch;         // next stream chunk to fetch (wraps at total)
    int buf;           // LDS buffer holding the current chunk (0/1)
    float *lds;        // 2 * AG_CHUNK_FLOATS
};

// Asynchronous global -> LDS copy of the next weight chunk (global_load_lds_dwordx4: LDS-DMA, no VGPR staging,
// no ds_write in the wave's LDS queue).  Each wave-instruction lands 64 x 16 B at a wave-uniform LDS base (M0), so
// the chunk image is copied linearly: thread t moves float4 t + 256u, u = 0..4.
// Issued from inline asm on purpose: through the builtin, hipcc (ROCm 7.2) treats the DMA as a pending LDS write
// and puts s_waitcnt vmcnt(0) in front of the very next ds_read, i.e. it waits ~1 us for the copy at the top of
// every tile.  With asm the copy stays in flight under the tile's MFMAs and is drained by pipe_wait() right
// before the tile's barrier (cdna_hip_programming.md §5 "Pipelining across barriers").  vmcnt retires in order,
// so compiler-counted waits for its own loads can only over-wait because of these extra entries, never under-wait.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off"
                 :: "s"(lds_byte_addr), "v"(gsrc) : "memory", "m0");
}

__device__ __forceinline__ void pipe_dma(ChunkPipe &P, int buf)
{
    // The chunk index is laundered through an SGPR so the optimiser cannot prove the (cyclic) address sequence
    // loop-invariant: otherwise LICM hoists ~100 64-bit addresses out of the persistent loop and spills them.
    int f = P.fetch;
    asm volatile("" : "+s"(f));
    const float4 *g = P.g + (size_t)f * AG_CHUNK_F4 + threadIdx.x;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)P.lds;
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + (buf * AG_CHUNK_FLOATS + (threadIdx.x >> 6) * 256) * 4);
#pragma unroll
    for (int u = 0; u < 5; ++u) dma16(g + 256 * u, dst + 4096 * u);
    P.fetch = P.fetch + 1 == P.total ? 0 : P.fetch + 1;
}

__device__ __forceinline__ void pipe_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void pipe_start(ChunkPipe &P)
{
    pipe_dma(P, 0);         // chunk 0
    pipe_wait();
    __syncthreads();
}

struct NoEpi {
    __device__ __forceinline__ void operator()(int, const f32x16 &) const {}
};
struct RowStoreEpi {        // store one finished 32-feature tile of the row-major [rows][160] table
    float *row;             // table + row*160 + 4h
    bool valid;
    __device__ __forceinline__ void operator()(int ti, const f32x16 &v) const
    {
        if (!valid) return;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(row + 32 * ti + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
};
struct PackStoreEpi {       // same for the fragment-image tables (h, Pn); blk_lane = table + block*5120 + h*128 + j*4
    float *blk_lane;
    __device__ __forceinline__ void operator()(int ti, const f32x16 &v) const
    {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(blk_lane + ((ti * 4 + q) * 2) * 128) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
};

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
// Weight pipeline per tile: the LDS-DMA of chunk c+1 into the idle buffer is issued first and lands under the
// tile's 76 MFMAs; one barrier per tile.  `epi(ti, acc)` runs right after a tile is finished (stores of tile ti
// then overlap the MFMAs of tile ti+1 instead of piling up behind the layer).
template <int K, int NT, bool RELU, bool BIAS, class Init, class Epi = NoEpi>
__device__ __forceinline__ void dense_layer(ChunkPipe &P, const f32x16 (&in)[(K + 32) / 32], f32x16 (&out)[NT],
                                            const Init &init, const Epi &epi = Epi{})
{
    constexpr int KE = K + (BIAS ? 1 : 0);
    constexpr int PT = (KE + 7) / 8;      // quads (= 4 k-steps = one ds_read_b128 per lane) per tile
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
        const float *buf = P.lds + P.buf * AG_CHUNK_FLOATS;
        pipe_dma(P, P.buf ^ 1);
        f32x16 acc = init(ti);
#pragma unroll
        for (int m = 0; m < PT; ++m) {
            const int t = m / 4, q = m % 4;
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
        epi(ti, acc);
        pipe_wait();
        __syncthreads();
        P.buf ^= 1;
    }
}

// ---- register image <-> HBM movers ------------------------------------------------------------
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

// Fused segment reduction for one propagation round (model.py:283-295 after the W_rp column split):
//     x[j] = sum_{e in CSR row g_j} relu( (Eterm[e] + Hr[g_j]) + Hs[send[e]] )       (ascending e = reference order)
// computed by the lane pair (j, h) that owns row j's B-operand image, so the result lands directly in the
// registers the next MFMA layer consumes (no `agg` table, no extra launch).  Memory pattern: per wave
// instruction 32 rows x 32 contiguous bytes; each 128-B line of an Eterm row is consumed by 4 consecutive
// instructions of the same wave (L1 hits), so HBM sees every Eterm byte once.  The loop runs to the largest
// in-degree in the wave with the shorter rows predicated off.
__device__ __forceinline__ void aggregate_rows(const AgFwdArgs &a, int g, bool valid, int h, f32x16 (&x)[AG_NT])
{
    int e0 = 0, deg = 0;
    if (valid) {
        e0 = a.row_ptr[g];
        deg = a.row_ptr[g + 1] - e0;
    }
    int maxdeg = deg;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
    f32x16 hr[AG_NT];
    load_rowmajor(a.hr + (size_t)g * AG_FP, hr, h);
#pragma unroll
    for (int t = 0; t < AG_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[t][r] = 0.0f;
    for (int d = 0; d < maxdeg; ++d) {
        const bool act = d < deg;
        const int e = act ? e0 + d : 0;
        const int sidx = a.edge_send[e];
        const float *er = a.eterm + (size_t)e * AG_FP + 4 * h;
        const float *sr = a.hs + (size_t)sidx * AG_FP + 4 * h;
#pragma unroll
        for (int t = 0; t < AG_NT; ++t) {
            float4 ev[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (32 * t + 8 * q < 152) {
                    ev[q] = *reinterpret_cast<const float4 *>(er + 32 * t + 8 * q);
                    sv[q] = *reinterpret_cast<const float4 *>(sr + 32 * t + 8 * q);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (32 * t + 8 * q < 152) {
                    const float m0 = fmaxf((ev[q].x + hr[t][4 * q + 0]) + sv[q].x, 0.0f);
                    const float m1 = fmaxf((ev[q].y + hr[t][4 * q + 1]) + sv[q].y, 0.0f);
                    const float m2 = fmaxf((ev[q].z + hr[t][4 * q + 2]) + sv[q].z, 0.0f);
                    const float m3 = fmaxf((ev[q].w + hr[t][4 * q + 3]) + sv[q].w, 0.0f);
                    x[t][4 * q + 0] += act ? m0 : 0.0f;
                    x[t][4 * q + 1] += act ? m1 : 0.0f;
                    x[t][4 * q + 2] += act ? m2 : 0.0f;
                    x[t][4 * q + 3] += act ? m3 : 0.0f;
                }
        }
    }
}

#define AG_LDS_DECL __shared__ __attribute__((aligned(16))) float lds[2 * AG_CHUNK_FLOATS];


// All three MLP kernels are PERSISTENT: gridDim.x <= 2 x #CUs workgroups (what the VGPR budget keeps resident)
// walk the 128-row tiles with a grid stride.  The weight-chunk ring keeps turning across row tiles (the stream is
// cyclic), so after the first tile there is no pipeline restart, no dispatch gap and no cold LDS.

// ---------------------------------------------------------------------------------------------
// Node encoder + pstep-invariant node terms.
//   enc = Encoder([attrs | phys | action])                      model.py:168-195, 268
//   h0  = enc                                                     model.py:269
//   Pn  = W_pp[:, :F] . enc + b_pp     (first column block of particle_propagator, model.py:300)
//   Hr  = W_rp[:, F:2F] . h0,  Hs = W_rp[:, 2F:3F] . h0   (receiver / sender column blocks of
//          relation_propagator applied at NODE level instead of per edge, model.py:283-289)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void node_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int ntiles = (Mn + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    ChunkPipe P{w.node_encode, 30, 0, 0, lds};
    pipe_start(P);
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = g < Mn;
        const int gc = valid ? g : 0;
        const int b = gc / a.N, i = gc - b * a.N;

        // p_inputs = [attrs(2) | physics_param (0 for tool slots) | action(3) | 1 (bias column)], k = 4h + p
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
        f32x16 x[AG_NT], y[AG_NT];
        dense_layer<AG_NODE_IN_MAX - 1, AG_NT, true, false>(P, in0, x, ZeroInit{});
        dense_layer<AG_F, AG_NT, true, true>(P, x, y, ZeroInit{});
        const size_t blk = (size_t)(tile * 4 + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
        const size_t rowoff = (size_t)gc * AG_FP + 4 * h;
        dense_layer<AG_F, AG_NT, true, true>(P, y, x, ZeroInit{}, PackStoreEpi{a.h + blk});          // x = particle_encode = h0
        dense_layer<AG_F, AG_NT, false, true>(P, x, y, ZeroInit{}, PackStoreEpi{a.pn + blk});        // Pn
        dense_layer<AG_F, AG_NT, false, false>(P, x, y, ZeroInit{}, RowStoreEpi{a.hr + rowoff, valid});  // Hr
        dense_layer<AG_F, AG_NT, false, false>(P, x, y, ZeroInit{}, RowStoreEpi{a.hs + rowoff, valid});  // Hs
    }
}

// ---------------------------------------------------------------------------------------------
// Edge encoder + pstep-invariant edge term.
//   rel_inputs = [attrs_r | attrs_s | sum|g_r - g_s| | state_norm_r - state_norm_s]   model.py:220-253
//   enc_e      = Encoder(rel_inputs)                                                   model.py:274
//   Eterm      = W_rp[:, :F] . enc_e + b_rp      (first column block of relation_propagator, model.py:289)
// The one-hot gathers Rr.bmm / Rs.bmm become indexed reads of the (L2-resident) raw node inputs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void edge_encode_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int E = a.row_ptr[Mn];
    if (a.edge_counter && blockIdx.x == 0 && tid == 0) atomicAdd(a.edge_counter, (unsigned long long)E);
    const int ntiles = (E + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    if ((int)blockIdx.x >= ntiles) return;
    ChunkPipe P{w.edge_encode, 20, 0, 0, lds};
    pipe_start(P);
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int e = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
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

        f32x16 x[AG_NT], y[AG_NT];
        dense_layer<AG_EDGE_IN + 1, AG_NT, true, false>(P, in0, x, ZeroInit{});
        dense_layer<AG_F, AG_NT, true, true>(P, x, y, ZeroInit{});
        dense_layer<AG_F, AG_NT, true, true>(P, y, x, ZeroInit{});    // relation_encode
        dense_layer<AG_F, AG_NT, false, true>(P, x, y, ZeroInit{},     // Eterm
                                              RowStoreEpi{a.eterm + (size_t)(valid ? e : 0) * AG_FP + 4 * h, valid});
    }
}

// ---------------------------------------------------------------------------------------------
// One propagation round at node level: fused segment reduce (aggregate_rows) or a pre-computed `agg` table,
// then the node update (model.py:299-301), then either the next round's node-level relation terms (Hr, Hs)
// or — after the last round — the decoder + clamp + integrate (model.py:306-309).
// ---------------------------------------------------------------------------------------------
template <bool LAST>
__global__ __launch_bounds__(AG_MLP_THREADS, 2) void node_update_kernel(AgWeights w, AgFwdArgs a)
{
    AG_LDS_DECL
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int Mn = a.B * a.N;
    const int ntiles = (Mn + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    ChunkPipe P{LAST ? w.node_last : w.node_mid, LAST ? 16 : 15, 0, 0, lds};
    pipe_start(P);
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int g = tile * AG_ROWS_PER_BLOCK + wave * 32 + j;
        const bool valid = g < Mn;
        const int gc = valid ? g : 0;

        f32x16 x[AG_NT], y[AG_NT];
        if (a.fuse_agg) aggregate_rows(a, gc, valid, h, x);
        else load_rowmajor(a.agg + (size_t)gc * AG_FP, x, h);
        const size_t blk = (size_t)(tile * 4 + wave) * AG_PACK_BLOCK + h * 128 + j * 4;
        const size_t rowoff = (size_t)gc * AG_FP + 4 * h;
        if (!LAST) {
            dense_layer<AG_F, AG_NT, true, false>(P, x, y, ResidInit{a.pn + blk, a.h + blk}, PackStoreEpi{a.h + blk});   // h'
            // Hr/Hs of the NEXT round go to the alternate tables: other workgroups of this launch may still be
            // gathering this round's Hs rows (fused aggregation reads them inside this kernel).
            dense_layer<AG_F, AG_NT, false, false>(P, y, x, ZeroInit{}, RowStoreEpi{a.hr_out + rowoff, valid});
            dense_layer<AG_F, AG_NT, false, false>(P, y, x, ZeroInit{}, RowStoreEpi{a.hs_out + rowoff, valid});
        } else {
            dense_layer<AG_F, AG_NT, true, false>(P, x, y, ResidInit{a.pn + blk, a.h + blk});   // particle_effect'

            dense_layer<AG_F, AG_NT, true, true>(P, y, x, ZeroInit{});    // linear_0 + ReLU
            dense_layer<AG_F, AG_NT, true, true>(P, x, y, ZeroInit{});    // linear_1 + ReLU
            f32x16 m[1];
            dense_layer<AG_F, 1, false, true>(P, y, m, ZeroInit{});       // linear_2 -> rows 0..2 of tile 0
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
}

}  // namespace

static inline int grid_for(int rows, int max_blocks)
{
    const int tiles = (rows + AG_ROWS_PER_BLOCK - 1) / AG_ROWS_PER_BLOCK;
    return tiles < max_blocks ? (tiles > 0 ? tiles : 1) : max_blocks;
}

void ag_launch_node_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(node_encode_kernel, dim3(grid_for(a.B * a.N, a.max_blocks)), dim3(AG_MLP_THREADS), 0, s, w, a);
}

void ag_launch_edge_encode(const AgWeights &w, const AgFwdArgs &a, hipStream_t s)
{
    if (a.e_cap <= 0) return;
    hipLaunchKernelGGL(edge_encode_kernel, dim3(grid_for(a.e_cap, a.max_blocks)), dim3(AG_MLP_THREADS), 0, s, w, a);
}

void ag_launch_node_update(const AgWeights &w, const AgFwdArgs &a, int last, hipStream_t s)
{
    const dim3 grid(grid_for(a.B * a.N, a.max_blocks)), block(AG_MLP_THREADS);
    if (last) hipLaunchKernelGGL(node_update_kernel<true>, grid, block, 0, s, w, a);
    else hipLaunchKernelGGL(node_update_kernel<false>, grid, block, 0, s, w, a);
}
