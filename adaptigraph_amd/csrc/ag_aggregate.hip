// ag_aggregate.hip — per-propagation-step edge -> node message passing over the CSR adjacency.
//
// Replaces, per pstep (src/dynamics/gnn/model.py:283-295):
//     effect_r = Rr.bmm(h); effect_s = Rs.bmm(h)
//     effect_rel = relu(W_rp . [relation_encode, effect_r, effect_s] + b_rp)
//     effect_rel_agg = Rr_t.bmm(effect_rel)
// with (column-block split of W_rp)
//     agg[n] = sum_{e in row n} relu( Eterm[e] + Hr[n] + Hs[send[e]] )
// where Eterm = W_rp[:, :F].enc_e + b_rp (edge_encode_kernel) and Hr/Hs = W_rp[:, F:2F].h / W_rp[:, 2F:].h
// (node kernels).  This stage is pure HBM/L2 streaming: Eterm rows are read once, in CSR order, as full
// 640-byte rows (40 lanes x 16 B); the sender rows Hs[send] are gathered from the L2-resident node table.
// Edges of a receiver are consecutive (reference edge order == CSR order), so the segment reduction is a
// register accumulation — no atomics, deterministic summation order (ascending sender id, as the reference).
#include "ag_common.h"

namespace {

constexpr int kNodesPerBlock = 6;   // 6 nodes x 40 float4 columns = 240 of 256 lanes busy

__global__ __launch_bounds__(256) void aggregate_kernel(AgFwdArgs a)
{
    ag_overflow_view(a);
    // XCD-aware block -> node-range mapping: the dispatcher puts block b on XCD b % 8 (MI355X_MICROARCH.md
    // §Workgroup dispatch); give each XCD a contiguous range of nodes (= whole graphs) so the gathered
    // Hs rows of a graph stay in ONE XCD's L2 instead of being replicated in all eight.
    // (shared-state rollout: the row count is a device word and the grid a capped upper bound — the workgroups stride over the blocks that exist;
    // otherwise gridDim.x IS the block count and the loop runs once)
    const int rows = ag_rows(a);
    const int nb = a.n_rows_dev ? (rows + kNodesPerBlock - 1) / kNodesPerBlock : (int)gridDim.x;
    const int tid = threadIdx.x;
    __shared__ float4 s_self[AG_SELF_ROWS * (AG_FP / 4)];      // the class rows of elided self-loops: one fetch per workgroup (see ag_reduce_node_q16)
    if (a.self_info) {      // (uniform)
        if (tid < AG_SELF_ROWS * (AG_FP / 4)) s_self[tid] = reinterpret_cast<const float4 *>(a.eterm)[(size_t)ag_edges(a) * (AG_FP / 4) + tid];
        __syncthreads();
    }
    if (tid >= kNodesPerBlock * 40) return;
    const int slot = tid / 40, c = tid - slot * 40;
    for (int bid = blockIdx.x; bid < nb; bid += gridDim.x) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int g = logical * kNodesPerBlock + slot;
    if (g >= rows) continue;
    const AgSelfView sv = ag_self_view(a, g);      // (an elided self-loop is a virtual edge: ag_common.h)
    const int n = sv.n;
    constexpr int kFly = 4;   // edges in flight per lane, sender indices fetched one iteration ahead (see aggregate_half_kernel)
    const size_t gr = a.hr_row ? (size_t)a.hr_row[g] : (size_t)g;
    auto sender = [&](int j) { return j < n ? (j == sv.kself ? (int)gr : a.edge_send[sv.row(j)]) : -1; };
    int s[kFly];
#pragma unroll
    for (int i = 0; i < kFly; ++i) s[i] = sender(i);
    const float4 hr = *reinterpret_cast<const float4 *>(a.hr + gr * AG_FP + 4 * c);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = 0; e < n; e += kFly) {
        int sn[kFly];
#pragma unroll
        for (int i = 0; i < kFly; ++i) sn[i] = sender(e + kFly + i);
        float4 t[kFly], u[kFly];
#pragma unroll
        for (int i = 0; i < kFly; ++i)
            if (s[i] >= 0) {
                if (e + i == sv.kself) t[i] = s_self[sv.cls * (AG_FP / 4) + c];
                else t[i] = ag_ld_nt(reinterpret_cast<const float4 *>(a.eterm + (size_t)(sv.e0 + e + i - (e + i > sv.kself ? 1 : 0)) * AG_FP + 4 * c));
                u[i] = *reinterpret_cast<const float4 *>(a.hs + (size_t)s[i] * AG_FP + 4 * c);
            }
#pragma unroll
        for (int i = 0; i < kFly; ++i)
            if (s[i] >= 0) {
                acc.x += fmaxf((t[i].x + hr.x) + u[i].x, 0.f); acc.y += fmaxf((t[i].y + hr.y) + u[i].y, 0.f);
                acc.z += fmaxf((t[i].z + hr.z) + u[i].z, 0.f); acc.w += fmaxf((t[i].w + hr.w) + u[i].w, 0.f);
            }
#pragma unroll
        for (int i = 0; i < kFly; ++i) s[i] = sn[i];
    }
    ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + 4 * c), acc);
    if (a.status && !isfinite((acc.x + acc.y) + (acc.z + acc.w))) atomicOr(a.status, 1);
    }
}

// Same reduction over the 16-bit table of precision mode 2 (q16, ag_common.h: half the dominant HBM stream).  A row is 20 x 16 B; twenty
// adjacent lanes of one wave own a node, three nodes per wave, twelve per workgroup (240 of 256 lanes busy).
constexpr int kNodesPerBlockH = 4 * AG_AGG_NODES_PER_WAVE;

template <bool HSQ>
__global__ __launch_bounds__(256) void aggregate_half_kernel(AgFwdArgs a)
{
    ag_overflow_view(a);
    const int rows = ag_rows(a);      // (shared-state rollout: a device word, the grid a capped upper bound; otherwise gridDim.x is the block count: one trip)
    const int nb = a.n_rows_dev ? (rows + kNodesPerBlockH - 1) / kNodesPerBlockH : (int)gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / AG_AGG_GROUP, c = lane - grp * AG_AGG_GROUP;
    // the class rows of elided self-loops (table rows E, E + 1): one fetch per workgroup, read by every node from LDS
    __shared__ int4 s_self[AG_SELF_ROWS * (AG_FP / 8)];
    if (a.self_info) {      // (uniform)
        if (threadIdx.x < AG_SELF_ROWS * (AG_FP / 8)) s_self[threadIdx.x] = reinterpret_cast<const int4 *>(a.eterm)[(size_t)ag_edges(a) * (AG_FP / 8) + threadIdx.x];
        __syncthreads();
    }
    if (grp >= AG_AGG_NODES_PER_WAVE) return;
    for (int bid = blockIdx.x; bid < nb; bid += gridDim.x) {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (a.agg_reverse) logical = nb - 1 - logical;
        const int g = logical * kNodesPerBlockH + wave * AG_AGG_NODES_PER_WAVE + grp;
        if (g >= rows) continue;
        float4 acc0, acc1;
        ag_reduce_node_q16<AG_AGG_IN_FLIGHT, HSQ>(a, g, c, grp * AG_AGG_GROUP, acc0, acc1, a.self_info ? s_self : nullptr);      // (6 or 8 edges in flight with the q16 sender table: no change)
        const int f0 = ag_half_lane_feature(c);
        ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + f0), acc0);
        ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + f0 + 8), acc1);
    }
}

}  // namespace

void ag_launch_aggregate(const AgFwdArgs &a, hipStream_t s)
{
    const int nodes = a.B * a.N;
    const int cap = a.n_rows_dev ? 8192 : 0x7fffffff;      // device-side row count: a bounded grid that strides over the blocks that exist
    if (a.eterm_half) {
        const int nbh = (nodes + kNodesPerBlockH - 1) / kNodesPerBlockH;
        const dim3 grid(nbh < cap ? nbh : cap);
        if (a.hs_q16) hipLaunchKernelGGL(aggregate_half_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(aggregate_half_kernel<false>, grid, dim3(256), 0, s, a);
        return;
    }
    const int nb = (nodes + kNodesPerBlock - 1) / kNodesPerBlock;
    hipLaunchKernelGGL(aggregate_kernel, dim3(nb < cap ? nb : cap), dim3(256), 0, s, a);
}
