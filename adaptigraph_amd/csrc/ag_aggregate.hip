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
    // XCD-aware block -> node-range mapping: the dispatcher puts block b on XCD b % 8 (MI355X_MICROARCH.md
    // §Workgroup dispatch); give each XCD a contiguous range of nodes (= whole graphs) so the gathered
    // Hs rows of a graph stay in ONE XCD's L2 instead of being replicated in all eight.
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;

    const int tid = threadIdx.x;
    if (tid >= kNodesPerBlock * 40) return;
    const int slot = tid / 40, c = tid - slot * 40;
    const int g = logical * kNodesPerBlock + slot;
    if (g >= a.B * a.N) return;
    const int e0 = a.row_ptr[g], e1 = a.row_ptr[g + 1];
    const float4 hr = *reinterpret_cast<const float4 *>(a.hr + (size_t)g * AG_FP + 4 * c);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0;
    for (; e + 1 < e1; e += 2) {   // two edges in flight per lane
        const int s0 = a.edge_send[e], s1 = a.edge_send[e + 1];
        const float4 t0 = *reinterpret_cast<const float4 *>(a.eterm + (size_t)e * AG_FP + 4 * c);
        const float4 t1 = *reinterpret_cast<const float4 *>(a.eterm + (size_t)(e + 1) * AG_FP + 4 * c);
        const float4 u0 = *reinterpret_cast<const float4 *>(a.hs + (size_t)s0 * AG_FP + 4 * c);
        const float4 u1 = *reinterpret_cast<const float4 *>(a.hs + (size_t)s1 * AG_FP + 4 * c);
        acc.x += fmaxf((t0.x + hr.x) + u0.x, 0.f); acc.y += fmaxf((t0.y + hr.y) + u0.y, 0.f);
        acc.z += fmaxf((t0.z + hr.z) + u0.z, 0.f); acc.w += fmaxf((t0.w + hr.w) + u0.w, 0.f);
        acc.x += fmaxf((t1.x + hr.x) + u1.x, 0.f); acc.y += fmaxf((t1.y + hr.y) + u1.y, 0.f);
        acc.z += fmaxf((t1.z + hr.z) + u1.z, 0.f); acc.w += fmaxf((t1.w + hr.w) + u1.w, 0.f);
    }
    if (e < e1) {
        const int s0 = a.edge_send[e];
        const float4 t0 = *reinterpret_cast<const float4 *>(a.eterm + (size_t)e * AG_FP + 4 * c);
        const float4 u0 = *reinterpret_cast<const float4 *>(a.hs + (size_t)s0 * AG_FP + 4 * c);
        acc.x += fmaxf((t0.x + hr.x) + u0.x, 0.f); acc.y += fmaxf((t0.y + hr.y) + u0.y, 0.f);
        acc.z += fmaxf((t0.z + hr.z) + u0.z, 0.f); acc.w += fmaxf((t0.w + hr.w) + u0.w, 0.f);
    }
    *reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + 4 * c) = acc;
}

}  // namespace

void ag_launch_aggregate(const AgFwdArgs &a, hipStream_t s)
{
    const int nodes = a.B * a.N;
    const int nb = (nodes + kNodesPerBlock - 1) / kNodesPerBlock;
    hipLaunchKernelGGL(aggregate_kernel, dim3(nb), dim3(256), 0, s, a);
}
