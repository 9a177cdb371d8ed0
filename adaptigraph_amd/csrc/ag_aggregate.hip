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
#include <cstdlib>

namespace {

constexpr int kNodesPerBlock = 6;   // 6 nodes x 40 float4 columns = 240 of 256 lanes busy

// Compile-time variants (a run-time test of a pointer around a load is a wave-uniform branch, and hipcc drains the memory pipeline at every such
// join — ag_common.h):  SELF = the graph has elided self-loops (AgFwdArgs::self_info: one table row per attribute class, read from LDS, added at the
// self-loop's position: ag_common.h);  DEV = the row count is a device word and the grid a capped upper bound (shared-state rollout): the workgroups stride over
// the blocks that exist.
template <bool SELF, bool DEV>
__global__ __launch_bounds__(256) void aggregate_kernel(AgFwdArgs a)
{
    ag_overflow_view(a);
    // XCD-aware block -> node-range mapping: the dispatcher puts block b on XCD b % 8 (MI355X_MICROARCH.md
    // §Workgroup dispatch); give each XCD a contiguous range of nodes (= whole graphs) so the gathered
    // Hs rows of a graph stay in ONE XCD's L2 instead of being replicated in all eight.
    const int rows = (DEV && a.n_rows_dev) ? *a.n_rows_dev : a.B * a.N;
    const int nb = DEV ? (rows + kNodesPerBlock - 1) / kNodesPerBlock : (int)gridDim.x;
    const int E = SELF ? ag_edges(a) : 0;
    const int tid = threadIdx.x;
    if (tid >= kNodesPerBlock * 40) return;
    const int slot = tid / 40, c = tid - slot * 40;
    const int elast = E > 0 ? E - 1 : 0;
    for (int bid = blockIdx.x; bid < nb; bid += gridDim.x) {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int g = logical * kNodesPerBlock + slot;
        if (g >= rows) { if constexpr (DEV) continue; else return; }
        const int e0 = a.row_ptr[g], e1 = a.row_ptr[g + 1];
        int n = e1 - e0, kself = 0x7fffffff, eself = 0;
        if constexpr (SELF) {
            const int si = a.self_info[g];
            kself = si >= 0 ? (si & 0xffff) : 0x7fffffff;
            eself = E + (si >= 0 ? si >> 16 : 0) * AG_SELF_REPL + (g & (AG_SELF_REPL - 1));      // this node's copy of its class row (ag_common.h)
            n += si >= 0 ? 1 : 0;
        }
        constexpr int kFly = 4;   // edges in flight per lane, sender indices fetched one iteration ahead (see aggregate_half_kernel)
        const size_t gr = a.hr_row ? (size_t)a.hr_row[g] : (size_t)g;
        auto pos = [&](int j) { return SELF ? min(e0 + j - (j > kself ? 1 : 0), elast) : e0 + j; };      // (an elided self-loop is a virtual edge: ag_common.h)
        auto pick = [&](int j, int raw) { return j < n ? ((SELF && j == kself) ? (int)gr : raw) : -1; };
        int s[kFly];
        if constexpr (SELF) {
            int raw[kFly];
#pragma unroll
            for (int i = 0; i < kFly; ++i) raw[i] = a.edge_send[pos(i)];
#pragma unroll
            for (int i = 0; i < kFly; ++i) { asm volatile("" : "+v"(raw[i])); s[i] = pick(i, raw[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < kFly; ++i) s[i] = e0 + i < e1 ? a.edge_send[e0 + i] : -1;
        }
        const float4 hr = *reinterpret_cast<const float4 *>(a.hr + gr * AG_FP + 4 * c);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = 0; e < n; e += kFly) {
            int sn[kFly];
#pragma unroll
            for (int i = 0; i < kFly; ++i) {
                if constexpr (SELF) sn[i] = a.edge_send[pos(e + kFly + i)];
                else sn[i] = e0 + e + kFly + i < e1 ? a.edge_send[e0 + e + kFly + i] : -1;
            }
            float4 t[kFly], u[kFly];
#pragma unroll
            for (int i = 0; i < kFly; ++i)
                if (s[i] >= 0) {
                    t[i] = ag_ld_nt(reinterpret_cast<const float4 *>(a.eterm + (size_t)((SELF && e + i == kself) ? eself : pos(e + i)) * AG_FP + 4 * c));
                    u[i] = *reinterpret_cast<const float4 *>(a.hs + (size_t)s[i] * AG_FP + 4 * c);
                }
#pragma unroll
            for (int i = 0; i < kFly; ++i)
                if (s[i] >= 0) {
                    acc.x += fmaxf((t[i].x + hr.x) + u[i].x, 0.f); acc.y += fmaxf((t[i].y + hr.y) + u[i].y, 0.f);
                    acc.z += fmaxf((t[i].z + hr.z) + u[i].z, 0.f); acc.w += fmaxf((t[i].w + hr.w) + u[i].w, 0.f);
                }
#pragma unroll
            for (int i = 0; i < kFly; ++i) {
                if constexpr (SELF) { asm volatile("" : "+v"(sn[i])); s[i] = pick(e + kFly + i, sn[i]); }
                else s[i] = sn[i];
            }
        }
        ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + 4 * c), acc);
        if (a.status && !isfinite((acc.x + acc.y) + (acc.z + acc.w))) atomicOr(a.status, 1);
        if constexpr (!DEV) return;
    }
}

// Same reduction over the 16-bit table of precision mode 2 (q16, ag_common.h: half the dominant HBM stream).  A row is 20 x 16 B; twenty
// adjacent lanes of one wave own a node, three nodes per wave, twelve per workgroup (240 of 256 lanes busy).
constexpr int kNodesPerBlockH = 4 * AG_AGG_NODES_PER_WAVE;

// Minimum workgroups per CU = waves per SIMD the register allocation must allow.  Left alone hipcc takes 94-110 registers (4 waves per SIMD); at 5
// (<= 96 registers) the reduce is 2-3 % shorter on all three workloads (0.2095 -> 0.2048 ms rope C2, 0.407 -> 0.399 granular, 0.158 -> 0.153 cloth); 6
// spills (0.349), 8 is hopeless (0.835); fewer or more edges in flight per lane (2, 3, 5, 6) change nothing or lose (profiles/r06_agg_occupancy_ab.txt).
#ifndef AG_AGG_MINB
#define AG_AGG_MINB 5
#endif
// AQ: `agg` goes out as q16 rows (option "agg_q16", ag_common.h: ag_q16_encode_segment) — 16 bytes per lane instead of 32.
template <bool HSQ, bool SELF, bool DEV, bool AQ>
__global__ __launch_bounds__(256, AG_AGG_MINB) void aggregate_half_kernel(AgFwdArgs a)
{
    ag_overflow_view(a);
    const int rows = (DEV && a.n_rows_dev) ? *a.n_rows_dev : a.B * a.N;
    const int nb = DEV ? (rows + kNodesPerBlockH - 1) / kNodesPerBlockH : (int)gridDim.x;
    const int E = SELF ? ag_edges(a) : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / AG_AGG_GROUP, c = lane - grp * AG_AGG_GROUP;
    if (grp >= AG_AGG_NODES_PER_WAVE) return;
    for (int bid = blockIdx.x; bid < nb; bid += gridDim.x) {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (a.agg_reverse) logical = nb - 1 - logical;
        const int g = logical * kNodesPerBlockH + wave * AG_AGG_NODES_PER_WAVE + grp;
        if (g >= rows) { if constexpr (DEV) continue; else return; }
        float4 acc0, acc1;
        ag_reduce_node_q16<AG_AGG_IN_FLIGHT, HSQ, SELF>(a, g, c, grp * AG_AGG_GROUP, acc0, acc1, E);      // (6 or 8 edges in flight with the q16 sender table: no change)
        if constexpr (AQ) {
            const int4 w = ag_q16_encode_segment(acc0, acc1, c, grp * AG_AGG_GROUP);
            ag_st_nt(reinterpret_cast<int4 *>(a.agg) + (size_t)g * (AG_FP / 8) + c, w);
        } else {
            const int f0 = ag_half_lane_feature(c);
            ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + f0), acc0);
            ag_st_nt(reinterpret_cast<float4 *>(a.agg + (size_t)g * AG_FP + f0 + 8), acc1);
        }
        if constexpr (!DEV) return;
    }
}

template <bool HSQ, bool SELF>
void launch_half(const AgFwdArgs &a, dim3 grid, hipStream_t s, bool loop)
{
    if (a.agg_q16) {
        if (loop) hipLaunchKernelGGL((aggregate_half_kernel<HSQ, SELF, true, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((aggregate_half_kernel<HSQ, SELF, false, true>), grid, dim3(256), 0, s, a);
        return;
    }
    if (loop) hipLaunchKernelGGL((aggregate_half_kernel<HSQ, SELF, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((aggregate_half_kernel<HSQ, SELF, false, false>), grid, dim3(256), 0, s, a);
}

}  // namespace

void ag_launch_aggregate(const AgFwdArgs &a, hipStream_t s)
{
    const int nodes = a.B * a.N;
    static const int env_cap = getenv("AG_AGG_CAP") ? atoi(getenv("AG_AGG_CAP")) : 0;      // A/B: a capped, striding grid on the plain path too
    const bool loop = a.n_rows_dev != nullptr || env_cap > 0;
    const int cap = loop ? (env_cap > 0 ? env_cap : 8192) : 0x7fffffff;      // device-side row count: a bounded grid that strides over the blocks that exist
    const bool self = a.self_info != nullptr;
    if (a.eterm_half) {
        const int nbh = (nodes + kNodesPerBlockH - 1) / kNodesPerBlockH;
        const dim3 grid(nbh < cap ? nbh : cap);
        if (a.hs_q16) { if (self) launch_half<true, true>(a, grid, s, loop); else launch_half<true, false>(a, grid, s, loop); }
        else { if (self) launch_half<false, true>(a, grid, s, loop); else launch_half<false, false>(a, grid, s, loop); }
        return;
    }
    const int nb = (nodes + kNodesPerBlock - 1) / kNodesPerBlock;
    const dim3 grid(nb < cap ? nb : cap);
    if (a.n_rows_dev) {
        if (self) hipLaunchKernelGGL((aggregate_kernel<true, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((aggregate_kernel<false, true>), grid, dim3(256), 0, s, a);
    } else {
        if (self) hipLaunchKernelGGL((aggregate_kernel<true, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((aggregate_kernel<false, false>), grid, dim3(256), 0, s, a);
    }
}
